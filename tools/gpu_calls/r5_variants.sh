#!/bin/bash
# round 5: kernel milliseconds of experiment builds (karpenter_core_amd/_variants/<name>, tools/mkvariant_dir.sh) on BASELINE configs[2], problem resident:  r5_variants.sh name...
mkdir -p gpurun_out/r5ab
for v in "$@"; do
KS_VARIANT=$v python - <<'PY' 2>&1 | tail -1
import os, sys
sys.path.insert(0, ".")
from karpenter_core_amd import scheduler as S, workloads as W
v = os.environ["KS_VARIANT"]
if v != "product": S._HERE = os.path.join(os.path.dirname(S.__file__), "_variants", v); S._LIBS = None; S.libs()
p = W.config3()
fp = S.FlatProblem(p); fp.upload(0); fp.grid(want_bits=False); fp.solve(decode=False)
ms = []
for _ in range(3): fp.solve(decode=False); ms.append(fp.kernel_ms)
res = fp.solve(); st = res.stats
import hashlib, json
h = hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest()[:8]
print("%-12s min %.2f ms  rounds %s window pods %s phases %s queries %s fingerprint %s" % (v, min(ms), st.get("eq_pods"), st.get("cyc_kind0"), st.get("cyc_kind1"), st.get("n_kind1"), h))
PY
done | tee gpurun_out/r5ab/variants.log
