#!/bin/bash
# round 6: kernel milliseconds of experiment builds (karpenter_core_amd/_variants/<name>, tools/mkvariant_dir.sh) on BASELINE configs[2], problem resident, with the
# oracle's fingerprint; then the pack kernel's own GPU tests on the product:   r6_ab.sh name...
mkdir -p gpurun_out/r6ab
for v in "$@"; do
KS_VARIANT=$v timeout 300 python - <<'PY' 2>&1 | tail -1
import os, sys, hashlib, json
sys.path.insert(0, ".")
from karpenter_core_amd import scheduler as S, workloads as W
v = os.environ["KS_VARIANT"]
if v != "product": S._HERE = os.path.join(os.path.dirname(S.__file__), "_variants", v); S._LIBS = None; S.libs()
p = W.config3()
fp = S.FlatProblem(p); fp.upload(0); fp.grid(want_bits=False); fp.solve(decode=False)
ms = []
for _ in range(4): fp.solve(decode=False); ms.append(fp.kernel_ms)
res = fp.solve(); st = res.stats
h = hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest()
want = json.load(open("tests/golden/config_hashes.json"))
ok = [k for k, x in want.items() if x == h or (isinstance(x, dict) and h in json.dumps(x))]
b = st.get("p25", 0)
print("%-12s min %.2f ms  all %s  rounds %s window pods %s phases %s queries %s (census answered %s) runs %s run pods %s | phases ended: plain stretch %d, the loop's stop %d, spare places taken %d, topology pod %d, no template %d | fingerprint %s %s" % (v, min(ms), ["%.1f" % m for m in ms], st.get("eq_pods"), st.get("cyc_kind0"), st.get("cyc_kind1"), st.get("n_kind1"), st.get("p26"), st.get("p24"), st.get("p22"), b & 4095, (b >> 12) & 4095, (b >> 24) & 4095, (b >> 36) & 4095, (b >> 48) & 4095, h[:8], ok))
PY
done | tee gpurun_out/r6ab/variants.log
