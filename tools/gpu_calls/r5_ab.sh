#!/bin/bash
# round 5: ks_pack_rr (head window) kernel milliseconds on BASELINE configs[2] with the problem resident, its statistics, and the kernel's own GPU tests
mkdir -p gpurun_out/r5ab
python - > gpurun_out/r5ab/ab.log 2>&1 <<'PY'
import os, sys, hashlib, json
sys.path.insert(0, ".")
from karpenter_core_amd import scheduler as S, workloads as W
os.environ.pop("KS_NO_RR", None)
p = W.config3()
if os.environ.get("KS_VARIANT"): S._HERE = os.path.join(os.path.dirname(S.__file__), "_variants", os.environ["KS_VARIANT"]); S._LIBS = None; S.libs()
fp = S.FlatProblem(p); fp.upload(0); fp.grid(want_bits=False); r = fp.solve(decode=False)
ms = []
for _ in range(5): fp.solve(decode=False); ms.append(fp.kernel_ms)
res = fp.solve(); st = res.stats
print("config3 100k ks_pack_rr: min %.2f ms  all %s  steps %s runs %s run_pods %s rounds %s window pods %s phases %s sorts %s batches %s stops %s" % (min(ms), ["%.1f" % m for m in ms], st.get("p23"), st.get("p24"), st.get("p22"), st.get("eq_pods"), st.get("cyc_kind0"), st.get("cyc_kind1"), st.get("n_kind1"), st.get("n_kind2"), [st.get("cyc_kind2", 0) & 0x1FFFFF, (st.get("cyc_kind2", 0) >> 21) & 0x1FFFFF, st.get("cyc_kind2", 0) >> 42]))
want = json.load(open("tests/golden/config_hashes.json"))
h = hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest()
print("fingerprint", h, [k for k, v in want.items() if v == h or (isinstance(v, dict) and h in json.dumps(v))])
print("raw", {k: v for k, v in st.items() if v})
fp.close()
PY
cat gpurun_out/r5ab/ab.log
timeout 900 python -m pytest tests/test_rr_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r5ab/ab_tests.log
