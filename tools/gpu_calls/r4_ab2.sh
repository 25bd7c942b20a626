#!/bin/bash
# round 4: ks_pack_rr kernel milliseconds on BASELINE configs[2] (resident problem) + the kernel's own GPU tests; used to A/B a kernel change against the committed profile
mkdir -p gpurun_out/r4ab
python - > gpurun_out/r4ab/ab2.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
from karpenter_core_amd import scheduler as S, workloads as W
os.environ.pop("KS_NO_RR", None)
p = W.config3()
fp = S.FlatProblem(p); fp.upload(0); fp.grid(want_bits=False); r = fp.solve(decode=False)
ms = []
for _ in range(5): fp.solve(decode=False); ms.append(fp.kernel_ms)
st = fp.solve().stats
print("config3 100k ks_pack_rr: min %.2f ms  all %s  steps %s runs %s run_pods %s rounds %s" % (min(ms), ["%.1f" % m for m in ms], st.get("p23"), st.get("p24"), st.get("p22"), st.get("eq_pods")))
fp.close()
PY
cat gpurun_out/r4ab/ab2.log
timeout 900 python -m pytest tests/test_rr_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4ab/ab2_tests.log
