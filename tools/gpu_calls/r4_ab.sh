#!/bin/bash
# round 4: ks_pack against ks_pack_rr (KS_RR=1) on BASELINE configs[2], kernel milliseconds of a resident problem
mkdir -p gpurun_out/r4ab
python - > gpurun_out/r4ab/ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
from karpenter_core_amd import scheduler as S, workloads as W
def run(p, rr):
    if rr: os.environ.pop("KS_NO_RR", None)
    else: os.environ["KS_NO_RR"] = "1"
    fp = S.FlatProblem(p); fp.upload(0); fp.grid(want_bits=False); fp.solve(decode=False)
    ms = []
    for _ in range(3): fp.solve(decode=False); ms.append(fp.kernel_ms)
    fp.close(); return min(ms)
p = W.config3()
print("config3 100k: ks_pack %.1f ms, ks_pack_rr %.1f ms" % (run(p, False), run(p, True)))
PY
cat gpurun_out/r4ab/ab.log
