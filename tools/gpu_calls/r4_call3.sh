#!/bin/bash
# round 4, call 3: the rr tests on the device, the whole GPU suite on the default path, rr against ks_pack on generic-only and config #3 workloads
mkdir -p gpurun_out/r4c3
timeout 900 python -m pytest tests/test_rr_gpu.py -m gpu -x -q > gpurun_out/r4c3/rr_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4c3/rr_tests.log
tail -4 gpurun_out/r4c3/rr_tests.log
python - > gpurun_out/r4c3/ab.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, ".")
from karpenter_core_amd import scheduler as S, workloads as W
def run(p, rr):
    if rr: os.environ.pop("KS_NO_RR", None)
    else: os.environ["KS_NO_RR"] = "1"
    fp = S.FlatProblem(p); fp.upload(0); fp.grid(want_bits=False); fp.solve(decode=False); fp.solve(decode=False)
    ms = fp.kernel_ms; fp.close(); return ms
for name, p in (("config1 shape, 100k generic pods / 2000 types", W.config1(pods=100000, types=2000, seed=42)), ("config3 100k", W.config3())):
    a = run(p, False); b = run(p, True)
    print(f"{name}: ks_pack {a:.1f} ms, ks_pack_rr {b:.1f} ms")
PY
cat gpurun_out/r4c3/ab.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4c3/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r4c3/gpu_suite.log
tail -4 gpurun_out/r4c3/gpu_suite.log
