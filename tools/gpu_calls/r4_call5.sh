#!/bin/bash
# round 4, call 5: the whole GPU suite with ks_pack_rr switched on for every single LEAN Solve (KS_RR=1), then the round's profile set
mkdir -p gpurun_out/r4c5
KS_RR=1 timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4c5/gpu_suite_rr.log 2>&1; echo "rc=$?" >> gpurun_out/r4c5/gpu_suite_rr.log
tail -5 gpurun_out/r4c5/gpu_suite_rr.log
bash tools/profile_bench.sh r04 > gpurun_out/r4c5/profile.log 2>&1
tail -3 gpurun_out/r4c5/profile.log
