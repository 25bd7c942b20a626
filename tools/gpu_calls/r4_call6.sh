#!/bin/bash
# round 4, call 6: ks_pack_rr as the default pack kernel -- the whole GPU suite, cold-process stress (plain and with the uninitialised arena poisoned), the profile set
mkdir -p gpurun_out/r4c6
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4c6/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r4c6/gpu_suite.log
tail -4 gpurun_out/r4c6/gpu_suite.log
timeout 600 python tools/stress_cold.py --cold 40 --batches 10 > gpurun_out/r4c6/stress.log 2>&1; echo "rc=$?" >> gpurun_out/r4c6/stress.log
timeout 400 python tools/stress_cold.py --cold 20 --batches 0 --poison 0xA5 > gpurun_out/r4c6/stress_poison.log 2>&1; echo "rc=$?" >> gpurun_out/r4c6/stress_poison.log
tail -3 gpurun_out/r4c6/stress.log; tail -3 gpurun_out/r4c6/stress_poison.log
bash tools/profile_bench.sh r04 > gpurun_out/r4c6/profile.log 2>&1
tail -2 gpurun_out/r4c6/profile.log | cut -c1-600
