#!/bin/bash
# round 6, call 18: the record of the round on the FINAL tree (host side changed since call 7: ksh_env_apply, the queue sort / confirmation beside the flattening) --
# whole GPU suite, cold-process stress (plain and poisoned), a fresh 240-problem fuzz campaign (seeds 8000-8059), the profile set, the env_apply timing
mkdir -p gpurun_out/r6c18
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6c18/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r6c18/gpu_suite.log; tail -4 gpurun_out/r6c18/gpu_suite.log
timeout 600 python tools/stress_cold.py --cold 40 --batches 10 > gpurun_out/r6c18/stress.log 2>&1; echo "rc=$?" >> gpurun_out/r6c18/stress.log
timeout 400 python tools/stress_cold.py --cold 20 --batches 0 --poison 0xA5 > gpurun_out/r6c18/stress_poison.log 2>&1; echo "rc=$?" >> gpurun_out/r6c18/stress_poison.log
tail -3 gpurun_out/r6c18/stress.log; tail -3 gpurun_out/r6c18/stress_poison.log
timeout 1800 python tools/debug_fuzz_campaign.py 8000 60 48 > gpurun_out/r6c18/fuzz.txt 2>&1; tail -3 gpurun_out/r6c18/fuzz.txt | cut -c1-600
bash tools/profile_bench.sh r06 > gpurun_out/r6c18/profile.log 2>&1; tail -2 gpurun_out/r6c18/profile.log | cut -c1-600
timeout 300 python tools/time_env_apply.py 2048 7 > gpurun_out/r6c18/env_apply.txt 2>&1; tail -1 gpurun_out/r6c18/env_apply.txt
