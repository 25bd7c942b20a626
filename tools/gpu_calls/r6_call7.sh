#!/bin/bash
# round 6, call 7: the record of the round on the tree as it stands -- whole GPU suite, cold-process stress (plain and poisoned), a fresh 240-problem fuzz campaign,
# the profile set (tools/profile_bench.sh r06: contract line, kernel stats, counter passes for the three dominant kernels), config #5 at its stated size
mkdir -p gpurun_out/r6c7
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6c7/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r6c7/gpu_suite.log; tail -4 gpurun_out/r6c7/gpu_suite.log
timeout 600 python tools/stress_cold.py --cold 40 --batches 10 > gpurun_out/r6c7/stress.log 2>&1; echo "rc=$?" >> gpurun_out/r6c7/stress.log
timeout 400 python tools/stress_cold.py --cold 20 --batches 0 --poison 0xA5 > gpurun_out/r6c7/stress_poison.log 2>&1; echo "rc=$?" >> gpurun_out/r6c7/stress_poison.log
tail -3 gpurun_out/r6c7/stress.log; tail -3 gpurun_out/r6c7/stress_poison.log
timeout 1800 python tools/debug_fuzz_campaign.py 7000 60 48 > gpurun_out/r6c7/fuzz.txt 2>&1; tail -3 gpurun_out/r6c7/fuzz.txt | cut -c1-600
bash tools/profile_bench.sh r06 > gpurun_out/r6c7/profile.log 2>&1; tail -2 gpurun_out/r6c7/profile.log | cut -c1-600
timeout 1200 python bench.py --config5 1000000 --steps 1 --warmup 0 > gpurun_out/r6c7/config5_1m.json 2> gpurun_out/r6c7/config5_1m.err; head -c 400 gpurun_out/r6c7/config5_1m.json; echo
