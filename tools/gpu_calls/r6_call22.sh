#!/bin/bash
# round 6, call 22: after the no_T fix in ks_pack_rr (+ the barrier in ks_pack's failure path) -- the GPU suite, the campaign that found the bug again (seeds 9000-9119) and 1 000 more
# problems (seeds 10000-10199), the A/B of the kernel before / after on the bench problem, then the profile set re-keyed to the new kernel sources
mkdir -p gpurun_out/r6c22
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6c22/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r6c22/gpu_suite.log; tail -4 gpurun_out/r6c22/gpu_suite.log
timeout 2400 python tools/debug_fuzz_campaign.py 9000 120 48 > gpurun_out/r6c22/fuzz_9000.txt 2>&1; tail -1 gpurun_out/r6c22/fuzz_9000.txt | cut -c1-600
timeout 3000 python tools/debug_fuzz_campaign.py 10000 200 48 > gpurun_out/r6c22/fuzz_10000.txt 2>&1; tail -1 gpurun_out/r6c22/fuzz_10000.txt | cut -c1-600
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --whatifs 0 --config5-sample 0 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f  ms/step %.2f  p50 %.2f ' % (o['value'], o['ms_per_step'], o['p50_solve_latency_ms']), o['phases_ms_mean'])"; done | tee gpurun_out/r6c22/bench_solve.txt
bash tools/profile_bench.sh r06 > gpurun_out/r6c22/profile.log 2>&1; tail -1 gpurun_out/r6c22/profile.log | cut -c1-300
timeout 600 python tools/stress_cold.py --cold 40 --batches 10 > gpurun_out/r6c22/stress.log 2>&1; echo "rc=$?" >> gpurun_out/r6c22/stress.log; tail -2 gpurun_out/r6c22/stress.log
