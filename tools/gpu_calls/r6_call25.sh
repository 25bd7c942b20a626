#!/bin/bash
# round 6, call 25: the FINAL tree (the barrier in ks_pack_rr's relaxation path is the last kernel edit) -- GPU suite, 600 fresh fuzz problems (seeds 30000-30119), the profile set re-keyed to
# the kernel sources as they stand, then the contract line once more with its roofline filled from those passes
mkdir -p gpurun_out/r6c25
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6c25/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r6c25/gpu_suite.log; tail -4 gpurun_out/r6c25/gpu_suite.log
timeout 2400 python tools/debug_fuzz_campaign.py 30000 120 48 > gpurun_out/r6c25/fuzz_30000.txt 2>&1; tail -1 gpurun_out/r6c25/fuzz_30000.txt | cut -c1-600
bash tools/profile_bench.sh r06 > gpurun_out/r6c25/profile.log 2>&1; tail -1 gpurun_out/r6c25/profile.log | cut -c1-200
python tools/summarize_profile.py gpurun_out/prof_r06 r06 > gpurun_out/r6c25/summarize.log 2>&1
timeout 900 python bench.py > gpurun_out/r6c25/bench.json 2> gpurun_out/r6c25/bench.err; python - <<'P'
import json
d = json.loads(open("gpurun_out/r6c25/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d["p50_solve_latency_ms"], d["phases_ms_mean"], "traffic", d["roofline"]["traffic"], "issue", (d["roofline"].get("issue") or {}).get("frac"))
P
timeout 400 python tools/stress_cold.py --cold 20 --batches 0 --poison 0xA5 > gpurun_out/r6c25/stress_poison.log 2>&1; echo "rc=$?" >> gpurun_out/r6c25/stress_poison.log; tail -2 gpurun_out/r6c25/stress_poison.log
