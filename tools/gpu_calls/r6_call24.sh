#!/bin/bash
# round 6, call 24: bug hunting after the fix -- 4 000 more fresh fuzz problems (seeds 20000-20799, five families); then the contract line once more (its roofline filled from the counter passes of call 22: same kernel sources)
mkdir -p gpurun_out/r6c24
timeout 4000 python tools/debug_fuzz_campaign.py 20000 800 48 > gpurun_out/r6c24/fuzz_20000.txt 2>&1; tail -1 gpurun_out/r6c24/fuzz_20000.txt | cut -c1-700
timeout 900 python bench.py > gpurun_out/r6c24/bench.json 2> gpurun_out/r6c24/bench.err; python - <<'P'
import json
d = json.loads(open("gpurun_out/r6c24/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d["p50_solve_latency_ms"], d["phases_ms_mean"], "traffic", d["roofline"]["traffic"], "issue", (d["roofline"].get("issue") or {}).get("frac"))
P
