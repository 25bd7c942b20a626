#!/bin/bash
# round 6, call 21: the final tree once more -- the contract line with the counter passes of call 18 behind its roofline (same kernel sources), 600 more fresh fuzz problems, smoke
mkdir -p gpurun_out/r6c21
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r6c21/bench.json 2> gpurun_out/r6c21/bench.err; python - <<'P'
import json
d = json.loads(open("gpurun_out/r6c21/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d["p50_solve_latency_ms"], d["phases_ms_mean"], "traffic", d["roofline"]["traffic"], "issue", (d["roofline"].get("issue") or {}).get("frac"))
print("whatif traffic", d["whatif_batch"]["roofline"]["traffic"], "config5 traffic", d["config5"].get("roofline", {}).get("traffic"))
print("after change", {k: v for k, v in d["whatif_batch"]["after_a_one_node_change"].items() if k not in ("runs", "what")})
P
timeout 2400 python tools/debug_fuzz_campaign.py 9000 120 48 > gpurun_out/r6c21/fuzz.txt 2>&1; tail -2 gpurun_out/r6c21/fuzz.txt | cut -c1-600
