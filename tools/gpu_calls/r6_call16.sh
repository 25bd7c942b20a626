#!/bin/bash
# round 6, call 16: the host flattening with the queue sort beside the classing (product) against the library before (hostprev), interleaved; then the bench line
mkdir -p gpurun_out/r6c16
for v in hostprev product hostprev product; do echo "== host library: $v"; if [ $v = product ]; then unset KS_VARIANT; else export KS_VARIANT=$v; fi; KSH_TIMING=1 timeout 300 python tools/time_flatten.py 100000 8 2>&1 | tail -14 | grep -v "uid table\|pass B"; done > gpurun_out/r6c16/flatten.txt 2>&1; unset KS_VARIANT; grep "==\|flatten ms\|batch over\|dedupe +" gpurun_out/r6c16/flatten.txt
KSH_SYNC_QUEUE_SORT=1 timeout 300 python tools/time_flatten.py 100000 8 2>&1 | tail -1
KSH_THREADS=32 timeout 300 python tools/time_flatten.py 100000 8 2>&1 | tail -1
timeout 300 python tools/step_gap.py 100000 10
timeout 900 python bench.py > gpurun_out/r6c16/bench.json 2> gpurun_out/r6c16/bench.err; python - <<'P'
import json
d = json.loads(open("gpurun_out/r6c16/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "p50", d["p50_solve_latency_ms"], d["phases_ms_mean"])
print("whatif after change", {k: v for k, v in d["whatif_batch"]["after_a_one_node_change"].items() if k not in ("runs", "what")})
P
