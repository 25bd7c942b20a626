#!/bin/bash
# round 6, call 8: the contract line's Solve leg alone (host teardown off the caller's thread), twice; the old host library beside it
mkdir -p gpurun_out/r6c8
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --whatifs 0 --config5-sample 0 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f  ms/step %.2f  p50 %.2f ' % (o['value'], o['ms_per_step'], o['p50_solve_latency_ms']), o['phases_ms_mean'])"; done | tee gpurun_out/r6c8/bench_solve.txt
