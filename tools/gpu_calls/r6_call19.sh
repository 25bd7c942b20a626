#!/bin/bash
# round 6, call 19: the host flattening with the topology terms' pure half on the pool (product) against the record call's library (hostr6b) and the session's first (hostprev), interleaved; the bench's Solve leg
mkdir -p gpurun_out/r6c19
for v in hostr6b product hostprev hostr6b product; do echo "== host library: $v"; if [ $v = product ]; then unset KS_VARIANT; else export KS_VARIANT=$v; fi; KSH_TIMING=1 timeout 300 python tools/time_flatten.py 100000 10 2>&1 | tail -17 | grep -v "uid table"; done > gpurun_out/r6c19/flatten.txt 2>&1; unset KS_VARIANT; grep "==\|flatten ms" gpurun_out/r6c19/flatten.txt; tail -16 gpurun_out/r6c19/flatten.txt
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --whatifs 0 --config5-sample 0 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f  ms/step %.2f  p50 %.2f ' % (o['value'], o['ms_per_step'], o['p50_solve_latency_ms']), o['phases_ms_mean'])"; done | tee gpurun_out/r6c19/bench_solve.txt
