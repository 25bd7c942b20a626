#!/bin/bash
# round 6, call 20: host flattening A/B that survives the box's drift -- eight alternations of (hostr6b, product), 8 repetitions each, medians listed side by side
mkdir -p gpurun_out/r6c20
for i in 1 2 3 4 5 6 7 8; do
  for v in hostr6b product; do if [ $v = product ]; then unset KS_VARIANT; else export KS_VARIANT=$v; fi; echo "$v $(timeout 300 python tools/time_flatten.py 100000 8 2>&1 | tail -1)"; done
done | tee gpurun_out/r6c20/ab.txt
unset KS_VARIANT
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --whatifs 0 --config5-sample 0 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f  ms/step %.2f  p50 %.2f ' % (o['value'], o['ms_per_step'], o['p50_solve_latency_ms']), o['phases_ms_mean'])"; done | tee gpurun_out/r6c20/bench_solve.txt
