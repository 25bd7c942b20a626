#!/bin/bash
# Round-3 GPU call 10: BASELINE configs[4] at its stated size through bench.py --config5 (one Solve, roofline from a statistics launch).
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python bench.py --config5 1000000 --steps 1 --warmup 0 > $O/v10_config5_1m.json 2> $O/v10_config5_1m.err
tail -c 2500 $O/v10_config5_1m.json; tail -3 $O/v10_config5_1m.err
