#!/bin/bash
# round 6, call 15: the snapshot kept current by events (ksh_env_apply) -- the GPU half of its tests, the what-if leg with the one-node change, the host timing on this box
mkdir -p gpurun_out/r6c15
timeout 900 python -m pytest tests/test_env_apply.py tests/test_whatif_derived.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python tools/time_env_apply.py 2048 5 2>&1 | tail -2
KSH_TIMING=1 timeout 600 python bench.py --whatifs-only > gpurun_out/r6c15/whatifs.json 2> gpurun_out/r6c15/whatifs.err; tail -c 2500 gpurun_out/r6c15/whatifs.json; grep "derived what-ifs\|what-ifs:" gpurun_out/r6c15/whatifs.err | tail -30
