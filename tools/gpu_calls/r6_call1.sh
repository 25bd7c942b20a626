#!/bin/bash
# round 6, call 1: the tree as round 5 left it -- whole GPU suite, the contract line, the two-wave pipeline's microbenchmarks, config #5 at its stated size
mkdir -p gpurun_out/r6c1
tools/ubench/pipe_costs > gpurun_out/r6c1/pipe_costs.txt 2>&1; cat gpurun_out/r6c1/pipe_costs.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6c1/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r6c1/gpu_suite.log
tail -3 gpurun_out/r6c1/gpu_suite.log
timeout 900 python bench.py > gpurun_out/r6c1/bench.json 2> gpurun_out/r6c1/bench.err; tail -c 1500 gpurun_out/r6c1/bench.json | head -c 400; echo
timeout 1200 python bench.py --config5 1000000 --steps 1 --warmup 0 > gpurun_out/r6c1/config5_1m.json 2> gpurun_out/r6c1/config5_1m.err; head -c 600 gpurun_out/r6c1/config5_1m.json; echo
