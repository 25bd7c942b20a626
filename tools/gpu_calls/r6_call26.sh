#!/bin/bash
# round 6, call 26: the GPU minutes that are left go to the randomised net -- 8 000 more fresh problems (seeds 40000-41599, five families) on the final tree
mkdir -p gpurun_out/r6c26
timeout 5000 python tools/debug_fuzz_campaign.py 40000 1600 48 > gpurun_out/r6c26/fuzz_40000.txt 2>&1; tail -1 gpurun_out/r6c26/fuzz_40000.txt | cut -c1-700
