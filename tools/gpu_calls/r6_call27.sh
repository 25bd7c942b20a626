#!/bin/bash
# round 6, call 27: 4 000 more fresh fuzz problems on the final tree (seeds 50000-50799)
mkdir -p gpurun_out/r6c27
timeout 3000 python tools/debug_fuzz_campaign.py 50000 800 48 > gpurun_out/r6c27/fuzz_50000.txt 2>&1; tail -1 gpurun_out/r6c27/fuzz_50000.txt | cut -c1-700
