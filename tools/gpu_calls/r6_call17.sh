#!/bin/bash
# round 6, call 17: host flattening with the spec confirmation beside it too (product) against the library before this session (hostprev); the phases of a continued snapshot
mkdir -p gpurun_out/r6c17
for v in hostprev product hostprev product; do echo "== host library: $v"; if [ $v = product ]; then unset KS_VARIANT; else export KS_VARIANT=$v; fi; KSH_TIMING=1 timeout 300 python tools/time_flatten.py 100000 8 2>&1 | tail -17 | grep -v "uid table"; done > gpurun_out/r6c17/flatten.txt 2>&1; unset KS_VARIANT; grep "==\|flatten ms" gpurun_out/r6c17/flatten.txt; tail -17 gpurun_out/r6c17/flatten.txt
KSH_TIMING=1 timeout 300 python tools/time_env_apply.py 2048 4 > gpurun_out/r6c17/apply.txt 2>&1; grep -A24 "apply 3" gpurun_out/r6c17/apply.txt | head -30; tail -1 gpurun_out/r6c17/apply.txt
timeout 300 python tools/step_gap.py 100000 10
