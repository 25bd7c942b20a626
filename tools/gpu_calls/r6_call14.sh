#!/bin/bash
# round 6, call 14: what a bench step spends outside the library's own total; the host flattening at 32 / 64 / 128 threads
mkdir -p gpurun_out/r6c14
( timeout 300 python tools/step_gap.py 100000 8
  for t in 64 128; do KSH_THREADS=$t timeout 300 python tools/step_gap.py 100000 8; done
  for t in 32 64 128; do echo "== KSH_THREADS=$t"; KSH_THREADS=$t KSH_TIMING=1 timeout 300 python tools/time_flatten.py 100000 6 2>&1 | tail -14; done ) > gpurun_out/r6c14/gap.txt 2>&1
cat gpurun_out/r6c14/gap.txt
