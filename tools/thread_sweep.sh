#!/bin/bash
# On the GPU box: Solve-from-pods timing (tools/time_from_pods.py) for several host thread counts (KSH_THREADS) -- how the flattening scales.
for t in 8 16 32 64; do echo "threads $t"; KSH_THREADS=$t python tools/time_from_pods.py 100000 5 2>&1 | grep -E "flatten|total_ms|decisions"; done
