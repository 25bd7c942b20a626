#!/bin/bash
# For the next round, first GPU call: price the two leads of DESIGN.md 8b before rewriting the pack kernel -- a commit without control flow, hostname items as per-counter
# slot masks -- on the synthetic loop (tools/ubench/lean_loop.hip; build the variants first, in the build container: tools/ubench/build_lean.sh).  Half a GPU-minute.
mkdir -p gpurun_out/r5ub
for b in tools/ubench/lean_loop_??; do timeout 20 $b | tail -1; done > gpurun_out/r5ub/lean_loop.txt 2>&1
timeout 30 tools/ubench/wave_costs > gpurun_out/r5ub/wave_costs.txt 2>&1
cat gpurun_out/r5ub/lean_loop.txt
