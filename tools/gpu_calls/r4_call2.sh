#!/bin/bash
# round 4, call 2: where a round of ks_pack_rr spends its cycles (KS_PROBES build shipped in the tree)
mkdir -p gpurun_out/r4c2
python tools/phase_profile_rr.py 100000 > gpurun_out/r4c2/rr_phase_100k.log 2>&1
python tools/phase_profile_rr.py 10000 > gpurun_out/r4c2/rr_phase_10k.log 2>&1
cat gpurun_out/r4c2/rr_phase_100k.log gpurun_out/r4c2/rr_phase_10k.log
