#!/bin/bash
# round 4: where a round / a run step of ks_pack_rr spends its cycles (KS_PROBES build, KS_RR=1)
mkdir -p gpurun_out/r4c4
python tools/phase_profile_rr.py 100000 > gpurun_out/r4c4/rr_phase_100k.log 2>&1
cat gpurun_out/r4c4/rr_phase_100k.log
