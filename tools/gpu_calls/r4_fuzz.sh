#!/bin/bash
# round 4: fresh-seed fuzz campaign on the final kernels (ks_pack_rr takes what it covers, ks_pack the rest), GPU == oracle, seeds 3000-3059 x 4 families
mkdir -p gpurun_out/r4fuzz
timeout 420 python tools/debug_fuzz_campaign.py 3000 60 64 > gpurun_out/r4fuzz/campaign.log 2>&1; echo "rc=$?" >> gpurun_out/r4fuzz/campaign.log
tail -5 gpurun_out/r4fuzz/campaign.log
