#!/bin/bash
# round 6, call 4: the census of zero counters against the kernel without it; where the cycles go now; a fresh fuzz campaign on the new window; the host flattening's laps
mkdir -p gpurun_out/r6c4
bash tools/gpu_calls/r6_ab.sh product nohz product
KS_VARIANT=probes timeout 300 python tools/win_profile.py > gpurun_out/r6c4/win_profile.txt 2>&1; KS_VARIANT=probeswq KS_WQ=1 timeout 300 python tools/win_profile.py >> gpurun_out/r6c4/win_profile.txt 2>&1; cut -c1-200 gpurun_out/r6c4/win_profile.txt | grep -v "^raw"
timeout 1500 python tools/debug_fuzz_campaign.py 6000 30 48 > gpurun_out/r6c4/fuzz.txt 2>&1; tail -3 gpurun_out/r6c4/fuzz.txt | cut -c1-600
KSH_TIMING=1 python tools/time_flatten.py 100000 4 > gpurun_out/r6c4/flatten.txt 2>&1; tail -24 gpurun_out/r6c4/flatten.txt
KSH_TIMING=1 python tools/time_from_pods.py 100000 3 2>&1 | grep "solve_from\|median" | tail -12
