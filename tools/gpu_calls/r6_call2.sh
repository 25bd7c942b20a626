#!/bin/bash
# round 6, call 2: the two-wave head window against round 5's kernel on one box; where its cycles go; the pack kernel's GPU tests
mkdir -p gpurun_out/r6c2
bash tools/gpu_calls/r6_ab.sh product sinkonly recsink s7f prio product
KS_VARIANT=probes timeout 300 python tools/win_profile.py > gpurun_out/r6c2/win_profile.txt 2>&1; KS_VARIANT=probeswq KS_WQ=1 timeout 300 python tools/win_profile.py >> gpurun_out/r6c2/win_profile.txt 2>&1; cut -c1-200 gpurun_out/r6c2/win_profile.txt | grep -v "^raw"
timeout 1200 python -m pytest tests/test_rr_gpu.py tests/test_parity.py tests/test_fuzz_mid.py tests/test_fuzz.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r6c2/tests.log
