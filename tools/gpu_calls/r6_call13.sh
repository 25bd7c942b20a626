#!/bin/bash
# round 6: the worker's node takes the pod where it is (no hand-over) against the build before; the cycles around the loop
mkdir -p gpurun_out/r6c13
bash tools/gpu_calls/r6_ab.sh product prev product prev
KS_VARIANT=probeswq KS_WQ=1 timeout 300 python tools/win_profile.py > gpurun_out/r6c13/win_profile.txt 2>&1; cut -c1-200 gpurun_out/r6c13/win_profile.txt | grep -v "^raw\|^leader\|^window\|^  rr_window"
timeout 900 python -m pytest tests/test_rr_gpu.py tests/test_fuzz_rr.py tests/test_parity.py -x -q -m gpu 2>&1 | tail -2
