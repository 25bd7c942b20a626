#!/bin/bash
# round 6, call 6: machines' loads in one trip (against the build before), questions to the workers for dyn1 pods again (RR_WIN_DYNQ), where the cycles go around the loop
mkdir -p gpurun_out/r6c6
bash tools/gpu_calls/r6_ab.sh product prevopen dynq product
KS_VARIANT=probeswq KS_WQ=1 timeout 300 python tools/win_profile.py > gpurun_out/r6c6/win_profile.txt 2>&1; cut -c1-200 gpurun_out/r6c6/win_profile.txt | grep -v "^raw\|^leader\|^window\|^  rr_window"
timeout 900 python -m pytest tests/test_rr_gpu.py tests/test_parity.py -x -q -m gpu 2>&1 | tail -2
