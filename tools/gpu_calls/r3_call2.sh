#!/bin/bash
# Round-3 GPU call 2: which variants fail the mid-scale fuzz, and where.
export TMPDIR=/tmp
O=gpurun_out
python tools/debug_mid.py 0 3 4 6 > $O/v2_diag_default.log 2>&1
KS_ONE_WAVE=1 python tools/debug_mid.py 0 3 4 6 > $O/v2_diag_onewave.log 2>&1
KS_NO_LEAN=1 python tools/debug_mid.py 0 3 4 6 > $O/v2_diag_nolean.log 2>&1
KS_NO_DYN=1 python tools/debug_mid.py 0 3 4 6 > $O/v2_diag_nodyn.log 2>&1
cp karpenter_core_amd/libksolve.so /tmp/keep.so
cp ab/t_nofast.so karpenter_core_amd/libksolve.so; python tools/debug_mid.py 0 3 4 6 > $O/v2_diag_nofast.log 2>&1
cp /tmp/keep.so karpenter_core_amd/libksolve.so
timeout 900 python -m pytest tests/test_fuzz_mid.py -m gpu -q 2>&1 | tail -15 > $O/v2_mid_all.log
timeout 600 python -m pytest tests/test_cabi.py tests/test_parity.py -m gpu -q -x 2>&1 | tail -5 > $O/v2_parity.log
for f in $O/v2_diag_*.log; do echo "== $f"; grep -v amdgpu.ids $f | head -30; done
cat $O/v2_mid_all.log
