cp karpenter_core_amd/libksolve.so /tmp/keep.so
cp ab/x_check.so karpenter_core_amd/libksolve.so
timeout 1200 python -m pytest tests/test_fuzz.py tests/test_parity.py tests/test_scenarios.py -m gpu -x -q 2>&1 | tail -4
cp /tmp/keep.so karpenter_core_amd/libksolve.so
