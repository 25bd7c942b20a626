#!/bin/bash
# Round-3 GPU call 11: node-opening pods folded into the rounds -- timing A/B (KS_NO_FOLD=1), parity, counters.
export TMPDIR=/tmp
O=gpurun_out
for i in 1 2; do python tools/phase_profile.py 2>&1 | grep kernel_ms; done > $O/v11_time_fold.log
for i in 1 2; do KS_NO_FOLD=1 python tools/phase_profile.py 2>&1 | grep kernel_ms; done > $O/v11_time_nofold.log
timeout 1200 python -m pytest tests/test_parity.py tests/test_fuzz_mid.py tests/test_fuzz.py tests/test_scenarios.py tests/test_value_classes.py -m gpu -q -x --deselect tests/test_parity.py::test_full_size_config4b_replacing_whatifs_match_reference_decisions 2>&1 | tail -6 > $O/v11_parity.log
cp karpenter_core_amd/libksolve.so /tmp/keep.so; cp ab/x_cut.so karpenter_core_amd/libksolve.so; python tools/p2_probe.py > $O/v11_x_cut.log 2>&1; cp /tmp/keep.so karpenter_core_amd/libksolve.so
cat $O/v11_time_fold.log $O/v11_time_nofold.log $O/v11_parity.log; grep -v amdgpu $O/v11_x_cut.log | tail -1
