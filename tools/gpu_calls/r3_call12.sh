#!/bin/bash
# Round-3 GPU call 12: the fold kernel under KS_CHECK + a poisoned arena, KS_NO_FOLD parity, cold-process stress, phase probes.
export TMPDIR=/tmp
O=gpurun_out
cp karpenter_core_amd/libksolve.so /tmp/keep.so
cp ab/check.so karpenter_core_amd/libksolve.so
KS_POISON=0xA5 timeout 900 python -m pytest tests/test_parity.py tests/test_fuzz.py tests/test_fuzz_mid.py tests/test_scenarios.py tests/test_consolidation.py tests/test_whatif_derived.py -m gpu -x -q --deselect tests/test_parity.py::test_full_size_config4b_replacing_whatifs_match_reference_decisions 2>&1 | tail -5 > $O/v12_check_poison.log
cp /tmp/keep.so karpenter_core_amd/libksolve.so
KS_NO_FOLD=1 timeout 900 python -m pytest tests/test_parity.py tests/test_fuzz_mid.py -m gpu -x -q --deselect tests/test_parity.py::test_full_size_config4b_replacing_whatifs_match_reference_decisions 2>&1 | tail -4 > $O/v12_nofold_parity.log
timeout 600 python tools/stress_cold.py --cold 120 --batches 40 > $O/v12_stress.json 2> $O/v12_stress.err
cp ab/x_probes.so karpenter_core_amd/libksolve.so; python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids > $O/v12_x_probes.log; cp /tmp/keep.so karpenter_core_amd/libksolve.so
cat $O/v12_check_poison.log $O/v12_nofold_parity.log $O/v12_stress.json; tail -4 $O/v12_x_probes.log
