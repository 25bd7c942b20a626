#!/bin/bash
# Round-3 GPU call 7: the ballot builtin re-examined (timing, parity, KS_CHECK + poison), per-phase cycle probes of the current kernel.
export TMPDIR=/tmp
O=gpurun_out
cp karpenter_core_amd/libksolve.so /tmp/keep.so
for i in 1 2; do python tools/phase_profile.py 2>&1 | grep kernel_ms; done > $O/v7_time_default.log
cp ab/t_ballot.so karpenter_core_amd/libksolve.so
for i in 1 2; do python tools/phase_profile.py 2>&1 | grep kernel_ms; done > $O/v7_time_ballot.log
timeout 900 python -m pytest tests/test_parity.py tests/test_fuzz.py tests/test_fuzz_mid.py tests/test_scenarios.py -m gpu -x -q 2>&1 | tail -5 > $O/v7_ballot_parity.log
timeout 300 python tools/stress_cold.py --cold 40 --batches 20 > $O/v7_ballot_stress.json 2> $O/v7_ballot_stress.err
cp ab/t_ballot_check.so karpenter_core_amd/libksolve.so
KS_POISON=0xA5 timeout 900 python -m pytest tests/test_parity.py tests/test_fuzz.py tests/test_fuzz_mid.py tests/test_scenarios.py tests/test_consolidation.py -m gpu -x -q 2>&1 | tail -5 > $O/v7_ballot_check_poison.log
cp /tmp/keep.so karpenter_core_amd/libksolve.so
bash tools/run_p2.sh > /dev/null 2>&1
cp $O/u1_x_probes.log $O/v7_x_probes.log; cp $O/u1_x_p2.log $O/v7_x_p2.log; cp $O/u1_x_cut.log $O/v7_x_cut.log
cp ab/check.so karpenter_core_amd/libksolve.so
KS_POISON=0xA5 timeout 900 python -m pytest tests/test_fuzz_mid.py tests/test_whatif_derived.py tests/test_value_classes.py -m gpu -x -q 2>&1 | tail -5 > $O/v7_check_poison.log
cp /tmp/keep.so karpenter_core_amd/libksolve.so
for f in v7_time_default v7_time_ballot v7_ballot_parity v7_ballot_check_poison v7_check_poison v7_x_probes; do echo "== $f"; cat $O/$f.log; done; cat $O/v7_ballot_stress.json
