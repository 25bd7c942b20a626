#!/bin/bash
# Round-3 GPU call 1: new resolver (scalar selection) -- timing A/B, parity, probes, KS_CHECK + poison suite, cold-process stress.
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build()" > $O/v1_build.log 2>&1
for i in 1 2; do python tools/phase_profile.py 2>&1 | grep kernel_ms; done > $O/v1_time_default.log
timeout 900 python -m pytest tests/test_parity.py tests/test_fuzz.py tests/test_fuzz_mid.py -m gpu -x -q 2>&1 | tail -6 > $O/v1_parity.log
bash tools/ab.sh > $O/v1_ab.log 2>&1
cp karpenter_core_amd/libksolve.so /tmp/keep.so
for v in x_p2 x_cut; do cp ab/$v.so karpenter_core_amd/libksolve.so; python tools/p2_probe.py 2>&1 | grep -v amdgpu.ids > $O/v1_$v.log; done
cp ab/x_probes.so karpenter_core_amd/libksolve.so; python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids > $O/v1_x_probes.log
cp ab/check.so karpenter_core_amd/libksolve.so
KS_POISON=0xA5 timeout 900 python -m pytest tests/test_parity.py tests/test_fuzz.py tests/test_fuzz_mid.py tests/test_scenarios.py tests/test_consolidation.py -m gpu -x -q 2>&1 | tail -6 > $O/v1_check_poison.log
cp /tmp/keep.so karpenter_core_amd/libksolve.so
timeout 600 python tools/stress_cold.py --cold 40 --batches 20 > $O/v1_stress.json 2>$O/v1_stress.err
timeout 300 python tools/stress_cold.py --cold 15 --batches 5 --poison 0xA5 > $O/v1_stress_poison.json 2>$O/v1_stress_poison.err
KSH_TIMING=1 python tools/time_from_pods.py 100000 3 > $O/v1_from_pods.log 2>&1
tail -3 $O/v1_time_default.log $O/v1_parity.log $O/v1_ab.log $O/v1_check_poison.log; cat $O/v1_stress.json | head -20
