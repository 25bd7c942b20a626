#!/bin/bash
# Round-3 GPU call 9: whole GPU suite, host flattening with the environment cache (phases), sequential-path counters, bench line.
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_parity.py::test_full_size_config4b_replacing_whatifs_match_reference_decisions 2>&1 | tail -6 > $O/v9_suite.log
KSH_TIMING=1 python tools/time_from_pods.py 100000 4 > $O/v9_from_pods.log 2>&1
KSH_NO_ENV_CACHE=1 python tools/time_from_pods.py 100000 4 > $O/v9_from_pods_nocache.log 2>&1
cp karpenter_core_amd/libksolve.so /tmp/keep.so; cp ab/x_cut.so karpenter_core_amd/libksolve.so; python tools/p2_probe.py > $O/v9_x_cut.log 2>&1; cp /tmp/keep.so karpenter_core_amd/libksolve.so
timeout 900 python bench.py --steps 5 --warmup 1 --cpu-sample-only > $O/v9_bench.json 2> $O/v9_bench.err
cat $O/v9_suite.log; grep -E "median|decisions" $O/v9_from_pods.log; grep -E "median|decisions" $O/v9_from_pods_nocache.log; grep -v amdgpu $O/v9_x_cut.log | tail -1
python -c "
import json; d=json.load(open('$O/v9_bench.json')); print(d['value'], d['p50_solve_latency_ms'], d['phases_ms_mean']); print(d['ingress']); print(d['roofline']['traffic'], d['roofline'].get('issue'))"
