#!/bin/bash
# Round-3 GPU call 3: the resolver fix -- mid-scale fuzz, whole GPU suite, timing, bench line with the ingress leg.
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_fuzz_mid.py -m gpu -q 2>&1 | tail -8 > $O/v3_mid.log
for i in 1 2; do python tools/phase_profile.py 2>&1 | grep kernel_ms; done > $O/v3_time.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/v3_suite.log
timeout 900 python bench.py --steps 5 --warmup 1 > $O/v3_bench.json 2> $O/v3_bench.err
cat $O/v3_mid.log $O/v3_time.log $O/v3_suite.log; python -c "
import json; d=json.load(open('$O/v3_bench.json')); print(d['value'], d['p50_solve_latency_ms'], d['phases_ms_mean']); print(d['ingress']); print({k:v for k,v in d['whatif_batch'].items() if not isinstance(v,(dict,list))})"
