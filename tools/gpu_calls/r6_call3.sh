#!/bin/bash
# round 6, call 3: the prepared-pod window as the product -- the whole GPU suite, the contract line, a few knobs
mkdir -p gpurun_out/r6c3
bash tools/gpu_calls/r6_ab.sh product slp2 slp8 prio slp2prio product
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6c3/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r6c3/gpu_suite.log; tail -3 gpurun_out/r6c3/gpu_suite.log
timeout 900 python bench.py > gpurun_out/r6c3/bench.json 2> gpurun_out/r6c3/bench.err; python - <<'PY'
import json
o=json.loads([l for l in open('gpurun_out/r6c3/bench.json') if l.startswith('{')][-1])
print(o['value'], o['ms_per_step'], o['phases_ms_mean'])
PY
