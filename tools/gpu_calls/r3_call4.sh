#!/bin/bash
# Round-3 GPU call 4: widened mid-scale fuzz (36 seeds), the long cold-process stress run.
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_fuzz_mid.py tests/test_ingress.py -m gpu -q 2>&1 | tail -12 > $O/v4_mid.log
timeout 1500 python tools/stress_cold.py --cold 400 --batches 150 > $O/v4_stress.json 2> $O/v4_stress.err
timeout 600 python tools/stress_cold.py --cold 100 --batches 50 --poison 0xA5 > $O/v4_stress_poison.json 2> $O/v4_stress_poison.err
cat $O/v4_mid.log; cat $O/v4_stress.json $O/v4_stress_poison.json
