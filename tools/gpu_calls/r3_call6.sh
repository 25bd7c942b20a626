#!/bin/bash
# Round-3 GPU call 6: value classes on the reference fixture, derived what-ifs timing (cold / warm phases).
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_value_classes.py tests/test_whatif_derived.py tests/test_scenarios.py -m gpu -q -x 2>&1 | tail -8 > $O/v6_tests.log
KSH_TIMING=1 timeout 600 python bench.py --whatifs-only > $O/v6_whatifs.json 2> $O/v6_whatifs.err
cat $O/v6_tests.log; python -c "
import json; d=json.load(open('$O/v6_whatifs.json'))
for k in ('first_batch_over_the_snapshot','end_to_end','resident'): print(k, {a:b for a,b in d[k].items() if a!='what'})"
grep "derived what-ifs" $O/v6_whatifs.err | head -24
