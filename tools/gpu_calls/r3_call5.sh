#!/bin/bash
# Round-3 GPU call 5: what-ifs derived on the device.
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_whatif_derived.py tests/test_whatif_flattening.py tests/test_consolidation.py tests/test_distributed.py -m gpu -q -x 2>&1 | tail -15 > $O/v5_tests.log
KSH_TIMING=1 timeout 600 python bench.py --whatifs-only > $O/v5_whatifs.json 2> $O/v5_whatifs.err
cat $O/v5_tests.log; python -c "
import json; d=json.load(open('$O/v5_whatifs.json'))
for k in ('first_batch_over_the_snapshot','end_to_end','resident'): print(k, {a:b for a,b in d[k].items() if a!='what'})"
tail -5 $O/v5_whatifs.err
