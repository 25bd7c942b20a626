#!/bin/bash
# Round-3 GPU call 13: the general-kernel mid-scale family (seeds 36-47), derived what-ifs with host ports, env-cache GPU tests.
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_fuzz_mid.py tests/test_whatif_derived.py tests/test_ingress.py -m gpu -q 2>&1 | tail -12 > $O/v13_tests.log
cat $O/v13_tests.log
