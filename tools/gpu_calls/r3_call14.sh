#!/bin/bash
# Round-3 GPU call 14: config #4b (replacing what-ifs) against the reference's decisions, derived what-if fuzz, the C99 usage program on a GPU.
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_parity.py::test_full_size_config4b_replacing_whatifs_match_reference_decisions tests/test_whatif_derived.py tests/test_cabi.py tests/test_env_cache.py -m gpu -q 2>&1 | tail -25 > $O/v14_tests.log
timeout 300 python -m pytest tests/test_cabi.py -q -k plain_c 2>&1 | tail -3 >> $O/v14_tests.log
cat $O/v14_tests.log
