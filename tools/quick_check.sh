#!/bin/bash
# On the GPU box: config #3 kernel time of the shipped library (two runs), then the parity suites that exercise the pack kernel.
#   tools/quick_check.sh [full]     full: every -m gpu test
for i in 1 2; do python tools/phase_profile.py 2>&1 | grep kernel_ms; done
if [ "$1" = full ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
else timeout 1200 python -m pytest tests/test_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -4; fi
