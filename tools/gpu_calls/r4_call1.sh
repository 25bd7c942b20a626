#!/bin/bash
# round 4, call 1: does ks_pack_rr run on the device, is it bit-exact on the config #3 family, how fast is it against ks_pack
mkdir -p gpurun_out/r4c1
cd /root/repo
timeout 600 python -m pytest tests/test_parity.py -m gpu -x -q -k "config1 or config3 or fingerprint" > gpurun_out/r4c1/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/r4c1/parity.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --whatifs 0 > gpurun_out/r4c1/bench_rr.log 2>&1
KS_NO_RR=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --whatifs 0 > gpurun_out/r4c1/bench_norr.log 2>&1
tail -3 gpurun_out/r4c1/parity.log
tail -c 1500 gpurun_out/r4c1/bench_rr.log
