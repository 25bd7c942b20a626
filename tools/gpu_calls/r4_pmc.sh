#!/bin/bash
# round 4: instruction / wait / instruction-cache counters of ks_pack_rr on config #3 (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; out=$R/gpurun_out/r4pmc/$name; mkdir -p $out; rocprofv3 --kernel-trace --pmc "$@" -d $out -o pmc -- python $R/tools/phase_profile_rr.py 100000 > $out/run.log 2>&1 || true; python $R/tools/read_pmc.py $out ks_pack_rr > $out/summary.txt 2>&1; cat $out/summary.txt; }
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
run insts2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT
run waits SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
run waits2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
tail -3 $R/gpurun_out/r4pmc/insts/run.log
