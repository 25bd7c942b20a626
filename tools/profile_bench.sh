#!/bin/bash
# rocprofv3 evidence for bench.py, run on the GPU box through gpurun:   tools/profile_bench.sh <tag>      (tag e.g. r02)
#   1. the contract command itself (`python bench.py`, default arguments) -> the JSON line
#   2. the same command under `rocprofv3 --kernel-trace --stats`          -> per-kernel durations
#   3. counters in their OWN passes (kernel-trace only, never with other tracing domains): FETCH_SIZE | WRITE_SIZE | SQ instruction mix,
#      on the Solve leg only (`--no-cpu-baseline --whatifs 0`: same pack kernel, same problem)
# then tools/summarize_profile.py gpurun_out/prof_<tag> <tag> (here, in the build container) writes the files kept under profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$1; rm -rf $OUT; mkdir -p $OUT
SOLVE="--steps 3 --warmup 1 --no-cpu-baseline --whatifs 0"
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
# (the same GPU work; the CPU oracle's full-size run -- two minutes of one host core -- is not repeated under the profiler)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --cpu-sample-only > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python $R/bench.py $SOLVE > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python $R/bench.py $SOLVE > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH --output-format csv -d $OUT/insts -o bench -- python $R/bench.py $SOLVE > $OUT/bench_insts.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/cycles -o bench -- python $R/bench.py $SOLVE > $OUT/bench_cycles.log 2>&1
# the same four passes for the two other dominant kernels: the single-wave batch kernel of the 512 what-ifs, the general 4-wave kernel of config #5 at 250 000 pods
WI="--whatifs-only"; C5="--config5 250000 --steps 1 --warmup 0"
for leg in wi c5; do
  if [ $leg = wi ]; then A="$WI"; else A="$C5"; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$leg -o bench -- python $R/bench.py $A > $OUT/bench_fetch_$leg.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$leg -o bench -- python $R/bench.py $A > $OUT/bench_write_$leg.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH --output-format csv -d $OUT/insts_$leg -o bench -- python $R/bench.py $A > $OUT/bench_insts_$leg.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/cycles_$leg -o bench -- python $R/bench.py $A > $OUT/bench_cycles_$leg.log 2>&1
done
# keep only what the summariser reads (gpurun_out is capped at 64 MiB)
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT; tail -c 600 $OUT/bench.json
