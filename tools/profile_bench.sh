#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun): kernel trace + stats, then HBM
# traffic counters in their own passes (never combined with tracing domains other than --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$1; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --whatifs 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python $R/bench.py $ARGS > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
python - <<PY
import csv, glob, collections
for kind in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % kind, recursive=True):
        for row in csv.DictReader(open(f)):
            a = agg[(row["Kernel_Name"][:30], row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()): print(kind, k, "sum", v[0], "dispatches", v[1])
PY
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
