#!/bin/bash
# usage: tools/pmc_run.sh <outdir-name> <counters...>   (runs tools/phase_profile.py under rocprofv3 --pmc; kernel-trace only)
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
rocprofv3 --kernel-trace --pmc "$@" -d $out -o pmc -- python $GRAFT_REPO_ROOT/tools/phase_profile.py 20000 > $out/run.log 2>&1 || true
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:40]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    print(k, dict(v))
PY
