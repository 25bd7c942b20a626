#!/bin/bash
# round 6, call 5: lend width / spare places; where the cycles go; the host flattening before and after (old library as a variant)
mkdir -p gpurun_out/r6c5
bash tools/gpu_calls/r6_ab.sh product lpw7 lpw6 lpw5 product
KS_VARIANT=probes timeout 300 python tools/win_profile.py > gpurun_out/r6c5/win_profile.txt 2>&1; KS_VARIANT=probeswq KS_WQ=1 timeout 300 python tools/win_profile.py >> gpurun_out/r6c5/win_profile.txt 2>&1; cut -c1-200 gpurun_out/r6c5/win_profile.txt | grep -v "^raw\|^leader"
for v in hostold product hostold product; do echo "== host library: $v"; if [ $v = product ]; then unset KS_VARIANT; else export KS_VARIANT=$v; fi; KSH_TIMING=1 python tools/time_flatten.py 100000 6 2>&1 | tail -15 | grep -v "uid table\|pass B\|chains"; done > gpurun_out/r6c5/flatten.txt 2>&1; unset KS_VARIANT; cat gpurun_out/r6c5/flatten.txt
