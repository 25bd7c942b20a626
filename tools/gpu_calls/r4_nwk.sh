#!/bin/bash
# round 4 experiment: ks_pack_rr with fewer worker waves (variant builds under karpenter_core_amd/_variants/nwkN), kernel milliseconds on config #3 shapes that fit every variant
mkdir -p gpurun_out/r4ab
for v in ${VARIANTS:-nwk7 nwk3 nwk4}; do
python - $v ${PODS:-50000} >> gpurun_out/r4ab/nwk.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
from karpenter_core_amd import scheduler as S, workloads as W
S._HERE = os.path.join(os.path.dirname(S.__file__), "_variants", sys.argv[1]); S._LIBS = None; S.libs()
for pods in [int(x) for x in sys.argv[2].split(",")]:
    p = W.config3(pods=pods)
    fp = S.FlatProblem(p); fp.upload(0); fp.grid(want_bits=False); fp.solve(decode=False)
    ms = []
    for _ in range(3): fp.solve(decode=False); ms.append(fp.kernel_ms)
    r = fp.solve(); st = r.stats
    print("%s pods %d: min %.2f ms  nodes %d steps %s rounds %s rr %s" % (sys.argv[1], pods, min(ms), len(r.new_nodes), st.get("p23"), st.get("eq_pods"), st.get("p22")))
    fp.close()
PY
done
cat gpurun_out/r4ab/nwk.log
