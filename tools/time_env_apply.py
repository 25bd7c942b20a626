#!/usr/bin/env python3
"""what the snapshot's flattening costs after a one-node change, continued (ksh_env_apply) against from scratch -- host only, no GPU needed.
   usage: tools/time_env_apply.py [nodes] [reps]      env KSH_TIMING=1 prints the phases"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from karpenter_core_amd import scheduler as S, workloads as W
nn = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
its, prov, nodes, bound = W.cluster_snapshot(nn, 50, 45)
snap, pn = W.snapshot_problem(its, prov, nodes, bound, False)
parsed = S.ParsedProblem(snap)
t = time.perf_counter(); parsed.snapshot_fingerprint(pn); cold0 = (time.perf_counter() - t) * 1e3
from test_env_apply import new_node
rs = np.random.RandomState(3)
warm, cold = [], []
for r in range(reps):
    name = f"late-{r}"
    ev = [("node+", new_node(its, name, rs))] + [("bind", name, W.generic_pod(rs, f"late-{r}-{k}")) for k in range(20)]
    if os.environ.get("KSH_TIMING"): sys.stderr.write(f"--- apply {r}\n")
    info = parsed.apply(ev, pn if r == 0 else None)
    assert info["continued"], info
    warm.append(info["ms"])
    t = time.perf_counter(); parsed.snapshot_fingerprint(cold=True); cold.append((time.perf_counter() - t) * 1e3)
print(f"snapshot: {nn} nodes, {len(pn)} bound pods | first flattening {cold0:.1f} ms | after one node + 20 pods: continued {statistics.median(warm):.2f} ms (min {min(warm):.2f}), "
      f"from scratch + hash {statistics.median(cold):.1f} ms")
