#!/usr/bin/env python3
"""Timing of the from-the-pod-list path (scheduler.solve_from_pods) on config #3: usage tools/time_from_pods.py [pods] [reps]
   env KSH_TIMING=1 prints the host flattening's phases."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pp = S.ParsedProblem(W.config3(pods=pods))
rows = []
for i in range(reps):
    fp, ms = S.solve_from_pods(pp, 0)
    rows.append(ms)
    if i == 0:
        print("nodes", len(fp.result().new_nodes))
    fp.close()
for k in S.TIMING_KEYS:
    print(f"{k:20s} median {statistics.median(r[k] for r in rows):9.3f} ms   first {rows[0][k]:9.3f} ms")
print("decisions/s (median total):", pods / (statistics.median(r['total_ms'] for r in rows) / 1e3))
