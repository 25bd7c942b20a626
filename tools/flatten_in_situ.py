#!/usr/bin/env python3
"""The host half of a Solve where it runs -- between two pack kernels, not in a loop of its own (tools/time_flatten.py): KSH_TIMING=1 python tools/flatten_in_situ.py
prints the flattening's phases of four consecutive `solve_from_pods` calls on config #3 and each call's timing breakdown.  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
pr = W.config3()
parsed = S.ParsedProblem(pr)
for i in range(4):
    fp, ms = S.solve_from_pods(parsed, 0)
    print("ITER", i, {k: round(v, 2) for k, v in ms.items()}, flush=True)
    sys.stderr.flush()
    fp.close()
