#!/usr/bin/env python3
"""Where a bench step's wall time goes OUTSIDE the library's own total (ms[5]): the ctypes call, the handle's teardown.   usage: tools/step_gap.py [pods] [reps]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from karpenter_core_amd import scheduler as S, workloads as W
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pp = S.ParsedProblem(W.config3(pods=pods, seed=44))
rows = []
for i in range(reps + 2):
    t0 = time.perf_counter(); fp, ms = S.solve_from_pods(pp, 0); t1 = time.perf_counter(); fp.close(); t2 = time.perf_counter()
    if i >= 2: rows.append(((t1 - t0) * 1e3, ms["total_ms"], (t2 - t1) * 1e3, ms["flatten_ms"], ms["pack_kernel_ms"], ms["solve_readback_ms"]))
for name, col in zip(("call wall", "library total", "close()", "flatten", "kernel", "solve+readback"), zip(*rows)):
    print("%-16s median %8.2f  min %8.2f  max %8.2f ms" % (name, statistics.median(col), min(col), max(col)))
print("threads", os.environ.get("KSH_THREADS", "default"))
