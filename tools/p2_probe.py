#!/usr/bin/env python3
"""Diagnostics for KS_P2PROBES builds: raw statistics slots of one config #3 Solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
fp = S.FlatProblem(W.config3(pods=int(sys.argv[1]) if len(sys.argv) > 1 else 100000)); fp.upload(0); fp.solve(decode=False); r = fp.solve()
print("kernel_ms", fp.kernel_ms); print(r.stats)
