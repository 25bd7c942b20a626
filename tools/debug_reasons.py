#!/usr/bin/env python3
"""Diagnostics: per-pod failure reasons of one fuzz case, GPU vs oracle (tests import path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz as F
from karpenter_core_amd import scheduler as S
seed = int(sys.argv[1])
p = F.fuzz_problem(seed)
ref = F.O.solve(p)
got = S.solve_problem(p)
print("same result", got.canonical() == ref.canonical(), "provisioners", [(q.name, q.weight, q.limits) for q in p.provisioners])
for k in sorted(set(ref.reasons) | set(got.reasons)):
    if ref.reasons.get(k) != got.reasons.get(k):
        print("pod", k, "oracle", hex(ref.reasons.get(k, -1)), "gpu", hex(got.reasons.get(k, -1)), "stage", ref.final_stage[k])
