#!/usr/bin/env python3
"""What a decline of ks_pack_rr costs: the Solve's GPU time when the register-resident kernel starts and gives the Solve back mid-run (its own time until the decline +
ks_pack's from scratch) against ks_pack alone (KS_FLAG_NO_RR) on the same resident problem.   tools/decline_penalty.py  (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from karpenter_core_amd import scheduler as S  # noqa: E402
import test_fuzz_mid as M  # noqa: E402
import test_rr_gpu as R  # noqa: E402

cases = [("mid seed 10 (code 3: > 8 exact-filter exclusions)", lambda: M.mid_problem(10)), ("mid seed 50 (code 3)", lambda: M.mid_problem(50)),
         ("mid seed 12 (code 7: a class outside the feature set)", lambda: M.mid_problem(12)), ("mid seed 57 (code 7)", lambda: M.mid_problem(57)),
         ("3 700 one-pod nodes (code 4: more nodes than the registers hold)", lambda: R._anti_affinity_herd(3700)), ("1 100 pods on one node (code 2: count field)", lambda: R._crowded_node(1100)),
         ("80 existing nodes (code 1: static, before anything runs)", lambda: R.W.whatif(*R.W.cluster_snapshot(existing=80, sizes=10, seed=5), candidates=[0, 1, 2], with_cluster_pods=False))]
print(f"{'problem':70s} {'pods':>6s} {'code':>4s} {'rr started: wall ms':>20s} {'ks_pack alone: wall ms':>23s} {'penalty':>8s}")
for name, mk in cases:
    p = mk()
    row = []
    for flags in (0, S.KS_FLAG_NO_RR):
        fp = S.FlatProblem(p, flags=flags); fp.upload(0); fp.grid(want_bits=False); fp.solve(decode=False)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); fp.solve(decode=False); best = min(best, (time.perf_counter() - t) * 1e3)
        row.append((best, fp.rr_status()))
        fp.close()
    (a, (st, code)), (b, _) = row
    print(f"{name:70s} {len(p.pods):6d} {code:4d} {a:20.2f} {b:23.2f} {a - b:8.2f}")
