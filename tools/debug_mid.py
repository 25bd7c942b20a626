#!/usr/bin/env python3
"""Where does a kernel variant part from the oracle on a mid-scale fuzz problem?  (run on the GPU box)
usage: tools/debug_mid.py SEED [SEED ...]      env KS_ONE_WAVE / KS_NO_LEAN / KS_NO_DYN pick the variant
Prints, per seed, node / unscheduled counts of both sides and the first pod in COMMIT order whose node differs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from karpenter_core_amd import scheduler as S  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
import test_fuzz_mid as T  # noqa: E402


def placement(res):
    """pod -> (kind, node index) and the commit sequence of the canonical result."""
    where = {}
    for j, n in enumerate(res.new_nodes):
        for p in n.pods:
            where[p] = ("new", j)
    for e, pods in res.existing.items():
        for p in pods:
            where[p] = ("existing", e)
    return where


for seed in [int(a) for a in sys.argv[1:]]:
    p = T.mid_problem(seed)
    fp = S.FlatProblem(p)
    got = fp.solve()
    ref = O.solve(p)
    same = got.canonical() == ref.canonical()
    print(f"seed {seed}: pods {len(p.pods)}  gpu nodes {len(got.new_nodes)} unsched {len(got.unscheduled)}  oracle nodes {len(ref.new_nodes)} unsched {len(ref.unscheduled)}  "
          f"identical {same}  reasons equal {got.reasons == ref.reasons}", flush=True)
    if not same:
        wg, wr = placement(got), placement(ref)
        # walk the oracle's nodes in creation order, pods in commit order: first pod placed elsewhere
        shown = 0
        for j, n in enumerate(ref.new_nodes):
            for k, pod in enumerate(n.pods):
                if wg.get(pod) != ("new", j):
                    q = p.pods[pod]
                    kind = "spread" if q.spread else ("anti" if q.anti_required else ("aff" if q.affinity_required else ("pref" if q.preferred_affinity else ("sel" if q.node_selector else "generic"))))
                    print(f"   oracle node {j} pod #{k} = pod {pod} ({kind}, {q.containers[0].requests}) -> gpu {wg.get(pod)}; stage gpu {got.final_stage[pod] if got.final_stage else None} oracle {ref.final_stage[pod] if ref.final_stage else None}")
                    shown += 1
                    break
            if shown >= 4:
                break
        st = got.stats
        print("   stats", {k: st[k] for k in ("queue_pops", "relaxations", "full_checks", "full_fails") if k in st})
    fp.close()
