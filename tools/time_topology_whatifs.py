"""Timing of consolidation what-ifs over a snapshot whose bound pods carry topology terms (spread / affinity / anti-affinity / preferred): the route
that derives them on the device (ks_whatifs_open with per-node tables) against the route that flattens them one by one on the host.
    python tools/time_topology_whatifs.py [NODES] [WHATIFS]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from karpenter_core_amd import scheduler as S, workloads as W      # noqa: E402
from test_whatif_derived import _topology_snapshot                  # noqa: E402  (the generator of the differential tests)

nodes_n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_whatifs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
its, prov, nodes, bound, snap, pod_node = _topology_snapshot(nodes_n, 50, 45, spare=-1, extras=True, anti=True)
sets = W.config4_sets(n_whatifs, nodes_n, 45)
parsed = S.ParsedProblem(snap)
out = {"nodes": len(nodes), "bound_pods": len(snap.pods), "cluster_pod_records": len(snap.cluster_pods), "whatifs": len(sets), "groups": S.FlatProblem(snap).dims["G"]}


def run(derive):
    t0 = time.perf_counter()
    flats = S.open_whatifs(parsed, pod_node, sets, derive=derive)
    t1 = time.perf_counter()
    res, kms, _ = S.solve_batch(flats, decode=False)
    t2 = time.perf_counter()
    words = (len(its) + 63) // 64
    rec = S.result_records(flats, list(range(len(sets))), words)
    t3 = time.perf_counter()
    for f in flats:
        f.close()
    return {"open_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3, "kernel_ms": kms, "records_ms": (t3 - t2) * 1e3, "total_ms": (t3 - t0) * 1e3}, rec


cold, rd = run(True)
warm = [run(True)[0] for _ in range(3)]
out["derived_first_batch"] = cold
out["derived_warm"] = sorted(warm, key=lambda x: x["total_ms"])[1]
fl, rf = run(False)
out["flattened_one_by_one_first"] = fl
out["flattened_one_by_one_warm"] = run(False)[0]
out["same_records"] = bool((rd == rf).all())
out["decisions"] = int(sum(len(b) for s in sets for b in [[p for n in s for p in bound[n]]]))
print(json.dumps(out))
