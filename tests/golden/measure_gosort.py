#!/usr/bin/env python3
"""How far can Go's unstable sort.Slice (scheduler.go:183) move BASELINE configs[2]'s result away from the canonical stable order?
Runs the CPU oracle twice on the 100k-pod problem (stable / restated pdqsort) and writes profiles/r02_gosort_config3.json.  ~5 minutes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from karpenter_core_amd import workloads as W
from oracle import oracle_py as O
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = W.config3(pods=pods)
t = time.time(); a = O.solve(p); ta = time.time() - t
t = time.time(); b = O.solve(p, gosort=True); tb = time.time() - t
na = {pod: i for i, n in enumerate(a.new_nodes) for pod in n.pods}
nb = {pod: i for i, n in enumerate(b.new_nodes) for pod in n.pods}
# nodes are compared as SETS of pods (creation indices differ once the orders diverge)
sa = {frozenset(n.pods) for n in a.new_nodes}; sb = {frozenset(n.pods) for n in b.new_nodes}
out = {"workload": f"workloads.config3(pods={pods}) -- BASELINE configs[2]", "stable": {"new_nodes": len(a.new_nodes), "unscheduled": len(a.unscheduled), "oracle_seconds": round(ta, 1)},
       "gosort_restated": {"new_nodes": len(b.new_nodes), "unscheduled": len(b.unscheduled), "oracle_seconds": round(tb, 1)},
       "pods_on_a_different_node_index": sum(1 for k in na if nb.get(k) != na[k]), "nodes_with_identical_pod_sets": len(sa & sb),
       "note": "gosort = SURVEY App. C.1 pdqsort_func restated from memory, not validated against a Go toolchain: the numbers bound the effect of an unstable sort, they do not predict a Go binary"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_gosort_config3.json"), "w"), indent=1)
print(out)
