#!/usr/bin/env python3
"""How far can Go's unstable sort.Slice (scheduler.go:183) move a result away from the canonical stable order -- and what of that can a caller see?
Runs the CPU oracle twice per workload (stable / restated pdqsort_func) and compares
  * pod placement: pods on a node with another index, nodes that keep their exact pod set;
  * the MACHINE multiset -- what `north_star` calls the Machine set and what the controller launches (machinetemplate.go:77-100 ToMachine):
    (provisioner, InstanceTypeOptions in order, requests, requirements) per new node, compared as a multiset, and the same without the requests;
  * the summed launch price under the in-memory provider's pick (fake/cloudprovider.go:79-84: cheapest available offering the requirements admit).
Writes profiles/r03_gosort_machines.json.   usage: measure_gosort.py [config2|config3|config5 ...]   (config3 at 100k pods: ~4 minutes)"""
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from karpenter_core_amd import workloads as W  # noqa: E402
from oracle import consolidation_ref as R, oracle_py as O  # noqa: E402

WORKLOADS = {"config2": lambda: W.config2(), "config3": lambda: W.config3(), "config5": lambda: W.config5(pods=30_000, sizes=50, seed=46)}


def machine(n, with_requests=True):
    reqs = tuple(sorted((k, r.complement, tuple(sorted(r.values)), r.greater_than, r.less_than) for k, r in n.requirements.items()))
    return (n.provisioner, tuple(n.instance_types), tuple(sorted(n.requests.items())) if with_requests else (), reqs)


def common(a, b):
    return sum((collections.Counter(a) & collections.Counter(b)).values())


def compare(name, p):
    t = time.time(); a = O.solve(p); ta = time.time() - t
    t = time.time(); b = O.solve(p, gosort=True); tb = time.time() - t
    na = {pod: i for i, n in enumerate(a.new_nodes) for pod in n.pods}
    nb = {pod: i for i, n in enumerate(b.new_nodes) for pod in n.pods}
    sa = {frozenset(n.pods) for n in a.new_nodes}; sb = {frozenset(n.pods) for n in b.new_nodes}
    price = lambda res: sum((R.launch_pick(p.instance_types, n) or (None, 0.0))[1] for n in res.new_nodes)
    pa, pb = price(a), price(b)
    return {"workload": name, "pods": len(p.pods), "instance_types": len(p.instance_types),
            "stable": {"new_nodes": len(a.new_nodes), "unscheduled": len(a.unscheduled), "summed_launch_price": pa, "oracle_seconds": round(ta, 1)},
            "gosort_restated": {"new_nodes": len(b.new_nodes), "unscheduled": len(b.unscheduled), "summed_launch_price": pb, "oracle_seconds": round(tb, 1)},
            "pods_on_a_different_node_index": sum(1 for k in na if nb.get(k) != na[k]),
            "nodes_with_identical_pod_sets": len(sa & sb),
            "machines_in_common_as_multiset": common([machine(n) for n in a.new_nodes], [machine(n) for n in b.new_nodes]),
            "machines_in_common_ignoring_requests": common([machine(n, False) for n in a.new_nodes], [machine(n, False) for n in b.new_nodes]),
            "launch_price_relative_difference": abs(pa - pb) / pa if pa else 0.0,
            "same_unscheduled_set": sorted(a.unscheduled) == sorted(b.unscheduled)}


if __name__ == "__main__":
    names = sys.argv[1:] or list(WORKLOADS)
    path = os.path.join(ROOT, "profiles", "r03_gosort_machines.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["note"] = ("gosort = SURVEY App. C.1 pdqsort_func restated from memory; offline-checkable properties of Go 1.19 sort/zsortfunc.go are pinned in tests/test_gosort.py "
                   "(insertion sort up to 12 elements, sorted-plus-one-increment inputs, always a sorting permutation); it was never run against a Go toolchain: the numbers "
                   "bound the effect of an unstable sort, they do not predict a Go binary")
    for nm in names:
        out[nm] = compare(nm, WORKLOADS[nm]())
        print(nm, out[nm], flush=True)
        json.dump(out, open(path, "w"), indent=1)
