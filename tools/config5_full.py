#!/usr/bin/env python3
"""BASELINE config #5 at its stated size (1 M pods, 5 000 instance types, full constraint set) on the GPU: one Solve, kernel time, and the
size-independent properties of the result (the CPU oracle needs hours here; parity for this shape is pinned at 100 k pods by
tests/golden/config_hashes.json).  usage: tools/config5_full.py [pods] [sizes]   -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import pod_requests_milli, parse_quantity_milli
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sizes = int(sys.argv[2]) if len(sys.argv) > 2 else 50
t0 = time.time(); pr = W.config5(pods=pods, sizes=sizes); t1 = time.time()
fp = S.FlatProblem(pr); fp.upload(0); fp.grid(want_bits=False); t2 = time.time()
res = fp.solve(); t3 = time.time()
kernel_ms = fp.kernel_ms
# ---- properties ----
placed = [i for n in res.new_nodes for i in n.pods]
assert len(placed) == len(set(placed)), "a pod placed twice"
assert sorted(placed + list(res.unscheduled)) == list(range(len(pr.pods))), "a pod neither placed nor reported"
alloc = {it.name: {k: parse_quantity_milli(v) - parse_quantity_milli(it.overhead.get(k, "0")) for k, v in it.capacity.items()} for it in pr.instance_types}
reqs = [pod_requests_milli(p) for p in pr.pods]
for n in res.new_nodes:
    tot = {}
    for i in n.pods:
        for k, v in reqs[i].items():
            tot[k] = tot.get(k, 0) + v
    assert tot == n.requests, "node request totals"
    assert n.instance_types, "a node with no instance type option"
    for name in n.instance_types[:8] + n.instance_types[-8:]:
        assert all(v <= alloc[name].get(k, 0) for k, v in tot.items()), "an option that does not fit"
    vals = [pr.pods[i].labels.get("my-affininity") for i in n.pods if pr.pods[i].anti_required]
    assert len(vals) == len(set(vals)), "hostname anti-affinity violated"
    ports = [hp.port for i in n.pods for c in pr.pods[i].containers for hp in (c.ports or [])]
    assert len(ports) == len(set(ports)), "host port conflict on a node"
print(json.dumps({"workload": f"config #5: {pods} pods, {fp.dims['T']} instance types, {fp.dims['G']} topology groups, {fp.dims['C']} pod classes",
                  "kernel_ms": kernel_ms, "decisions_per_s": pods / (kernel_ms / 1e3), "new_nodes": len(res.new_nodes), "unschedulable": len(res.unscheduled),
                  "generate_s": round(t1 - t0, 1), "flatten_upload_tables_s": round(t2 - t1, 1), "solve_decode_s": round(t3 - t2, 1), "properties": "ok"}))
