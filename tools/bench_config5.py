#!/usr/bin/env python3
"""BASELINE config #5 shape (full constraint set: taints, Gt/Lt selectors, zonal/hostname/capacity-type spread, pod (anti-)affinity,
host ports, two weighted provisioners with limits) at a size that generates in seconds.  Timing only -- parity for this shape
is covered at sizes the CPU oracle finishes (tests/test_parity.py)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sizes = int(sys.argv[2]) if len(sys.argv) > 2 else 56
t0 = time.time(); pr = W.config5(pods=pods, sizes=sizes); t1 = time.time()
fp = S.FlatProblem(pr); fp.upload(0); fp.grid(want_bits=False); t2 = time.time()
fp.solve(decode=False)
ms = []
for _ in range(3):
    r = fp.solve()
    ms.append(fp.kernel_ms)
ms.sort()
print(json.dumps({"workload": f"config #5 shape: {pods} pods, {fp.dims['T']} instance types, {fp.dims['G']} topology groups, {fp.dims['C']} pod classes",
                  "kernel_ms": ms[1], "decisions_per_s": pods / (ms[1] / 1e3), "new_nodes": len(r.new_nodes), "unschedulable": len(r.unscheduled),
                  "generate_s": t1 - t0, "flatten_upload_s": t2 - t1}))
