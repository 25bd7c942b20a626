#!/usr/bin/env python3
"""The reference's own benchmark matrix (scheduling_benchmark_test.go:54-74,113-171): fake.InstanceTypes(400), one provisioner,
makeDiversePods for {1, 50, 100, 500, 1000, 2000, 5000} pods; pods/sec = len(pods) / Solve time, scheduler construction
excluded.  Two differences, both on the strict side: topology is LIVE here (the reference benchmark passes an inert
`&scheduling.Topology{}`, :123) and pod UIDs are unique.  The reference asserts a floor of 100 pods/s for
batches over 100 pods (:48,178-182); no measured numbers are published.  Prints one JSON object.
Lives under tests/ (not collected by pytest) because it runs the CPU oracle next to the GPU path, which only test
infrastructure may do.  usage: python tests/perf_reference_matrix.py"""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import parse_result
from oracle import oracle_py

rows = []
for pods in (1, 50, 100, 500, 1000, 2000, 5000):
    pr = W.reference_benchmark(pods, instance_count=400, seed=42)
    fp = S.FlatProblem(pr); fp.upload(0); fp.grid(want_bits=False)
    fp.solve(decode=False)
    ks, ws = [], []
    for _ in range(5):
        r = fp.solve()
        ks.append(fp.kernel_ms); ws.append(fp.wall_ms)
    o = parse_result(oracle_py.solve_text(pr.to_ksp()))
    same = r.canonical() == o.canonical()
    osec = o.stats["solve_ns"] / 1e9
    rows.append({"pods": pods, "nodes": len(r.new_nodes), "gpu_kernel_ms": statistics.median(ks), "gpu_solve_wall_ms": statistics.median(ws),
                 "gpu_pods_per_s": pods / (statistics.median(ws) / 1e3), "cpu_oracle_ms": osec * 1e3, "cpu_oracle_pods_per_s": pods / osec,
                 "bit_identical": same})
    fp.close()
print(json.dumps({"benchmark": "scheduling_benchmark_test.go matrix, 400 instance types, live topology", "reference_floor_pods_per_s": 100.0, "rows": rows}))
