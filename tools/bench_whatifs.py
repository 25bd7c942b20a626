#!/usr/bin/env python3
"""BASELINE config #4: N independent consolidation what-if Solve()s over one cluster snapshot, solved in ONE
launch (one single-wave workgroup per what-if).  Reports aggregate decisions/s.  (Parity of batched what-ifs against the CPU
oracle is a test: tests/test_parity.py::test_whatifs_single_and_batched, tests/test_consolidation.py.)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--whatifs", type=int, default=512)
ap.add_argument("--existing", type=int, default=2048)
ap.add_argument("--sizes", type=int, default=50)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--threads", type=int, default=0, help="host threads for the native what-if flattening (0 = min(32, cores))")
a = ap.parse_args()
t0 = time.time()
its, prov, nodes, bound = W.cluster_snapshot(a.existing, a.sizes, 45)
sets = W.config4_sets(a.whatifs, a.existing, 45)
snap, pod_node = W.snapshot_problem(its, prov, nodes, bound, False)
t1 = time.time()
flats = S.open_whatifs(snap, pod_node, sets, threads=a.threads)   # one snapshot, native per-what-if flattening on host threads
t15 = time.time()
for f in flats:
    f.upload(0)
t2 = time.time()
pods = sum(f.dims["P"] for f in flats)
S.solve_batch(flats, decode=False)                      # warm-up (also builds the static tables)
ms = []
for _ in range(a.steps):
    _, kms, wms = S.solve_batch(flats, decode=False)
    ms.append((kms, wms))
kms = sorted(m[0] for m in ms)[len(ms) // 2]; wms = sorted(m[1] for m in ms)[len(ms) // 2]
print(json.dumps({"workload": f"config #4: {a.whatifs} what-ifs over {a.existing} existing nodes, {len(its)} instance types",
                  "whatifs": a.whatifs, "pod_decisions": pods, "largest_whatif_pods": max(f.dims["P"] for f in flats),
                  "kernel_ms": kms, "wall_ms": wms, "decisions_per_s": pods / (wms / 1e3), "whatifs_per_s": a.whatifs / (wms / 1e3),
                  "generate_s": t1 - t0, "flatten_s": t15 - t1, "upload_s": t2 - t15, "host_cores": os.cpu_count()}))
