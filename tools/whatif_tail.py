#!/usr/bin/env python3
"""How long the LARGEST what-ifs of BASELINE configs[3] take alone: single-wave kernel (what a batch runs today) vs the multi-wave one.
The batch's kernel time is the longest what-if's time once the grid has fewer workgroups than the chip has slots."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from karpenter_core_amd import scheduler as S, workloads as W  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    its, prov, nodes, bound = W.cluster_snapshot(2048, 50, 45)
    snap, pn = W.snapshot_problem(its, prov, nodes, bound, False)
    parsed = S.ParsedProblem(snap)
    sets = W.config4_sets(n, 2048, 45)
    flats = S.open_whatifs(parsed, pn, sets)
    for f in flats:
        f.upload(0)
    _, kms, wms = S.solve_batch(flats, decode=False)
    _, kms, wms = S.solve_batch(flats, decode=False)
    order = sorted(range(len(flats)), key=lambda i: -flats[i].dims["P"])
    out = {"batch_kernel_ms": kms, "batch_wall_ms": wms, "whatifs": len(flats), "largest": []}
    for i in order[:6] + order[len(order) // 2:len(order) // 2 + 2]:
        f = flats[i]
        row = {"whatif": i, "pods": f.dims["P"], "classes": f.dims["C"], "groups": f.dims["G"], "E": f.dims["E"]}
        for name, env in (("multi_wave_ms", None), ("one_wave_ms", "1")):
            if env:
                os.environ["KS_ONE_WAVE"] = env
            else:
                os.environ.pop("KS_ONE_WAVE", None)
            f.solve(decode=False)
            f.solve(decode=False)
            row[name] = f.kernel_ms
        os.environ.pop("KS_ONE_WAVE", None)
        out["largest"].append(row)
    print(json.dumps(out, indent=1))
    for f in flats:
        f.close()


if __name__ == "__main__":
    main()
