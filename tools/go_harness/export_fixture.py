#!/usr/bin/env python3
"""Writes a Solve() problem as the JSON fixture tools/go_harness/solve_dump_test.go reads (the day a Go toolchain exists: BASELINE.md section 2), and -- with --want -- the
canonical result the CPU oracle gives for it, in the line format the Go test prints, so that the two can be diffed.
  python tools/go_harness/export_fixture.py config3 --pods 20000 > /tmp/config3.json
  python tools/go_harness/export_fixture.py config3 --pods 20000 --want > /tmp/config3.want
Covers what BASELINE configs[0..2] use (new-node problems: pods, instance types, provisioners); state nodes / cluster pods are not exported."""
import argparse
import dataclasses
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from karpenter_core_amd import workloads as W      # noqa: E402


def lines(res, problem):
    """One line per new node in creation order: provisioner | pod uids in Node.Pods order | instance type names (sorted) | requests -- what solve_dump_test.go prints."""
    out = []
    for n in res.new_nodes:
        out.append("NODE %s | %s | %s | %s" % (n.provisioner, " ".join(problem.pods[i].uid for i in n.pods), " ".join(sorted(n.instance_types)),
                                                " ".join(f"{k}={v}" for k, v in sorted(n.requests.items()))))
    out.append("UNSCHEDULED " + " ".join(problem.pods[i].uid for i in res.unscheduled))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["config1", "config2", "config3", "reference_benchmark"])
    ap.add_argument("--pods", type=int, default=1000)
    ap.add_argument("--want", action="store_true")
    a = ap.parse_args()
    p = getattr(W, a.config)(a.pods) if a.config != "reference_benchmark" else W.reference_benchmark(a.pods)
    if a.want:
        from oracle import oracle_py
        print("\n".join(lines(oracle_py.solve(p), p)))
    else:
        json.dump({"instance_types": [dataclasses.asdict(t) for t in p.instance_types], "provisioners": [dataclasses.asdict(v) for v in p.provisioners],
                   "pods": [dataclasses.asdict(q) for q in p.pods]}, sys.stdout)
