#!/usr/bin/env python3
"""Fingerprints (ksh_fingerprint: a hash of every array of the flat problem) of the host flattening on a fixed set of problems:
the 64 fuzz problems (every feature incl. relaxation chains, volumes, existing nodes), the reference benchmark mix, the BASELINE
config shapes at small sizes, and four what-ifs over a snapshot.  tests/test_cabi.py::test_flattening_fingerprints_are_pinned
compares against them, so a change of the host code that is meant to be result-neutral (threading, caching, ordering of work)
is checked on the CPU.  Regenerate after a DELIBERATE change of the encoding:  python tests/golden/make_flat_fingerprints.py"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))


def compute():
    from karpenter_core_amd import scheduler as S, workloads as W
    from test_fuzz import fuzz_problem
    out = {}

    def one(name, problem):
        fp = S.FlatProblem(problem); out[name] = "%016x" % fp.fingerprint(); fp.close()
    for seed in range(64):
        one(f"fuzz-{seed}", fuzz_problem(seed))
    one("reference-benchmark-3000x60", W.reference_benchmark(3000, 60, seed=2))
    one("config1-1000x50", W.config1(pods=1000, types=50, seed=42))
    one("config2-4000x25", W.config2(pods=4000, sizes=25, seed=44))
    one("config3-20000x50", W.config3(pods=20000, sizes=50, seed=44))
    one("config5-20000x50", W.config5(pods=20000, sizes=50, seed=46))
    its, prov, nodes, bound = W.cluster_snapshot(existing=64, sizes=6, seed=5)
    snap, pod_node = W.snapshot_problem(its, prov, nodes, bound, True)
    for i, f in enumerate(S.open_whatifs(snap, pod_node, [[0, 1, 2], [5], [7, 3], list(range(10))], threads=3)):
        out[f"whatif-{i}"] = "%016x" % f.fingerprint()
    return out


if __name__ == "__main__":
    json.dump(compute(), open(os.path.join(HERE, "flat_fingerprints.json"), "w"), indent=0, sort_keys=True)
    print("written")
