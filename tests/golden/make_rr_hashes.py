#!/usr/bin/env python3
"""Golden fingerprints of the rr-covered fuzz family (tests/test_fuzz_rr.py), produced by the CPU oracle (a few minutes in all; some
seeds take the oracle a minute or two, which is why the GPU test compares fingerprints instead of re-running it on the GPU box).
    python tests/golden/make_rr_hashes.py        # rewrites tests/golden/rr_hashes.json"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as O  # noqa: E402
import test_fuzz_rr as T  # noqa: E402

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rr_hashes.json")
out = json.load(open(path)) if os.path.exists(path) and "--all" not in sys.argv else {}      # (seeds already there are kept: pass --all to redo them)
for seed in T.RR_SEEDS:
    if str(seed) in out:
        continue
    p = T.rr_problem(seed)
    t = time.time()
    r = O.solve(p)
    out[str(seed)] = dict(T.fingerprints(r), pods=len(p.pods), instance_types=len(p.instance_types), new_nodes=len(r.new_nodes),
                          unscheduled=len(r.unscheduled), relaxed_pods=sum(1 for s in r.final_stage if s > 0) if isinstance(r.final_stage, list) else None,
                          oracle_seconds=round(time.time() - t, 1))
    print(seed, out[str(seed)], flush=True)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
