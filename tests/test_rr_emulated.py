"""(CPU) The register-resident pack kernel's SOURCE (karpenter_core_amd/csrc/ks_pack_rr.inc), compiled by g++ against the lane-fibre emulator of tests/sim and
driven through the same C ABI, against the oracle: rounds, RUN steps, the leader's batch formation, commits, dyn1 answers -- everything the kernel does on the
GPU but the timing.  Runs in a subprocess: the emulator build replaces the two libraries for the whole process (tests/simlib.py)."""
import json
import os
import subprocess
import sys

import pytest

from oracle import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import simlib
S = simlib.use_sim()
from karpenter_core_amd import workloads as W
import test_fuzz_mid as T
out = {}
def fp(res):
    return hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest()
cases = json.loads(sys.argv[1])
for name, kind, args in cases:
    p = getattr(W, kind)(**args) if kind != "mid" else T.mid_problem(args["seed"])
    try:
        r = S.solve_problem(p)
        out[name] = {"fp": fp(r), "rounds": r.stats.get("eq_pods", 0), "run_pods": r.stats.get("p22", 0)}
    except Exception as e:
        out[name] = {"error": str(e)[:200]}
print("RESULT " + json.dumps(out))
"""

CASES = [
    ("config1_1000", "config1", {"pods": 1000, "types": 50, "seed": 42}),
    ("config3_140", "config3", {"pods": 140, "sizes": 3, "seed": 1}),
    ("config3_700", "config3", {"pods": 700, "sizes": 10, "seed": 7}),
    ("config3_3500", "config3", {"pods": 3500, "sizes": 20, "seed": 44}),
    ("mid_0", "mid", {"seed": 0}),
    ("mid_3", "mid", {"seed": 3}),
]


@pytest.fixture(scope="module")
def emulated():
    env = dict(os.environ)
    env.pop("KS_TEST_SIM", None)
    code = CHILD % {"root": ROOT, "tests": HERE}
    pr = subprocess.run([sys.executable, "-c", code, json.dumps(CASES)], capture_output=True, text=True, env=env, timeout=1500)
    line = [l for l in pr.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, pr.stdout[-2000:] + pr.stderr[-2000:]
    return json.loads(line[-1][7:])


def _oracle_fp(name, kind, args):
    import hashlib
    from karpenter_core_amd import workloads as W
    import test_fuzz_mid as T
    p = getattr(W, kind)(**args) if kind != "mid" else T.mid_problem(args["seed"])
    return hashlib.sha256(json.dumps(O.solve(p).canonical(), sort_keys=True).encode()).hexdigest()


@pytest.mark.parametrize("name,kind,args", CASES, ids=[c[0] for c in CASES])
def test_rr_kernel_source_matches_oracle_on_the_emulator(emulated, name, kind, args):
    got = emulated[name]
    assert "error" not in got, got
    assert got["fp"] == _oracle_fp(name, kind, args)


def test_rr_runs_are_exercised(emulated):
    """The generic replicas of the config #3 shape go through RUN rounds (several pods per barrier): the test above covers that path, not only the one-pod picks."""
    assert emulated["config3_3500"]["run_pods"] > 1000
