"""Model check of the pack kernel's round speculation (CPU only).

The multi-wave kernel evaluates several queued pods against one snapshot of the first 64 candidates and lets a
resolver pick each pod's node from the fit bitmaps alone (karpenter_core_amd/csrc/ksolve.hip, "Speculation round").
`oracle.solve_spec` restates the resolver's rules on top of the sequential reference algorithm: every prediction
is committed through the real `add()` and compared.  A rule that could pick a different node than the reference
shows up here as a violation, without a GPU."""
import pytest

from karpenter_core_amd import workloads as W
from oracle import oracle_py as O


def _check(problem, width=7):
    want = O.solve(problem).canonical()
    got, ctr = O.solve_spec(problem, width)
    assert ctr["violations"] == 0, ctr
    assert got.canonical() == want
    return ctr


@pytest.mark.parametrize("maker", [
    lambda: W.config1(pods=300, types=20, seed=3),
    lambda: W.config2(pods=400, sizes=5, seed=4),
    lambda: W.config3(pods=700, sizes=10, seed=7),
    lambda: W.config5(pods=500, sizes=8, seed=9),
    lambda: W.reference_benchmark(350, 40, seed=2),
])
def test_predictions_match_the_sequential_algorithm(maker):
    ctr = _check(maker())
    assert ctr["predicted"] > 0


def test_existing_nodes_keep_their_place():
    its, prov, nodes, bound = W.cluster_snapshot(existing=48, sizes=6, seed=5)
    ctr = _check(W.whatif(its, prov, nodes, bound, list(range(0, 9))))
    assert ctr["predicted"] > 0


@pytest.mark.parametrize("width", [2, 4, 16])
def test_round_width_does_not_matter(width):
    _check(W.config3(pods=350, sizes=6, seed=11), width)
