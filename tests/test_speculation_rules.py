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


# ---- the resolver as the kernel runs it since round 2 (oracle.solve_spec_v2): moved candidates stay in play, runs (SWEEP / CLIMB), hostname
# ---- slack / exact reject, zonal spread followed exactly on pinned nodes (dd / rdyn / unknown candidates), closed candidates ----
V2 = 2          # flags bit 1: the v2 model; bit 2: leave the rdyn rule out (mutation)


def _check_v2(problem, width=64, max_classes=7, flags=V2):
    want = O.solve(problem).canonical()
    got, ctr = O.solve_spec(problem, width, flags, max_classes)
    assert got.canonical() == want
    return ctr


def _mid(seed):
    import test_fuzz_mid as T
    return T.mid_problem(seed)


@pytest.mark.parametrize("maker", [
    lambda: W.config2(pods=600, sizes=5, seed=4),
    lambda: W.config3(pods=3000, sizes=10, seed=3),
    lambda: W.config5(pods=800, sizes=8, seed=9),
    lambda: _mid(0),            # zonal spreads at maxSkew 1-3 next to classes that read the same groups with a second topology item
    lambda: _mid(3),
    lambda: _mid(14),           # in-flight nodes, two provisioners, a spread next to a zone requirement of the pod's own
])
def test_v2_resolver_rules_match_the_sequential_algorithm(maker):
    ctr = _check_v2(maker())
    assert ctr["violations"] == 0, ctr
    assert ctr["predicted"] > 0 and ctr["rounds"] > 0


def test_v2_rules_on_existing_nodes():
    its, prov, nodes, bound = W.cluster_snapshot(existing=48, sizes=6, seed=5)
    ctr = _check_v2(W.whatif(its, prov, nodes, bound, list(range(0, 9))))
    assert ctr["violations"] == 0 and ctr["predicted"] > 0


def test_the_model_catches_a_missing_rdyn_rule():
    """Round 2's kernel let a class the resolver does not follow (a second topology item, or a requirement of its own on the spread key) go on
    reading a zonal spread group after the round had recorded into it exactly -- found on the GPU by the mid-scale fuzz.  With that rule left
    out the model reports it on the CPU."""
    assert _check_v2(_mid(0), flags=V2 | 4)["violations"] > 0
    assert _check_v2(_mid(0))["violations"] == 0


# ---------------- the watermark over the existing nodes (ksolve.hip ClsPlan::mono) ----------------
def _topology_whatif(seed, cand, spare=-1):
    import sys
    import os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_whatif_derived import _topology_snapshot, _whatif_problem
    its, prov, nodes, bound, snap, pod_node = _topology_snapshot(40, 6, seed, spare=spare, anti=True)
    return _whatif_problem(snap, pod_node, cand)


@pytest.mark.parametrize("maker", [lambda s=s: _topology_whatif(s, list(range(0, 14))) for s in range(8)] +          # roomy clusters: anti-affinity per hostname and per zone,
                                  [lambda s=s: _topology_whatif(s, [1, 4, 7, 9, 20, 21, 22, 30], spare=2) for s in (3, 5)] +      # spreads, affinities; clusters full by pod count
                                  [lambda: W.whatif(*W.cluster_snapshot(existing=48, sizes=6, seed=5), list(range(0, 9)))])
def test_existing_nodes_that_refused_a_watermark_class_keep_refusing(maker):
    """The kernel starts a pod's scan of the existing nodes where the last pod of its class stopped.  Since round 3 that includes
    classes whose only topology items are anti-affinity ones: a count of 0 is needed and counts only grow.  The CPU model dry-runs every such pod
    against EVERY existing node just before it is placed: a node on record as having refused the class must refuse again."""
    pr = maker()
    res, ctr = O.watermark_check(pr)
    assert res.canonical() == O.solve(pr).canonical()
    assert ctr["violations"] == 0, ctr
    assert ctr["watermark_pods"] > 0 and ctr["dry_runs"] > 0


def test_the_watermark_model_covers_anti_affinity_and_rejects_spread():
    """... the anti-affinity classes are really among the checked ones, and the check can tell: with spread / affinity classes ALSO treated as
    watermark classes (a mutation) it reports nodes that refused a class and accept it later (the minimum moved, a domain began to count)."""
    checked = [O.watermark_check(_topology_whatif(s, list(range(0, 14))))[1] for s in (0, 3)]
    assert all(c["violations"] == 0 for c in checked) and all(c["with_anti_affinity"] > 0 for c in checked)
    assert O.watermark_check(_topology_whatif(0, list(range(0, 14))), mutate=True)[1]["violations"] > 0


@pytest.mark.parametrize("seed", range(6))
def test_refusals_before_the_topology_steps_are_monotone_for_every_class(seed):
    """A lead for the next round, checked here so that the kernel can adopt it safely: whatever the class -- spread and affinity too -- an
    existing node that refused it BEFORE the topology steps (taints, ports, volume limits, resources, the pod's own requirements) refuses it for the rest
    of the Solve.  Classes with requirements on custom keys stay out (a node's requirement on a custom key can appear with another pod's NotIn)."""
    pr = _topology_whatif(seed, list(range(0, 14)), spare=[-1, 2][seed % 2])
    res, ctr = O.watermark_check(pr, pre_topology_only=True)
    assert res.canonical() == O.solve(pr).canonical()
    assert ctr["violations"] == 0, ctr
    assert ctr["refusals_recorded"] > 0 and ctr["recorded_pairs_rechecked"] > 0
