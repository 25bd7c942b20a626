"""`scheduler.open_whatifs` (one snapshot parsed once, what-ifs derived and flattened natively on host threads) must produce
exactly the flat problems the one-Problem-per-what-if route produces (deprovisioning/helpers.go:42-115 simulateScheduling).
CPU only: compares a hash over every array behind `ks_problem`."""
import numpy as np

from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import DO_NOT_SCHEDULE, LABEL_ZONE, LabelSelector, TopologySpreadConstraint


def _both(its, prov, nodes, bound, sets, with_cluster_pods):
    snap, pod_node = W.snapshot_problem(its, prov, nodes, bound, with_cluster_pods)
    native = S.open_whatifs(snap, pod_node, sets, threads=3)
    ref = [S.FlatProblem(W.whatif(its, prov, nodes, bound, cs, with_cluster_pods)) for cs in sets]
    try:
        return [f.fingerprint() for f in native], [f.fingerprint() for f in ref], [f.dims for f in native], [f.dims for f in ref]
    finally:
        for f in native + ref:
            f.close()


def test_native_whatifs_flatten_identically():
    its, prov, nodes, bound = W.cluster_snapshot(existing=64, sizes=6, seed=5)
    sets = [list(range(0, i + 1)) for i in range(6)] + [[9], [33, 12], [63, 0, 31]]
    a, b, da, db = _both(its, prov, nodes, bound, sets, False)
    assert da == db and a == b
    assert len(set(a)) == len(sets)                       # different what-ifs, different problems


def test_native_whatifs_with_topology_and_cluster_pods():
    its, prov, nodes, bound = W.cluster_snapshot(existing=40, sizes=5, seed=8)
    rs = np.random.RandomState(1)
    for i, pods in enumerate(bound):                      # some bound pods carry a zonal spread: countDomains reads the cluster pods
        for p in pods:
            if rs.rand() < 0.3:
                p.spread = [TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, LabelSelector({"my-label": p.labels["my-label"]}))]
    sets = [[0, 1, 2], [5], [7, 3], list(range(10))]
    a, b, da, db = _both(its, prov, nodes, bound, sets, True)
    assert da == db and a == b
