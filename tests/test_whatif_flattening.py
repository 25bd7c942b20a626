"""`scheduler.open_whatifs` flattens the cluster snapshot ONCE (catalogue, label universes, templates, every state node's row) and derives
each what-if from it natively -- candidates leave the state nodes, their pods become the pending batch (deprovisioning/helpers.go:42-115
simulateScheduling).  The one-Problem-per-what-if route flattens every what-if from scratch.  The two encodings may number label values and
instance-type states differently (the shared one sees the whole snapshot), so they are compared by what they MEAN: same shape on the CPU,
bit-identical Solve results on the GPU (and, at BASELINE's size, tests/test_parity.py::test_full_size_config4_512_whatifs...)."""
import numpy as np
import pytest

from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import DO_NOT_SCHEDULE, LABEL_ZONE, LabelSelector, TopologySpreadConstraint, Volume

SHAPE = ("P", "C", "T", "M", "E", "R", "G", "GH")


def _cases():
    its, prov, nodes, bound = W.cluster_snapshot(existing=64, sizes=6, seed=5)
    sets = [list(range(0, i + 1)) for i in range(6)] + [[9], [33, 12], [63, 0, 31]]
    yield its, prov, nodes, bound, sets, False
    its, prov, nodes, bound = W.cluster_snapshot(existing=40, sizes=5, seed=8)
    rs = np.random.RandomState(1)
    for i, pods in enumerate(bound):                      # some bound pods carry a zonal spread: countDomains reads the cluster pods
        for p in pods:
            if rs.rand() < 0.3:
                p.spread = [TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, LabelSelector({"my-label": p.labels["my-label"]}))]
    yield its, prov, nodes, bound, [[0, 1, 2], [5], [7, 3], list(range(10))], True
    its, prov, nodes, bound = W.cluster_snapshot(existing=24, sizes=5, seed=11)
    rs = np.random.RandomState(2)
    for i, pods in enumerate(bound):                      # CSI volume limits: the candidates' pods bring their claims along, the other nodes keep theirs
        for p in pods:
            if rs.rand() < 0.5:
                p.volumes = [Volume("ebs.csi", f"default/shared-{rs.randint(8)}")] + ([Volume("ebs.csi", f"default/{p.uid}-data")] if rs.rand() < 0.5 else [])
        nodes[i].volumes = [v for j, v in enumerate(x for p in pods for x in p.volumes) if v not in [y for q in pods for y in q.volumes][:j]]
        nodes[i].volume_limits = {"ebs.csi": int(rs.randint(1, 6))} if rs.rand() < 0.8 else {}
    yield its, prov, nodes, bound, [[0], [1, 2], [5, 9, 13], list(range(8))], False


def _both(its, prov, nodes, bound, sets, with_cluster_pods):
    snap, pod_node = W.snapshot_problem(its, prov, nodes, bound, with_cluster_pods)
    native = S.open_whatifs(snap, pod_node, sets, threads=3)
    ref = [S.FlatProblem(W.whatif(its, prov, nodes, bound, cs, with_cluster_pods)) for cs in sets]
    return native, ref


def test_native_whatifs_have_the_same_shape():
    for case in _cases():
        native, ref = _both(*case)
        try:
            for a, b in zip(native, ref):
                assert {k: a.dims[k] for k in SHAPE} == {k: b.dims[k] for k in SHAPE}
            assert len({f.fingerprint() for f in native}) == len(native)      # different what-ifs, different problems
        finally:
            for f in native + ref:
                f.close()


@pytest.mark.gpu
def test_native_whatifs_solve_identically():
    for case in _cases():
        native, ref = _both(*case)
        try:
            got, _, _ = S.solve_batch(native)
            want, _, _ = S.solve_batch(ref)
            for a, b in zip(got, want):
                assert a.canonical() == b.canonical()
        finally:
            for f in native + ref:
                f.close()
