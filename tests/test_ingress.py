"""Binary pod ingress (include/kshost.h `ksh_pods_ingest` / `ksh_solve_from_batch`, grammar in host/kspb.hpp): the pending pods of
provisioner.go:301-307 handed over as flat arrays instead of KSP1 text.  The flat problem must be the one the text route produces -- every
array of it (`ksh_fingerprint`) -- whatever the number of blocks, and malformed blocks must be refused, not read out of bounds."""
import dataclasses

import numpy as np
import pytest

from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import (ClusterPod, LABEL_HOSTNAME, LabelSelector, PodAffinityTerm, PodBlockWriter, Volume, pods_to_blocks)
from test_fuzz import fuzz_problem
from test_fuzz_mid import mid_problem


def _routes(pr, n_blocks):
    env = S.ParsedProblem(dataclasses.replace(pr, pods=[]))
    batch = S.PodBatch(pods_to_blocks(pr.pods, n_blocks))
    return env, batch


def _problems():
    yield "config3", W.config3(pods=3000, sizes=10, seed=3)
    yield "config2", W.config2(pods=1500)
    for seed in (0, 3, 7, 11, 19, 42):
        yield f"fuzz{seed}", fuzz_problem(seed)          # host ports, volumes, existing nodes, limits, hostname selectors, relaxations
    yield "mid4", mid_problem(4)


@pytest.mark.parametrize("n_blocks", [1, 3])
def test_binary_route_flattens_to_the_same_problem(n_blocks):
    for name, pr in _problems():
        env, batch = _routes(pr, n_blocks)
        a, b = S.open_batch(env, batch), S.FlatProblem(pr)
        try:
            assert batch.n_pods == len(pr.pods) and 1 <= batch.n_specs <= len(pr.pods)
            assert a.dims == b.dims, name
            assert a.fingerprint() == b.fingerprint(), name
        finally:
            a.close(); b.close(); batch.close(); env.close()


def test_cluster_pods_and_volumes_through_the_binary_route():
    """countDomains / inverse anti-affinity ask which cluster pods are in the batch (by uid); volume claims ride in the spec record."""
    its, prov, nodes, bound = W.cluster_snapshot(existing=24, sizes=5, seed=11)
    rs = np.random.RandomState(2)
    for i, pods in enumerate(bound):
        for p in pods:
            if rs.rand() < 0.5:
                p.volumes = [Volume("ebs.csi", f"default/shared-{rs.randint(8)}")] + ([Volume("ebs.csi", f"default/{p.uid}-data")] if rs.rand() < 0.5 else [])
        nodes[i].volumes = [v for j, v in enumerate(x for p in pods for x in p.volumes) if v not in [y for q in pods for y in q.volumes][:j]]
        nodes[i].volume_limits = {"ebs.csi": int(rs.randint(1, 6))} if rs.rand() < 0.8 else {}
    pr = W.whatif(its, prov, nodes, bound, [0, 3, 5, 9], True)
    assert pr.cluster_pods and any(p.volumes for p in pr.pods)
    pr.cluster_pods.append(ClusterPod(uid="other", namespace="default", node_name=nodes[1].name, labels={"my-label": "a"},
                                      anti_required=[PodAffinityTerm(LABEL_HOSTNAME, LabelSelector({"my-label": "a"}))]))
    env, batch = _routes(pr, 2)
    a, b = S.open_batch(env, batch), S.FlatProblem(pr)
    assert a.fingerprint() == b.fingerprint()


def test_map_order_costs_a_spec_not_the_result():
    """A caller that lists a map in another order gets two specs where one would do -- and still the same flat problem."""
    pr = W.config3(pods=600, sizes=6, seed=9)
    w = PodBlockWriter()
    for p in pr.pods:
        w.add(p)
    arr = w.arrays()
    # rewrite pod 0's record with its label map reversed by hand: labels = N {k v}; config3 pods carry >= 1 label -- give pod 0 two, reversed
    p0 = dataclasses.replace(pr.pods[0], labels={"zz": "1", **pr.pods[0].labels})
    p1 = dataclasses.replace(p0, uid="pod-twin")
    pods = [p0, p1] + pr.pods[1:]
    pr2 = dataclasses.replace(pr, pods=pods)
    w2 = PodBlockWriter()
    for p in pods:
        w2.add(p)
    arr2 = w2.arrays()
    words = arr2["spec_words"].copy()
    o0 = int(arr2["spec_off"][0])
    # record 0: ns, then labels: count, (k,v)*: swap the two pairs
    assert words[o0 + 1] == 2
    words[o0 + 2], words[o0 + 3], words[o0 + 4], words[o0 + 5] = words[o0 + 4], words[o0 + 5], words[o0 + 2], words[o0 + 3]
    arr3 = dict(arr2, spec_words=words)
    env = S.ParsedProblem(dataclasses.replace(pr2, pods=[]))
    sorted_b, swapped_b = S.PodBatch([arr2]), S.PodBatch([arr3])
    assert swapped_b.n_specs == sorted_b.n_specs       # the content merge across records finds the twin again
    a, b, c = S.open_batch(env, sorted_b), S.open_batch(env, swapped_b), S.FlatProblem(pr2)
    assert a.fingerprint() == b.fingerprint() == c.fingerprint()
    assert arr["n_pods"] == 600


def test_malformed_blocks_are_refused():
    pr = W.config3(pods=50, sizes=4, seed=1)
    (blk,) = pods_to_blocks(pr.pods, 1)
    bad = dict(blk, uid=np.full(50, blk["n_strings"] + 7, dtype=np.uint32))                  # uid names no string
    with pytest.raises(S.KSolveError):
        S.PodBatch([bad])
    w = blk["spec_words"].copy(); w[int(blk["spec_off"][3])] = 0xFFFFFF                       # namespace id out of range
    with pytest.raises(S.KSolveError):
        S.PodBatch([dict(blk, spec_words=w)])
    off = blk["spec_off"].copy(); off[10] -= 2                                                # record 9 loses its tail, record 10 starts mid-record
    with pytest.raises(S.KSolveError):
        S.PodBatch([dict(blk, spec_off=off)])
    w = blk["spec_words"].copy(); w[int(blk["spec_off"][0]) + 1] = 1 << 30                    # a count that runs past the record
    with pytest.raises(S.KSolveError):
        S.PodBatch([dict(blk, spec_words=w)])
    with pytest.raises(S.KSolveError):                                                        # round 6: offsets that ascend but reach beyond the buffers the block says it has
        S.PodBatch([dict(blk, str_bytes_len=int(blk["str_off"][-1]) - 1)])
    with pytest.raises(S.KSolveError):
        S.PodBatch([dict(blk, spec_words_len=int(blk["spec_off"][-1]) - 1)])
    req = [tuple(sorted(p.containers[0].requests.items())) for p in pr.pods]
    i, j = next((i, j) for i in range(50) for j in range(i + 1, 50) if req[i] == req[j])
    dup = dict(blk, uid=np.where(np.arange(50) == j, blk["uid"][i], blk["uid"]).astype(np.uint32))   # two pods that tie on cpu / memory / timestamp share a uid: the queue order is not total
    env = S.ParsedProblem(dataclasses.replace(pr, pods=[]))
    with pytest.raises(S.KSolveError):
        S.open_batch(env, S.PodBatch([dup]))
    full = S.ParsedProblem(pr)                                                                # an environment that still carries pods of its own
    with pytest.raises(S.KSolveError):
        S.open_batch(full, S.PodBatch([blk]))


def test_empty_batch():
    pr = W.config3(pods=10, sizes=4, seed=1)
    env = S.ParsedProblem(dataclasses.replace(pr, pods=[]))
    b = S.PodBatch(pods_to_blocks([], 1))
    f = S.open_batch(env, b)
    assert f.dims["P"] == 0


@pytest.mark.gpu
def test_gpu_solve_from_batch_matches_text_route_and_oracle():
    from oracle import oracle_py as O
    for name, pr in [("config3", W.config3(pods=4000, sizes=10, seed=5)), ("fuzz7", fuzz_problem(7)), ("mid3", mid_problem(3))]:
        env, batch = _routes(pr, 2)
        fb, ms = S.solve_from_batch(env, batch, 0)
        ft, _ = S.solve_from_pods(S.ParsedProblem(pr), 0)
        want = O.solve(pr)
        assert fb.result().canonical() == ft.result().canonical() == want.canonical(), name
        assert ms["total_ms"] > 0
