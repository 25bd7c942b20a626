"""the snapshot kept current by EVENTS (kshost.h `ksh_env_apply`) instead of ingested again -- what state.Cluster does with UpdateNode / DeleteNode /
UpdatePod / DeletePod (reference pkg/controllers/state/cluster.go, state/node.go:113,161-182) for the snapshot consolidation simulates over
(deprovisioning/helpers.go:42-115, controller.go:64: a pass every 10 s).

CPU half (no GPU): (1) a flattening CONTINUED after events equals, byte for byte, one made from scratch over the same objects -- flat problem and the per-node tables
the device derivation reads; (2) with events that only add (no tombstones in the way) the what-ifs over the patched snapshot are, array for array, the what-ifs over a
snapshot a caller would have built fresh from the cluster as it is now (`workloads.cluster_after`, the model of state.Cluster); (3) the device derivation's CPU
restatement (`ksh_check_whatif_derivation`) stays green after every batch of events; (4) what the door refuses.  The GPU half (below, `-m gpu`) solves."""
import dataclasses

import numpy as np
import pytest

from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import Pod, Problem, StateNode, TopologySpreadConstraint, LabelSelector, DO_NOT_SCHEDULE
from karpenter_core_amd import fake


new_node = W.fresh_node


def random_events(rs, its, nodes, bound, n, tag, removes=True, make_pod=W.generic_pod):
    """n events against the cluster (nodes, bound) as it is; returns them with the cluster they lead to."""
    events = []
    for k in range(n):
        nodes, bound, _ = W.cluster_after(nodes, bound, events[-1:]) if events else (nodes, bound, None)
        kind = rs.choice(["node+", "bind", "bind", "bind"] + (["unbind", "unbind", "node-"] if removes else []))
        if kind == "node+":
            events.append(("node+", new_node(its, f"{tag}-node-{k}", rs)))
        elif kind == "node-" and len(nodes) > 4:
            events.append(("node-", nodes[int(rs.randint(len(nodes)))].name))
        elif kind == "unbind" and any(bound):
            i = int(rs.choice([j for j, b in enumerate(bound) if b]))
            events.append(("unbind", bound[i][int(rs.randint(len(bound[i])))].uid))
        else:
            events.append(("bind", nodes[int(rs.randint(len(nodes)))].name, make_pod(rs, f"{tag}-pod-{k}")))
    nodes, bound, slot = W.cluster_after(nodes, bound, events[-1:]) if events else (nodes, bound, None)
    return events, nodes, bound


def flat_hashes(parsed, pod_node, sets):
    flats = S.open_whatifs(parsed, pod_node, sets, derive=False)
    kh = S.libs()[1]
    out = [int(kh.ksh_fingerprint(f._h)) for f in flats]
    for f in flats:
        f.close()
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_continued_flattening_equals_one_from_scratch(seed):
    its, prov, nodes, bound = W.cluster_snapshot(48, 12, 100 + seed)
    snap, pn = W.snapshot_problem(its, prov, nodes, bound, False)
    parsed = S.ParsedProblem(snap)
    assert parsed.snapshot_fingerprint(pn) == parsed.snapshot_fingerprint(pn, cold=True)
    rs = np.random.RandomState(seed)
    first, continued = True, 0
    for batch in range(12):
        events, nodes, bound = random_events(rs, its, nodes, bound, int(rs.randint(1, 6)), f"s{seed}b{batch}")
        info = parsed.apply(events, pn if first else None)
        first = False
        assert info["applied"] == len(events)
        continued += info["continued"]
        assert parsed.snapshot_fingerprint() == parsed.snapshot_fingerprint(cold=True), f"batch {batch}: {[e[:2] for e in events]}"
        bind, slots = parsed.bindings()
        assert slots == info["nodes"] and len(bind) == info["pods"]
        assert int((bind >= 0).sum()) == sum(len(b) for b in bound)
    assert continued == 12          # same universes throughout: every flattening took the short road
    parsed.close()


def spread_pod(rs, uid):
    return W.spread_pod(rs, uid, W.LABEL_ZONE if rs.randint(2) else W.LABEL_HOSTNAME)


@pytest.mark.parametrize("seed,topology", [(11, False), (12, False), (13, True)])
def test_patched_snapshot_equals_a_fresh_one_when_nothing_left(seed, topology):
    """Adds only: the objects the library holds after the events are, in the same order, what a caller would have listed -- the what-ifs must come out identical."""
    its, prov, nodes, bound = W.cluster_snapshot(32, 10, 200 + seed)
    make = spread_pod if topology else W.generic_pod
    if topology:
        rs0 = np.random.RandomState(seed)
        bound = [[make(rs0, p.uid) if rs0.randint(3) == 0 else p for p in b] for b in bound]
    snap, pn = W.snapshot_problem(its, prov, nodes, bound, topology)
    parsed = S.ParsedProblem(snap)
    parsed.snapshot_fingerprint(pn)
    rs = np.random.RandomState(seed)
    all_events = []
    for batch in range(4):
        events, nodes2, bound2 = random_events(rs, its, *W.cluster_after(nodes, bound, all_events)[:2], 5, f"a{seed}b{batch}", removes=False, make_pod=make)
        parsed.apply(events, pn if batch == 0 else None)
        all_events += events
    bind, slots = parsed.bindings()
    nodes3, bound3, slot3 = W.cluster_after(nodes, bound, all_events)
    assert slot3 == list(range(len(nodes3))) and slots == len(nodes3)
    # the fresh snapshot in the library's pod order: the original pods, then the bound ones in event order
    pods, pod_node = [], []
    for i in range(len(nodes)):
        for p in bound[i]:
            pods.append(p); pod_node.append(i)
    name_to = {n.name: i for i, n in enumerate(nodes3)}
    for ev in all_events:
        if ev[0] == "bind":
            pods.append(ev[2]); pod_node.append(name_to[ev[1]])
    assert list(bind) == pod_node
    fresh = dataclasses.replace(snap, nodes=[dataclasses.replace(n, in_state=True) for n in nodes3], pods=pods,
                                cluster_pods=[W.ClusterPod(uid=p.uid, namespace=p.namespace, node_name=nodes3[pod_node[i]].name, labels=p.labels, anti_required=list(p.anti_required))
                                              for i, p in enumerate(pods)] if topology else [])
    fresh_parsed = S.ParsedProblem(fresh)
    sets = [[0], [len(nodes3) - 1], [1, 2, 3], list(range(0, len(nodes3), 3))]
    assert flat_hashes(parsed, None, sets) == flat_hashes(fresh_parsed, pod_node, sets)
    assert parsed.snapshot_fingerprint() == fresh_parsed.snapshot_fingerprint(pod_node)
    for cs in sets:
        S.check_whatif_derivation(parsed, None, cs)
    parsed.close(); fresh_parsed.close()


def test_derivation_check_after_a_thousand_events():
    """`ksh_check_whatif_derivation` green after 1 000 random deltas (topology terms in the snapshot, binds / unbinds / nodes coming and going)."""
    its, prov, nodes, bound = W.cluster_snapshot(40, 10, 77)
    rs0 = np.random.RandomState(5)
    bound = [[spread_pod(rs0, p.uid) if rs0.randint(3) == 0 else p for p in b] for b in bound]
    snap, pn = W.snapshot_problem(its, prov, nodes, bound, True)
    parsed = S.ParsedProblem(snap)
    parsed.snapshot_fingerprint(pn)
    rs = np.random.RandomState(9)
    done, first = 0, True
    while done < 1000:
        events, nodes, bound = random_events(rs, its, nodes, bound, 25, f"k{done}", make_pod=lambda r, u: spread_pod(r, u) if r.randint(3) == 0 else W.generic_pod(r, u))
        try:
            info = parsed.apply(events, pn if first else None)
        except S.KSolveError as e:
            assert "spare room" in str(e)      # (the snapshot was parsed with room for a quarter more: a caller ingests it again)
            break
        first = False
        done += len(events)
        if done % 100 == 0:
            bind, slots = parsed.bindings()
            live = [i for i in range(slots) if (bind == i).any()]
            for cs in ([live[0]], live[1:4], live[::5]):
                S.check_whatif_derivation(parsed, None, cs)
            assert parsed.snapshot_fingerprint() == parsed.snapshot_fingerprint(cold=True)
    assert done >= 1000
    parsed.close()


def test_what_the_door_refuses():
    its, prov, nodes, bound = W.cluster_snapshot(8, 6, 5)
    snap, pn = W.snapshot_problem(its, prov, nodes, bound, False)
    parsed = S.ParsedProblem(snap)
    rs = np.random.RandomState(0)
    with pytest.raises(S.KSolveError, match="needs the bindings"):
        parsed.apply([("node-", nodes[0].name)])
    with pytest.raises(S.KSolveError, match="no state node named"):
        parsed.apply([("node-", "nobody")], pn)
    with pytest.raises(S.KSolveError, match="is bound already"):
        parsed.apply([("bind", nodes[1].name, bound[2][0])])
    with pytest.raises(S.KSolveError, match="no bound pod with uid"):
        parsed.apply([("unbind", "nobody")])
    with pytest.raises(S.KSolveError, match="exists"):
        parsed.apply([("node+", dataclasses.replace(nodes[3]))])
    # events before the refused one stay applied; a name and a uid may come back
    with pytest.raises(S.KSolveError, match="event 2"):
        parsed.apply([("node-", nodes[0].name), ("unbind", bound[1][0].uid), ("unbind", bound[1][0].uid)])
    info = parsed.apply([("node+", dataclasses.replace(nodes[0])), ("bind", nodes[0].name, bound[1][0])])
    assert info["applied"] == 2 and info["nodes"] == 9
    bind, slots = parsed.bindings()
    assert slots == 9 and bind[-1] == 8 and (bind[:len(bound[0])] == -1).all()
    parsed.close()


# ---- the GPU half: the what-ifs over the patched snapshot SOLVE like the what-ifs over a snapshot built fresh from the cluster as it is now, and like the oracle ----
@pytest.mark.gpu
@pytest.mark.parametrize("seed,topology", [(21, False), (22, False), (23, True), (24, True)])
def test_whatifs_over_the_patched_snapshot_solve_like_a_fresh_one(seed, topology):
    from oracle import oracle_py as O
    its, prov, nodes0, bound0 = W.cluster_snapshot(64, 10, 300 + seed, spare_pod_slots=(6 if seed % 2 else -1))
    make = (lambda r, u: spread_pod(r, u) if r.randint(3) == 0 else W.generic_pod(r, u)) if topology else W.generic_pod
    if topology:
        rs0 = np.random.RandomState(seed)
        bound0 = [[spread_pod(rs0, p.uid) if rs0.randint(3) == 0 else p for p in b] for b in bound0]
    snap, pn = W.snapshot_problem(its, prov, nodes0, bound0, topology)
    parsed = S.ParsedProblem(snap)
    rs = np.random.RandomState(seed)
    for f in S.open_whatifs(parsed, pn, [[0], [1, 2], [5]], derive=True):      # (the snapshot is flattened and resident BEFORE the events: they continue that flattening)
        f.close()
    all_events, nodes, bound = [], nodes0, bound0
    for batch in range(3):
        events, nodes, bound = random_events(rs, its, nodes, bound, 10, f"g{seed}b{batch}", make_pod=make)
        assert parsed.apply(events, pn if batch == 0 else None)["continued"]
        all_events += events
    # the cluster as it is now, the way a caller would list it; its node j sits in the library's slot slot_of[j]
    nodes, bound, slot_of = W.cluster_after(nodes0, bound0, all_events)
    fresh_snap, fresh_pn = W.snapshot_problem(its, prov, nodes, bound, topology)
    fresh = S.ParsedProblem(fresh_snap)
    sets = [[int(x) for x in rs.choice(len(nodes), size=int(rs.choice([1, 1, 2, 4, 8])), replace=False)] for _ in range(24)]
    got_f = S.open_whatifs(parsed, None, [[slot_of[j] for j in cs] for cs in sets], derive=True)
    want_f = S.open_whatifs(fresh, fresh_pn, sets, derive=True)
    try:
        got, _, _ = S.solve_batch(got_f)
        want, _, _ = S.solve_batch(want_f)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g.canonical() == w.canonical() and g.reasons == w.reasons, (seed, i, sets[i])
        for i in range(0, len(sets), 6):
            ref = O.solve(W.whatif(its, prov, nodes, bound, sets[i], topology))
            assert got[i].canonical() == ref.canonical(), (seed, i, sets[i])
    finally:
        for f in got_f + want_f:
            f.close()
        parsed.close(); fresh.close()
