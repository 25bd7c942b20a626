"""Consolidation decisions on top of batched what-ifs (SURVEY 8f-2 price stage, 8f-3 speculative search).

Product: every probe the reference would make is solved in ONE ks_solve_batch launch, filterByPrice / worstLaunchPrice run
in a HIP kernel over the results still on the device (ks_price_filter_dev), and the reference's binary search / first-success
scan is replayed.  Oracle: `oracle/consolidation_ref.py` follows the Go code literally, one CPU Solve per probe.
Scenarios restate pkg/controllers/deprovisioning/suite_test.go (each cites its lines); validation / TTL / PDB handling needs
the live cluster and is not part of the path."""
import numpy as np
import pytest

from karpenter_core_amd import consolidation as C, fake, workloads as W
from karpenter_core_amd.consolidation import Snapshot
from karpenter_core_amd.model import (parse_quantity_milli, Container, LABEL_ARCH, LABEL_CAPACITY_TYPE, LABEL_HOSTNAME, LABEL_INSTANCE_TYPE, LABEL_OS,
                                      LABEL_PROVISIONER, LABEL_ZONE, Offering, Pod, StateNode)
from oracle import consolidation_ref as CR


def node(name, it, capacity_type, zone, cpu="32", pods="100"):
    """test.Node with the labels the deprovisioner reads (suite_test.go:896-905), made ready the way ExpectMakeNodesReady does
    (suite_test.go:2894-2916: karpenter.sh/initialized = "true", taints cleared)."""
    labels = {LABEL_PROVISIONER: "default", LABEL_INSTANCE_TYPE: it.name, LABEL_CAPACITY_TYPE: capacity_type, LABEL_ZONE: zone,
              LABEL_HOSTNAME: name, "karpenter.sh/initialized": "true"}
    alloc = {"cpu": cpu, "pods": pods}
    return StateNode(name=name, labels=labels, available=dict(alloc), capacity=dict(alloc))


def pod(uid, cpu="1"):
    return Pod(uid=uid, labels={"app": "test"}, containers=[Container(requests={"cpu": cpu})])


def _on_demand_by_price(its):     # suite_test.go:105-121: instances with an available on-demand offering, cheapest offering ascending
    od = [i for i in its if any(o.available and o.capacity_type == "on-demand" for o in i.offerings)]
    return sorted(od, key=lambda i: min(o.price for o in i.offerings))


def most_expensive(its):
    it = _on_demand_by_price(its)[-1]
    return it, it.offerings[0]


def least_expensive(its):
    it = _on_demand_by_price(its)[0]
    return it, it.offerings[0]


def snapshot(its, nodes, bound):
    return Snapshot(its, fake.provisioner("default", len(its)), nodes, bound)


def scenarios():
    out = {}
    its = fake.instance_types_assorted()          # suite_test.go:104
    big, big_of = most_expensive(its)
    small, small_of = least_expensive(its)
    # "can replace node" suite_test.go:874-928: one pod on the most expensive instance -> a cheaper node replaces it
    out["can_replace_node"] = (snapshot(its, [node("n1", big, big_of.capacity_type, big_of.zone)], [[pod("p1")]]), [0], "replace")
    # "won't replace node if any spot replacement is more expensive" :1155-1241
    cur = fake.new_instance_type("current-on-demand", offerings=[Offering("on-demand", "test-zone-1a", 0.5, False)])
    rep = fake.new_instance_type("potential-spot-replacement", offerings=[Offering("spot", "test-zone-1a", 1.0), Offering("spot", "test-zone-1b", 0.2),
                                                                           Offering("spot", "test-zone-1c", 0.4)])
    out["spot_replacement_more_expensive"] = (snapshot([cur, rep], [node("n1", cur, "on-demand", "test-zone-1a")], [[pod("p1")]]), [0], "do-nothing")
    # "won't replace on-demand node if on-demand replacement is more expensive" :1243-1344
    rep2 = fake.new_instance_type("on-demand-replacement", offerings=[Offering("on-demand", "test-zone-1a", 0.6), Offering("on-demand", "test-zone-1b", 0.6),
                                                                       Offering("spot", "test-zone-1b", 0.2), Offering("spot", "test-zone-1c", 0.3)])
    p = pod("p1")
    p.node_selector = {LABEL_CAPACITY_TYPE: "on-demand"}          # :1306-1314 the pod insists on on-demand
    out["on_demand_replacement_more_expensive"] = (snapshot([cur, rep2], [node("n1", cur, "on-demand", "test-zone-1a")], [[p]]), [0], "do-nothing")
    # "can delete nodes" :1423-1495: two nodes, the pods of one fit on the other
    n1, n2 = node("n1", big, big_of.capacity_type, big_of.zone), node("n2", big, big_of.capacity_type, big_of.zone)
    out["can_delete_nodes"] = (snapshot(its, [n1, n2], [[pod("p1")], [pod("p2"), pod("p3")]]), [0], "delete")
    return out


def multi_scenarios():
    out = {}
    its = fake.instance_types_assorted()
    big, big_of = most_expensive(its)
    small, small_of = least_expensive(its)
    # "can merge 3 nodes into 1" :2555-2642: three expensive nodes with one pod each -> one cheaper replacement
    ns = [node(f"n{i}", big, big_of.capacity_type, big_of.zone) for i in range(3)]
    out["merge_3_into_1"] = (snapshot(its, ns, [[pod("p1")], [pod("p2")], [pod("p3")]]), [0, 1, 2], "replace", 3)
    # "won't merge 2 nodes into 1 of the same type" :2644-2719: the only candidate replacement is the type being removed
    ns = [node(f"n{i}", small, small_of.capacity_type, small_of.zone) for i in range(2)]
    out["wont_merge_same_type"] = (snapshot(its, ns, [[pod("p1")], [pod("p2"), pod("p3")]]), [0, 1], "do-nothing", 0)
    return out


def busy_cluster(existing, seed, util=(0.75, 0.98), sizes=8):
    """A snapshot tight enough that removing nodes needs replacements, with a mix of spot / on-demand nodes."""
    rs = np.random.RandomState(seed)
    its, prov, nodes, bound = W.cluster_snapshot(existing=existing, sizes=sizes, seed=seed)
    # fill the nodes up: give every node pods until it is `util` full (the generator stops at 30-70 %)
    uid = 0
    for i, n in enumerate(nodes):
        target = rs.uniform(*util)
        cap = parse_quantity_milli(n.capacity["cpu"])
        used = cap - parse_quantity_milli(n.available["cpu"])
        added = 0
        while used + 500 <= target * cap and int(n.available["pods"]) - added > 1:
            bound[i].append(Pod(uid=f"fill-{uid:06d}", labels={"my-label": "a"}, containers=[Container(requests={"cpu": "500m", "memory": "64Mi"})]))
            uid += 1
            used += 500
            added += 1
        n.available = {"cpu": f"{parse_quantity_milli(n.available['cpu']) - 500 * added}m",
                       "memory": f"{parse_quantity_milli(n.available['memory']) // 1000 // 2**20 - 64 * added}Mi",
                       "pods": str(int(n.available["pods"]) - added)}
    return Snapshot(its, prov, nodes, bound)


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the restated scenarios on the oracle (pins the restatement against the reference's expectations)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(scenarios()))
def test_oracle_single_node_scenarios(name):
    snap, cands, want = scenarios()[name]
    assert CR.compute_consolidation(snap, cands)[0] == want


@pytest.mark.parametrize("name", sorted(multi_scenarios()))
def test_oracle_multi_node_scenarios(name):
    snap, cands, want, n_removed = multi_scenarios()[name]
    cmd = CR.first_n_node_consolidation_option(snap, cands)
    assert cmd[0] == want and len(cmd[1]) == n_removed
    if name == "merge_3_into_1":
        removed_type = snap.nodes[0].labels[LABEL_INSTANCE_TYPE]
        assert removed_type not in cmd[2] and cmd[2]        # cheaper types only


def _uninitialized_neighbour():
    """simulateScheduling refuses to lean on a node that is not ready (helpers.go:102-111): the loop runs over EVERY existing node Solve
    returns (scheduler.go:132), so one uninitialised node left in the cluster turns the simulation into 'not all pods scheduled'."""
    snap, cands, _ = scenarios()["can_delete_nodes"]
    del snap.nodes[1].labels["karpenter.sh/initialized"]
    return snap, cands


def test_oracle_uninitialized_node_blocks_consolidation():
    snap, cands = _uninitialized_neighbour()
    assert CR.compute_consolidation(snap, cands)[0] == "do-nothing"


def _missing_offering():
    """getNodePrices fails for a candidate whose (capacity-type, zone) offering the instance type does not list (consolidation.go:277-287): the
    single-node scan logs the error and moves on to the next candidate (singlenodeconsolidation.go:57-60)."""
    its = fake.instance_types_assorted()
    big, big_of = most_expensive(its)
    bad = node("n0", big, big_of.capacity_type, "no-such-zone")          # needs a replacement, but its own price is unknown
    good = node("n1", big, big_of.capacity_type, big_of.zone)
    return snapshot(its, [bad, good], [[pod("p0", "33")], [pod("p1")]]), [0, 1]


def test_oracle_offering_error_skips_the_candidate():
    snap, cands = _missing_offering()
    cmd = CR.single_node_consolidation_option(snap, cands)
    assert cmd[0] in ("replace", "delete") and cmd[1] == ("n1",)


def _pending_and_deleting():
    """simulateScheduling's batch (helpers.go:76-84): pending pods + the candidates' pods + the pods of nodes that are already deleting, which are
    no state nodes either (:48-55).  Three half-empty nodes: alone, n0's pod fits elsewhere (delete); with a deleting neighbour and pending
    pods competing for the same room it does not."""
    its = fake.instance_types_assorted()
    big, big_of = most_expensive(its)
    nodes = [node(f"n{i}", big, big_of.capacity_type, big_of.zone, cpu="2") for i in range(3)]          # Available(): 2 cpu free on each
    bound = [[pod("p0", "2")], [pod("p1", "2")], [pod("p2", "2")]]
    return snapshot(its, nodes, bound)


def test_oracle_pending_pods_and_deleting_nodes():
    snap = _pending_and_deleting()
    assert CR.compute_consolidation(snap, [0])[0] == "delete"                    # p0 moves next to p1 or p2
    snap.deleting = (2,)                                                            # n2 is going away: p2 needs a home too -> n1 takes p0 XOR p2
    assert CR.compute_consolidation(snap, [0])[0] != "delete"
    with pytest.raises(ValueError):
        CR.compute_consolidation(snap, [2])                                         # errCandidateNodeDeleting
    snap.deleting = ()
    snap.pending = [pod("q0", "2"), pod("q1", "2")]                                 # the pending pods take the free halves of n1 and n2 first
    assert CR.compute_consolidation(snap, [0])[0] != "delete"
    assert CR.single_node_consolidation_option(snap, [0, 1, 2])[0] in ("do-nothing", "replace")


def test_worst_launch_price_prefers_spot_then_on_demand():
    """helpers.go:292-315 on the reference's own offering lists (suite_test.go:1166-1186, :1254-1280)."""
    from karpenter_core_amd.model import RequirementOut
    ofs = [Offering("on-demand", "a", 0.6), Offering("on-demand", "b", 0.7), Offering("spot", "b", 0.2), Offering("spot", "c", 0.3), Offering("spot", "d", 9.0, False)]
    assert CR.worst_launch_price(ofs, {}) == 0.3                                             # spot allowed: worst available spot
    only_od = {LABEL_CAPACITY_TYPE: RequirementOut(LABEL_CAPACITY_TYPE, False, ("on-demand",), None, None)}
    assert CR.worst_launch_price(ofs, only_od) == 0.7
    zone_a = {LABEL_ZONE: RequirementOut(LABEL_ZONE, False, ("a",), None, None)}
    assert CR.worst_launch_price(ofs, zone_a) == 0.6                                         # no spot in zone a -> on-demand
    nowhere = {LABEL_ZONE: RequirementOut(LABEL_ZONE, False, ("z",), None, None)}
    assert CR.worst_launch_price(ofs, nowhere) == CR.MAX_FLOAT64


def _with_zone(its, zone, most=False):          # leastExpensiveInstanceWithZone / mostExpensiveInstanceWithZone, suite_test.go:2816-2833
    od = _on_demand_by_price(its)
    for it in (reversed(od) if most else od):
        if any(o.zone == zone for o in it.offerings):
            return it
    return od[0] if most else od[-1]


def _zonal_nodes(its, most_expensive_zone=None):
    nodes = []
    for z in ("test-zone-1", "test-zone-2", "test-zone-3"):
        it = _with_zone(its, z, most=(z == most_expensive_zone))
        nodes.append(node(f"n-{z}", it, it.offerings[0].capacity_type, z, cpu="1"))
    return nodes


def test_oracle_replace_keeps_zonal_spread():
    """ "can replace node maintaining zonal topology spread" suite_test.go:1828-1934: three one-pod nodes, one per zone, the pods
    spread over zones with maxSkew 1; only the zone-2 node is expensive.  Its replacement must come up in zone 2."""
    from karpenter_core_amd.model import DO_NOT_SCHEDULE, LabelSelector, TopologySpreadConstraint
    its = fake.instance_types_assorted()
    nodes = _zonal_nodes(its, most_expensive_zone="test-zone-2")
    labels = {"app": "test-zonal-spread"}
    bound = []
    for i in range(3):
        p = Pod(uid=f"p{i}", labels=dict(labels), containers=[Container(requests={"cpu": "1"})],
                spread=[TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, LabelSelector(dict(labels)))])
        bound.append([p])
    snap = snapshot(its, nodes, bound)
    cmd = CR.single_node_consolidation_option(snap, [0, 1, 2])
    assert cmd[0] == "replace" and cmd[1] == ("n-test-zone-2",)
    reqs = dict(cmd[3])
    assert reqs[LABEL_ZONE] == (False, ("test-zone-2",), None, None)
    assert all("test-zone-2" in n for n in cmd[2])             # InstanceTypesAssorted names carry their zone


def test_oracle_anti_affinity_blocks_consolidation():
    """ "won't delete node if it would violate pod anti-affinity" suite_test.go:1936-2030: the cheapest instance in each zone,
    one pod each with required hostname anti-affinity against its own label -- nothing can be deleted or replaced cheaper."""
    from karpenter_core_amd.model import LabelSelector, PodAffinityTerm
    its = fake.instance_types_assorted()
    nodes = _zonal_nodes(its)
    labels = {"app": "test"}
    bound = [[Pod(uid=f"p{i}", labels=dict(labels), containers=[Container(requests={"cpu": "1"})],
                  anti_required=[PodAffinityTerm(LABEL_HOSTNAME, LabelSelector(dict(labels)))])] for i in range(3)]
    snap = snapshot(its, nodes, bound)
    assert CR.single_node_consolidation_option(snap, [0, 1, 2])[0] == "do-nothing"
    assert CR.first_n_node_consolidation_option(snap, [0, 1, 2])[0] == "do-nothing"


# ---------------------------------------------------------------------------------------------------------------------
# GPU: batched what-ifs + device price stage + replayed search  ==  the literal sequential reference path
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(scenarios()))
def test_gpu_single_node_scenarios(name):
    from karpenter_core_amd import consolidation as C
    snap, cands, want = scenarios()[name]
    cmds, flats, _ = C.compute_consolidations(snap, [cands])
    for f in flats:
        f.close()
    assert cmds[0].action == want
    assert cmds[0].canonical() == CR.canonical(CR.compute_consolidation(snap, cands))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(multi_scenarios()))
def test_gpu_multi_node_scenarios(name):
    from karpenter_core_amd import consolidation as C
    snap, cands, want, _ = multi_scenarios()[name]
    got = C.first_n_node_consolidation_option(snap, cands)
    assert got.action == want
    assert got.canonical() == CR.first_n_node_consolidation_option(snap, cands)


@pytest.mark.gpu
def test_gpu_uninitialized_node_and_offering_error():
    from karpenter_core_amd import consolidation as C
    snap, cands = _uninitialized_neighbour()
    cmds, flats, _ = C.compute_consolidations(snap, [cands])
    for f in flats:
        f.close()
    assert cmds[0].action == "do-nothing"
    snap, cands = _missing_offering()
    assert C.single_node_consolidation_option(snap, cands).canonical() == CR.single_node_consolidation_option(snap, cands)


@pytest.mark.gpu
def test_gpu_pending_pods_and_deleting_nodes():
    from karpenter_core_amd import consolidation as C
    for deleting, pending in (((), []), ((2,), []), ((), [pod("q0", "2"), pod("q1", "2")]), ((1,), [pod("q0", "1")])):
        snap = _pending_and_deleting()
        snap.deleting, snap.pending = deleting, pending
        cands = [[0], [1], [2], [0, 1]]
        cmds, flats, _ = C.compute_consolidations(snap, cands)
        for f in flats:
            f.close()
        for cs, cmd in zip(cands, cmds):
            try:
                want = CR.canonical(CR.compute_consolidation(snap, cs))
            except ValueError:
                assert cmd.error is not None
                continue
            assert cmd.error is None and cmd.canonical() == want
        assert C.single_node_consolidation_option(snap, [0, 1, 2]).canonical() == CR.single_node_consolidation_option(snap, [0, 1, 2])
    # busy cluster with pending pods and two deleting nodes: the searches agree with the one-Solve-per-probe restatement
    snap = busy_cluster(24, 5, util=(0.9, 0.99))
    snap.deleting = (3, 17)
    snap.pending = [pod(f"pend-{i}", "1") for i in range(6)]
    cands = [i for i in range(12) if i not in snap.deleting]
    assert C.single_node_consolidation_option(snap, cands).canonical() == CR.single_node_consolidation_option(snap, cands)
    assert C.first_n_node_consolidation_option(snap, cands, max_nodes=100).canonical() == CR.first_n_node_consolidation_option(snap, cands, 100)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 11])
def test_gpu_searches_on_a_busy_cluster(seed):
    from karpenter_core_amd import consolidation as C
    snap = busy_cluster(24, seed, util=(0.93, 0.999))        # tight: singletons come out as replace / delete / do-nothing
    rs = np.random.RandomState(seed)
    cands = [int(x) for x in rs.choice(len(snap.nodes), size=12, replace=False)]
    multi = C.first_n_node_consolidation_option(snap, cands, max_nodes=100)
    assert multi.canonical() == CR.first_n_node_consolidation_option(snap, cands, 100)
    single = C.single_node_consolidation_option(snap, cands)
    assert single.canonical() == CR.single_node_consolidation_option(snap, cands)
    # every singleton, action by action (replace / delete / do-nothing all occur on a busy cluster)
    cmds, flats, _ = C.compute_consolidations(snap, [[c] for c in cands])
    for f in flats:
        f.close()
    want = [CR.canonical(CR.compute_consolidation(snap, [c])) for c in cands]
    assert [c.canonical() for c in cmds] == want
    assert len({w[0] for w in want}) == 3


@pytest.mark.gpu
def test_gpu_multi_node_search_on_a_roomy_cluster():
    from karpenter_core_amd import consolidation as C
    snap = busy_cluster(60, 11)                               # roomier: the binary search finds a multi-node replace
    cands = [int(x) for x in np.random.RandomState(11).choice(len(snap.nodes), size=16, replace=False)]
    want = CR.first_n_node_consolidation_option(snap, cands, 100)
    assert want[0] == "replace"
    assert C.first_n_node_consolidation_option(snap, cands, max_nodes=100).canonical() == want


@pytest.mark.gpu
def test_gpu_launch_pick_and_subset():
    """The launch-time arg-min of the in-memory provider (fake/cloudprovider.go:79-84 over Offerings.Cheapest, types.go:141) and
    instanceTypesAreSubset (helpers.go:118-122) on device-resident results, against their literal restatements, for every new node of
    Solves whose nodes carry different zone / capacity-type requirements."""
    from karpenter_core_amd import scheduler as S
    for pr in (W.config2(pods=600, sizes=6, seed=5), W.config3(pods=500, sizes=6, seed=6), W.config5(pods=500, sizes=6, seed=7)):
        tindex = {it.name: i for i, it in enumerate(pr.instance_types)}
        fp = S.FlatProblem(pr)
        res = fp.solve()
        assert res.new_nodes
        picks = S.launch_pick([fp] * len(res.new_nodes), list(range(len(res.new_nodes))))
        for j, (node, got) in enumerate(zip(res.new_nodes, picks)):
            want = CR.launch_pick(pr.instance_types, node)
            if want is None:
                assert got is None
                continue
            t, zone, ct, price = got
            assert (pr.instance_types[t].name, price) == want, (j, got, want)
            it = pr.instance_types[t]
            assert any(o.available and o.zone == zone and o.capacity_type == ct and o.price == price for o in it.offerings)
        sets, wants = [], []
        for node in res.new_nodes:
            opts = [tindex[n] for n in node.instance_types]
            outsider = next(i for i in range(len(pr.instance_types)) if i not in set(opts)) if len(opts) < len(pr.instance_types) else None
            sets += [opts[: max(1, len(opts) // 2)], opts + ([outsider] if outsider is not None else [])]
            wants += [True, outsider is None]
        nodes = [j for j in range(len(res.new_nodes)) for _ in range(2)]
        assert S.types_subset([fp] * len(nodes), nodes, sets) == wants
        names = [it.name for it in pr.instance_types]
        for j, s_, w in zip(nodes, sets, wants):        # the literal restatement agrees with the expectation
            assert CR.instance_types_are_subset([names[i] for i in s_], res.new_nodes[j].instance_types) == w
        fp.close()


# ---------------------------------------------------------------------------------------------------------------------
# Drift / Expiration: the other two callers of simulateScheduling (drift.go:59-98, expiration.go:68-113)
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_simulate(snapshot, candidate_sets):
    """simulate_candidates with the oracle behind it: what the batched GPU launch computes, one Solve at a time"""
    out = []
    for cs in candidate_sets:
        if set(cs) & set(snapshot.deleting):
            out.append(None)
            continue
        sink = []
        CR.compute_consolidation(snapshot, list(cs), sink)
        out.append(sink[0])
    return out


def replacement_scenarios():
    out = {}
    its = fake.instance_types_assorted()
    big, big_of = most_expensive(its)
    # "can delete drifted nodes" suite_test.go:243-275 / "can delete expired nodes" :503-534: an empty node -> nothing to replace
    out["delete_empty"] = (snapshot(its, [node("n1", big, big_of.capacity_type, big_of.zone)], [[]]), [0], "delete", 0)
    # "can replace drifted nodes" :277-330 / "can replace node for expiration" :580-632: one pod, nowhere else to go -> one replacement
    out["replace_one"] = (snapshot(its, [node("n1", big, big_of.capacity_type, big_of.zone)], [[pod("p1")]]), [0], "replace", 1)
    # "can replace drifted nodes with multiple nodes" :332-422 / "... for expiration with multiple nodes" :725-818: three 2-cpu pods, the only launchable
    # type has 3 cpu -> three replacements (no "one node only" rule, no price stage here)
    cur = fake.new_instance_type("current-on-demand", offerings=[Offering("on-demand", "test-zone-1a", 0.5, False)])
    rep = fake.new_instance_type("replacement-on-demand", {"cpu": "3"}, offerings=[Offering("on-demand", "test-zone-1a", 0.3)])
    out["replace_with_three"] = (snapshot([cur, rep], [node("n1", cur, "on-demand", "test-zone-1a", cpu="8")], [[pod("p1", "2"), pod("p2", "2"), pod("p3", "2")]]), [0], "replace", 3)
    # "should expire one node at a time, starting with most expired" :536-578: the caller's order decides, the first candidate is the command
    n1, n2 = node("to-expire", big, big_of.capacity_type, big_of.zone), node("not-yet", big, big_of.capacity_type, big_of.zone)
    out["first_candidate_only"] = (snapshot(its, [n1, n2], [[], []]), [0, 1], "delete", 0)
    return out


@pytest.mark.parametrize("name", sorted(replacement_scenarios()))
def test_oracle_drift_and_expiration_commands(name):
    snap, cands, want, n_new = replacement_scenarios()[name]
    action, removed, replacements = CR.replacement_command(snap, cands)
    assert action == want and len(replacements) == n_new
    assert removed == [snap.nodes[cands[0]].name]
    if name == "replace_with_three":
        assert all(opts == ["replacement-on-demand"] for opts, _ in replacements)
    # the product's control flow (one batched simulation of every candidate, the same first candidate decides) over the same simulations
    assert C.replacement_command(snap, cands, simulate=_oracle_simulate) == (action, removed, replacements)


def test_a_deleting_candidate_is_passed_over():
    """errCandidateNodeDeleting -> `continue` (drift.go:73-77): the next candidate decides"""
    its = fake.instance_types_assorted()
    big, big_of = most_expensive(its)
    snap = snapshot(its, [node("n1", big, big_of.capacity_type, big_of.zone), node("n2", big, big_of.capacity_type, big_of.zone)], [[pod("p1")], []])
    snap.deleting = (0,)
    want = CR.replacement_command(snap, [0, 1])
    assert want[0] == "replace" and want[1] == ["n2"]      # n1's pod joins every simulation (helpers.go:81-84) and needs a node once n2 goes too
    assert C.replacement_command(snap, [0, 1], simulate=_oracle_simulate) == want
    snap.deleting = (0, 1)
    assert CR.replacement_command(snap, [0, 1]) == ("do-nothing", [], []) == C.replacement_command(snap, [0, 1], simulate=_oracle_simulate)


# ---------------------------------------------------------------------------------------------------------------------
# Validation.ValidateCommand: the fourth caller of simulateScheduling (validation.go:109-172)
# ---------------------------------------------------------------------------------------------------------------------
def _both_validate(snap, cmd, cands):
    a = CR.validate_command(snap, cmd.action, cmd.nodes_to_remove, cmd.replacement_types, cands)
    b = C.validate_command(snap, cmd, cands, simulate=_oracle_simulate)
    assert a == b
    return a


def test_commands_are_validated_against_the_cluster_as_it_is_now():
    """A command computed a TTL ago is simulated again before it runs ("should not consolidate if the action becomes invalid during the node TTL wait",
    suite_test.go:2338-2395, is the reference's case of it): still valid on an unchanged cluster; invalid once the simulation needs another node count, a
    node type outside the command's options, or leaves a pod unscheduled."""
    snap, cands, _ = scenarios()["can_delete_nodes"]
    cmd = C.Command(*CR.compute_consolidation(snap, cands)[:2])
    assert cmd.action == "delete" and _both_validate(snap, cmd, cands)
    snap.bound[1].extend(pod(f"late-{i}", "6") for i in range(5))        # pods arrived on the other node meanwhile: n1's pod no longer fits beside them
    snap.nodes[1].available = dict(snap.nodes[1].available, cpu="1")
    snap.bound[0][0] = pod("p1", "2")
    assert not _both_validate(snap, cmd, cands)                           # the delete would now need a new node

    snap, cands, _ = scenarios()["can_replace_node"]
    action, removed, options, reqs = CR.compute_consolidation(snap, cands)
    cmd = C.Command(action, removed, options)
    assert action == "replace" and _both_validate(snap, cmd, cands)       # the simulation lists MORE types (no price filter): the command's are a subset
    wrong = C.Command(action, removed, ["no-such-type"])
    assert not _both_validate(snap, wrong, cands)                          # a type the simulation does not offer
    assert not _both_validate(snap, C.Command("delete", removed), cands)   # expected no node, the simulation wants one
    assert not _both_validate(snap, cmd, [])                               # none of the command's nodes is a candidate any more
    snap.bound[0].append(pod("huge", "100000"))                            # a pod nothing can hold: not all pods schedule
    assert not _both_validate(snap, cmd, cands)

    snap, cands, _, _ = multi_scenarios()["merge_3_into_1"]
    got = CR.first_n_node_consolidation_option(snap, cands)
    cmd = C.Command(got[0], got[1], got[2])
    assert got[0] == "replace" and _both_validate(snap, cmd, cands)
    snap.bound[0].extend(pod(f"more-{i}", "20") for i in range(3))         # the three nodes now need more than one replacement
    assert not _both_validate(snap, cmd, cands)
