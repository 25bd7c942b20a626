"""Parity proper: the HIP path (through the C ABI) against the CPU oracle, bit-identical on the canonical
result (new nodes in creation order with pod lists in commit order, instance-type option lists, request
vectors, requirement sets; existing-node pod lists; unscheduled queue; relaxation stages)."""
import numpy as np
import pytest

from helpers import solve
from karpenter_core_amd import fake, scheduler as S, workloads as W
from karpenter_core_amd.model import Problem
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def assert_same(problem, **kw):
    got = S.solve_problem(problem).canonical()
    want = O.solve(problem).canonical()
    if got != want:
        for k in want:
            if got[k] != want[k]:
                if k == "new_nodes":
                    assert len(got[k]) == len(want[k]), f"new node count {len(got[k])} != {len(want[k])}"
                    for i, (a, b) in enumerate(zip(got[k], want[k])):
                        assert a == b, f"new node {i} differs:\n gpu   {a}\n oracle {b}"
                assert got[k] == want[k], k
    return want


@pytest.mark.parametrize("pods,types,seed", [(1, 5, 1), (50, 5, 2), (1000, 50, 42), (3000, 64, 5)])
def test_config1_resources_only(pods, types, seed):
    assert_same(W.config1(pods=pods, types=types, seed=seed))


@pytest.mark.parametrize("pods,sizes,seed", [(300, 5, 1), (2000, 10, 43), (4000, 25, 44)])
def test_config2_taints_and_selectors(pods, sizes, seed):
    assert_same(W.config2(pods=pods, sizes=sizes, seed=seed))


@pytest.mark.parametrize("pods,sizes,seed", [(140, 3, 1), (700, 10, 7), (3500, 20, 44)])
def test_config3_topology(pods, sizes, seed):
    assert_same(W.config3(pods=pods, sizes=sizes, seed=seed))


@pytest.mark.parametrize("pods,seed", [(700, 1), (2100, 2)])
def test_reference_benchmark_mix(pods, seed):
    # makeDiversePods incl. pod-affinity pods, with topology live (64-size ladder keeps `integer` <= 64 values)
    rs = np.random.RandomState(seed)
    its = fake.instance_types(64)
    pr = Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its), limits={}, discovery_label=True)],
                 pods=W.diverse_pods(rs, pods), extra_well_known=fake.EXTRA_WELL_KNOWN)
    assert_same(pr)


def test_reference_benchmark_catalogue_of_400():
    """The reference benchmark's own catalogue, fake.InstanceTypes(400) (scheduling_benchmark_test.go:113-133): its `integer`
    label has 400 values, but only instance types carry it, so it never meets a node requirement and stays out of the encoding."""
    p = W.reference_benchmark(1400, instance_count=400, seed=7)
    got = S.solve_problem(p).canonical()
    assert got == O.solve(p).canonical()


@pytest.mark.parametrize("pods,seed", [(1000, 3), (4000, 46)])
def test_config5_full_constraint_set(pods, seed):
    assert_same(W.config5(pods=pods, sizes=10, seed=seed))


def test_general_kernel_variant_on_lean_problems(monkeypatch):
    """Problems without ports / hostname selectors / limits / instance-type selectors normally run the LEAN kernel
    variant; KS_NO_LEAN forces the general one, which must give the same bits."""
    monkeypatch.setenv("KS_NO_LEAN", "1")
    assert_same(W.config3(pods=700, sizes=10, seed=7))
    assert_same(W.config1(pods=400, types=20, seed=3))


def test_single_wave_kernel_on_a_multi_wave_problem(monkeypatch):
    """A single LEAN Solve normally runs the 8-wave kernel (speculation rounds + scan-ahead); KS_ONE_WAVE forces the
    single-wave variant (the one batches use), which must give the same bits."""
    monkeypatch.setenv("KS_ONE_WAVE", "1")
    assert_same(W.config3(pods=3500, sizes=20, seed=44))
    assert_same(W.config2(pods=2000, sizes=10, seed=43))


def test_to_machine_wire_format():
    """MachineTemplate.ToMachine (machinetemplate.go:77-100) over a Solve result: `instance-type In [options]` is added,
    every requirement goes out as a NodeSelectorRequirement, requests are the node's accumulated requests."""
    from karpenter_core_amd.model import Container, LABEL_INSTANCE_TYPE, LABEL_ZONE, Pod
    its = fake.default_instance_types()
    prov = fake.provisioner("default", len(its), labels={"team": "a"})
    s = S.NewScheduler([prov], its, extra_well_known=fake.EXTRA_WELL_KNOWN)
    pods = [Pod(uid="p1", node_selector={LABEL_ZONE: "test-zone-2"}, containers=[Container(requests={"cpu": "1", "memory": "100Mi"})])]
    nodes, _, err = s.Solve(pods)
    assert err is None and len(nodes) == 1
    m = nodes[0].ToMachine(prov)
    reqs = {k: (op, vals) for k, op, vals in m["requirements"]}
    assert reqs[LABEL_ZONE] == ("In", ("test-zone-2",))
    assert reqs[LABEL_INSTANCE_TYPE] == ("In", tuple(sorted(it.name for it in nodes[0].InstanceTypeOptions)))
    assert m["generateName"] == "default" and m["labels"] == {"team": "a"}
    assert m["resources"]["requests"]["cpu"] == 1000 + 0 and m["resources"]["requests"]["pods"] == 1000
    want = O.solve(s_problem(s, pods)).new_nodes[0]
    assert sorted(r.node_selector_requirement() for r in want.requirements.values() if r.key != LABEL_INSTANCE_TYPE) == \
        [r for r in m["requirements"] if r[0] != LABEL_INSTANCE_TYPE]


def s_problem(s, pods):
    return Problem(instance_types=s.instance_types, provisioners=s.provisioners, pods=list(pods), daemonset_pods=s.daemonset_pods,
                   nodes=s.state_nodes, cluster_pods=s.cluster_pods, extra_well_known=s.extra_well_known, simulation_mode=s.opts.SimulationMode)


@pytest.mark.parametrize("maker", [lambda: W.config5(pods=1200, sizes=56, seed=3), lambda: W.config5(pods=2500, sizes=60, seed=8)])
def test_more_than_4096_instance_types(maker):
    """T > 4096: a node's surviving-type mask spans two words per lane in the multi-wave kernel (4 480 / 4 800 types here; more than 64 ladder sizes would exceed the 64-values-per-key limit of the encoding)."""
    p = maker()
    assert len(p.instance_types) > 4096
    assert_same(p)


def test_whatifs_single_and_batched():
    its, prov, nodes, bound = W.cluster_snapshot(existing=96, sizes=8, seed=45)
    probs = [W.whatif(its, prov, nodes, bound, list(range(0, i + 1))) for i in range(6)] + \
            [W.whatif(its, prov, nodes, bound, [i]) for i in (7, 20, 33)]
    wants = [O.solve(p).canonical() for p in probs]
    flats = [S.FlatProblem(p) for p in probs]
    res, kms, _ = S.solve_batch(flats)
    for r, w in zip(res, wants):
        assert r.canonical() == w
    for p, w in zip(probs[:3], wants[:3]):
        assert S.solve_problem(p).canonical() == w


def test_existing_nodes_with_instance_type_selectors():
    """> 255 distinct instance-type labels on existing nodes (one node-side state each) against pods that select
    instance types with In / NotIn / a spread filtered by an instance-type node selector (pod-side columns)."""
    from karpenter_core_amd.model import (Container, Expr, LabelSelector, Pod, TopologySpreadConstraint, DO_NOT_SCHEDULE,
                                          LABEL_INSTANCE_TYPE, LABEL_ZONE)
    its, prov, nodes, bound = W.cluster_snapshot(existing=400, sizes=20, seed=9)
    base = W.whatif(its, prov, nodes, bound, list(range(0, 12)))
    rs = np.random.RandomState(5)
    names = sorted({n.labels[LABEL_INSTANCE_TYPE] for n in nodes})
    assert len(names) > 255
    # NotIn sets are drawn from a small pool: every subset of distinct NotIn requirements is a distinct node state
    notin = [[names[j] for j in rs.choice(len(names), size=40, replace=False)] for _ in range(3)]
    extra = []
    for i in range(120):
        pick = [names[j] for j in rs.choice(len(names), size=3, replace=False)]
        kind = i % 4
        c = [Container(requests={"cpu": "250m", "memory": "256Mi"} if i % 3 else {"cpu": "6", "memory": "20Gi"})]
        if kind == 0:
            extra.append(Pod(uid=f"sel-{i:04d}", labels={"app": "x"}, containers=c, node_selector={LABEL_INSTANCE_TYPE: pick[0]}))
        elif kind == 1:
            extra.append(Pod(uid=f"sel-{i:04d}", labels={"app": "x"}, containers=c, required_affinity=[[Expr(LABEL_INSTANCE_TYPE, "In", pick)]]))
        elif kind == 2:
            extra.append(Pod(uid=f"sel-{i:04d}", labels={"app": "x"}, containers=c, required_affinity=[[Expr(LABEL_INSTANCE_TYPE, "NotIn", notin[i % 3])]]))
        else:
            extra.append(Pod(uid=f"sel-{i:04d}", labels={"app": "y"}, containers=c, required_affinity=[[Expr(LABEL_INSTANCE_TYPE, "NotIn", notin[i % 2][:1])]],
                             spread=[TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, LabelSelector({"app": "y"}))]))
    base.pods = base.pods + extra
    want = assert_same(base)
    first_sel = len(base.pods) - len(extra)
    sel_on_existing = [u for v in want["existing"].values() for u in v if u >= first_sel]
    assert sel_on_existing and want["new_nodes"]


def test_feasibility_grid_matches_first_pod_option_lists():
    # grid[m][c] must equal the InstanceTypeOptions of a fresh node that receives one pod of class c
    pr = W.config2(pods=120, sizes=6, seed=9)
    fp = S.FlatProblem(pr)
    grid, ms = fp.grid()
    assert grid.shape[0] == 5
    its = pr.instance_types
    seen = 0
    for i, pod in enumerate(pr.pods[:40]):
        single = Problem(instance_types=its, provisioners=pr.provisioners, pods=[pod], extra_well_known=pr.extra_well_known)
        want = O.solve(single)
        f1 = S.FlatProblem(single)
        g1, _ = f1.grid()
        # templates are in weight order; the oracle opens the first template that works
        if not want.new_nodes:
            assert not g1.any()
            continue
        names = want.new_nodes[0].instance_types
        m = [p.name for p in sorted(pr.provisioners, key=lambda p: -p.weight)].index(want.new_nodes[0].provisioner)
        bits = [its[t].name for t in range(len(its)) if (int(g1[m, 0, t // 64]) >> (t % 64)) & 1]
        assert bits == names
        for mm in range(m):
            assert not g1[mm, 0].any()
        seen += 1
    assert seen > 10


def test_properties_at_scale_config3():
    """Size-independent properties at a size the oracle is too slow for."""
    pr = W.config3(pods=20000, sizes=50, seed=44)
    res = S.solve_problem(pr)
    placed = [i for n in res.new_nodes for i in n.pods]
    assert len(placed) == len(set(placed))
    assert sorted(placed + res.unscheduled) == list(range(len(pr.pods)))
    from karpenter_core_amd.model import pod_requests_milli, parse_quantity_milli
    alloc = {it.name: {k: parse_quantity_milli(v) - parse_quantity_milli(it.overhead.get(k, "0")) for k, v in it.capacity.items()} for it in pr.instance_types}
    for n in res.new_nodes:
        tot = {}
        for i in n.pods:
            for k, v in pod_requests_milli(pr.pods[i]).items():
                tot[k] = tot.get(k, 0) + v
        assert tot == n.requests
        assert n.instance_types, "a node with no instance type option"
        for name in n.instance_types:                      # every surviving option fits the packed requests
            assert all(v <= alloc[name].get(k, 0) for k, v in tot.items())
        # hostname anti-affinity: at most one pod per my-affininity value on a node
        vals = [pr.pods[i].labels.get("my-affininity") for i in n.pods if pr.pods[i].anti_required]
        assert len(vals) == len(set(vals))


@pytest.mark.parametrize("maker", [lambda: W.config1(pods=500, types=20, seed=3), lambda: W.config2(pods=800, sizes=6, seed=4),
                                    lambda: W.config3(pods=1400, sizes=8, seed=5)])
def test_reference_work_counters_match_oracle(maker):
    """KS_FLAG_STATS: the kernel's count of Node.Add calls / instance types the REFERENCE algorithm would
    scan (the roofline's algorithmic-bytes basis) equals the oracle's own counters."""
    pr = maker()
    want = O.solve(pr)
    got = S.solve_problem(pr, stats=True)
    assert got.canonical() == want.canonical()
    assert got.stats["queue_pops"] == want.stats["queue_pops"]
    assert got.stats["attempts"] == want.stats["attempts"]
    assert got.stats["types_scanned"] == want.stats["types_scanned"]


@pytest.mark.parametrize("case", ["config1_1k_50", "config2_10k_500", "config3_100k_2k"])
def test_full_size_configs_match_oracle_fingerprint(case):
    """BASELINE.json's full sizes: the sha256 of the canonical result equals the fingerprint the CPU oracle
    produced offline (tests/golden/make_config_hashes.py; ~2 minutes of oracle time for the 100k case)."""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_hashes.json")))[case]
    pr = {"config1_1k_50": W.config1, "config2_10k_500": W.config2, "config3_100k_2k": W.config3}[case]()
    res = S.solve_problem(pr)
    assert len(res.new_nodes) == gold["new_nodes"] and len(res.unscheduled) == gold["unscheduled"]
    assert hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest() == gold["sha256"]


def test_full_size_config3_through_ks_pack(monkeypatch):
    """BASELINE configs[2] at its full size through ks_pack's 8-wave variant (the kernel of rounds 1-3; ks_pack_rr takes this Solve by default since round 4)."""
    import hashlib
    import json
    import os
    monkeypatch.setenv("KS_NO_RR", "1")
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_hashes.json")))["config3_100k_2k"]
    res = S.solve_problem(W.config3())
    assert hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest() == gold["sha256"]
    import os as _os
    assert _os.environ.get("KS_TEST_SIM") or not res.stats.get("p22")      # (the emulator build has ks_pack_rr only)


def _golden(case):
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_hashes.json")))[case]


def _fingerprint(res):
    import hashlib
    import json
    return hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest()


def test_full_size_config4_512_whatifs_match_oracle_fingerprints():
    """BASELINE configs[3] at its stated size: 512 consolidation what-ifs (256 multi-node prefixes, 256 singletons;
    multinodeconsolidation.go:74-114, singlenodeconsolidation.go:43-78) over the 2048-node snapshot, flattened natively over one
    snapshot and solved in ONE batched launch; every what-if's canonical result has the fingerprint the CPU oracle produced
    offline from the one-problem-per-what-if construction (tests/golden/make_config_hashes.py config4_512x2048)."""
    gold = _golden("config4_512x2048")
    its, prov, nodes, bound = W.cluster_snapshot(2048, 50, 45)
    snap, pod_node = W.snapshot_problem(its, prov, nodes, bound, False)
    flats = S.open_whatifs(snap, pod_node, W.config4_sets(512, 2048, 45))
    res, _, _ = S.solve_batch(flats)
    for f in flats:
        f.close()
    assert len(res) == gold["whatifs"] == 512
    got = [_fingerprint(r) for r in res]
    bad = [i for i, (a, b) in enumerate(zip(got, gold["whatif_sha256"])) if a != b]
    assert not bad, f"what-ifs differing from the oracle: {bad[:10]}"
    assert sum(len(r.new_nodes) for r in res) == gold["new_nodes"] and sum(len(r.unscheduled) for r in res) == gold["unscheduled"]


def test_config5_at_5000_instance_types_matches_oracle_fingerprint():
    """BASELINE configs[4]'s catalogue at its stated size -- 5 000 instance types, the full constraint set (taints, Gt selectors on an
    integer label, zonal / hostname / capacity-type spread, pod affinity and anti-affinity, host ports, two weighted provisioners, one
    with a cpu limit) -- at the largest pod count the CPU oracle finishes in minutes.  Host ports and limits route the Solve through the
    general (non-LEAN, BOUNDS) 4-wave kernel."""
    gold = _golden("config5_5k_types")
    pr = W.config5(pods=gold["pods"], sizes=50, seed=46)
    assert len(pr.instance_types) == 5000
    res = S.solve_problem(pr)
    assert len(res.new_nodes) == gold["new_nodes"] and len(res.unscheduled) == gold["unscheduled"]
    assert _fingerprint(res) == gold["sha256"]


def test_full_size_config4b_replacing_whatifs_match_reference_decisions():
    """BASELINE configs[3]'s shape over a cluster that is full by pod count (workloads.config4b_snapshot): the what-ifs open nodes -- about half
    REPLACE, the long prefixes fail on "more than one node".  All 512 through `consolidation.compute_consolidations` (one batched launch, price
    stage on the device): every simulation's fingerprint, every command (action, nodes to remove, price-filtered options in order, requirements
    incl. the spot pin) and every launch-time pick equal what the CPU restatement of computeConsolidation (consolidation.go:190-274,
    helpers.go:148-157,292-315; oracle/consolidation_ref.py) produced offline."""
    import hashlib
    import json
    from karpenter_core_amd import consolidation as C
    gold = _golden("config4b_512x2048_replace")
    its, prov, nodes, bound = W.config4b_snapshot()
    sets = W.config4_sets(512, 2048, 47)
    cmds, flats, results = C.compute_consolidations(C.Snapshot(its, prov, nodes, bound), sets)
    try:
        assert gold["one_new_node"] >= 0.3 * 512 and gold["actions"].get("replace", 0) >= 64      # what-ifs that open exactly one node / that end in a replace command
        bad = [i for i, r in enumerate(results) if _fingerprint(r) != gold["whatif_sha256"][i]]
        assert not bad, f"simulations differing from the oracle: {bad[:10]}"
        got = [hashlib.sha256(json.dumps(c.canonical(), sort_keys=True, default=str).encode()).hexdigest() for c in cmds]
        bad = [i for i, (a, b) in enumerate(zip(got, gold["command_sha256"])) if a != b]
        assert not bad, f"commands differing from the reference's: {bad[:10]} e.g. {cmds[bad[0]].canonical()[:2] if bad else None}"
        acts = {}
        for c in cmds:
            acts[c.action] = acts.get(c.action, 0) + 1
        assert acts == gold["actions"]
        one = [i for i, r in enumerate(results) if len(r.new_nodes) == 1]
        picks = S.launch_pick([flats[i] for i in one], [0] * len(one))
        for i, pk in zip(one, picks):
            want = gold["launch_pick"][i]
            assert want is not None and pk is not None and its[pk[0]].name == want[0] and pk[3] == want[1], (i, pk, want)
    finally:
        for f in flats:
            f.close()
