"""(GPU) The register-resident pack kernel (ks_pack_rr: it takes the single LEAN Solves it covers since round 4; KS_NO_RR=1 leaves everything to ks_pack) against
the oracle: the same C ABI, the same canonical result.  The kernel declines what it does not cover and ks_pack takes over -- `ran_rr` says which one ran."""
import hashlib
import json
import os

import pytest

from karpenter_core_amd import scheduler as S, workloads as W
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _fp(res):
    return hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest()


def ran_rr(res) -> bool:
    """ks_pack_rr leaves its round count in stats slot 8 and the pods it placed in RUN rounds in slot 22; ks_pack's multi-wave variants (not a KS_PROBES build) leave 0 there."""
    return res.stats.get("p22", 0) > 0 or res.stats.get("eq_pods", 0) > 0


def not_rr(res) -> bool:
    """The statistics are ks_pack's.  (Told from the slots above on the GPU; the emulator runs ks_pack's SINGLE-wave variants, which count other things in those slots:
    there `ksh_rr_status` alone says which kernel ran -- the tests below that hold a handle ask it.)"""
    return bool(os.environ.get("KS_TEST_SIM")) or not ran_rr(res)


@pytest.mark.parametrize("maker", [lambda: W.config1(pods=1000, types=50, seed=42), lambda: W.config3(pods=700, sizes=10, seed=7),
                                   lambda: W.config3(pods=3500, sizes=20, seed=44), lambda: W.config3(pods=20000, sizes=50, seed=45)])
def test_rr_matches_oracle(maker, monkeypatch):
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = maker()
    got = S.solve_problem(p)
    assert got.canonical() == O.solve(p).canonical()
    assert ran_rr(got)


@pytest.mark.parametrize("seed", [0, 1, 3, 5, 8, 13])
def test_rr_mid_scale_family(seed, monkeypatch):
    """The mid-scale family of the timed kernel (tests/test_fuzz_mid.py) through ks_pack_rr: in-flight nodes, two provisioners, relaxations, exact-filter winners."""
    import test_fuzz_mid as T
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = T.mid_problem(seed)
    gold = T._gold()[str(seed)]
    got = S.solve_problem(p)
    assert T.fingerprints(got) == {"sha256": gold["sha256"], "reasons_sha256": gold["reasons_sha256"]}
    assert ran_rr(got) if seed != 13 else not_rr(got)           # (13 is of the wide family and not LEAN: ks_pack's from the start)


def test_rr_full_size_config3_fingerprint(monkeypatch):
    """BASELINE configs[2] at its full size (100 000 pods / 2 000 types) through ks_pack_rr: the oracle's offline fingerprint."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_hashes.json")))["config3_100k_2k"]
    res = S.solve_problem(W.config3())
    assert len(res.new_nodes) == gold["new_nodes"] and _fp(res) == gold["sha256"]
    assert ran_rr(res)


def test_rr_twice_on_one_resident_problem(monkeypatch):
    """A resident problem solved twice: the kernel re-initialises what it caches per class (rr_memo, rr_mcnrc)."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = W.config3(pods=3500, sizes=20, seed=44)
    fp = S.FlatProblem(p)
    try:
        a = fp.solve(); b = fp.solve()
        assert a.canonical() == b.canonical() == O.solve(p).canonical()
    finally:
        fp.close()


def test_both_pack_kernels_agree(monkeypatch):
    """The same problem through ks_pack_rr and, with KS_NO_RR=1, through ks_pack: one canonical result."""
    p = W.config3(pods=20000, sizes=50, seed=46)
    monkeypatch.delenv("KS_NO_RR", raising=False)
    a = S.solve_problem(p)
    monkeypatch.setenv("KS_NO_RR", "1")
    b = S.solve_problem(p)
    assert ran_rr(a) and not_rr(b)
    assert a.canonical() == b.canonical() == O.solve(p).canonical()


# ---- what ks_pack_rr declines, and that the Solve does not notice (ks_problem_rr_status: it was launched, why it gave the Solve back; the result is ks_pack's) ----
def _solve_with_status(p):
    fp = S.FlatProblem(p)
    try:
        res = fp.solve()
        return res, fp.rr_status()
    finally:
        fp.close()


def _anti_affinity_herd(n):
    """n pods that each need a node of their own (self-selecting hostname anti-affinity on one label): n machines."""
    from karpenter_core_amd import fake
    from karpenter_core_amd.model import Container, LabelSelector, Pod, PodAffinityTerm, Problem, LABEL_HOSTNAME
    its = fake.instance_types(5)
    pods = [Pod(uid=f"pod-{i:07d}", labels={"app": "x"}, containers=[Container(requests={"cpu": "100m", "memory": "64Mi"})],
                anti_required=[PodAffinityTerm(LABEL_HOSTNAME, LabelSelector({"app": "x"}))]) for i in range(n)]
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=pods, extra_well_known=fake.EXTRA_WELL_KNOWN)


def _crowded_node(n):
    """n tiny pods and one instance type that holds them all: more pods on one node than a key's count field (1023)."""
    from karpenter_core_amd import fake
    from karpenter_core_amd.model import Container, Pod, Problem
    its = [fake.new_instance_type("huge", {"cpu": "4000", "memory": "4000Gi", "pods": str(n + 10)})]
    pods = [Pod(uid=f"pod-{i:07d}", containers=[Container(requests={"cpu": "10m", "memory": "8Mi"})]) for i in range(n)]
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=pods, extra_well_known=fake.EXTRA_WELL_KNOWN)


@pytest.mark.parametrize("name,maker,code", [
    ("more_existing_nodes_than_it_holds", lambda: W.whatif(*W.cluster_snapshot(existing=80, sizes=10, seed=5), candidates=[0, 1, 2], with_cluster_pods=False), 1),
    ("more_nodes_than_the_registers_hold", lambda: _anti_affinity_herd(3700), 4),
    ("more_pods_on_a_node_than_the_count_field", lambda: _crowded_node(1100), 2),
])
def test_rr_declines_and_ks_pack_takes_over(name, maker, code, monkeypatch):
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = maker()
    res, (started, why) = _solve_with_status(p)
    assert started and why == code, (started, why)
    assert not_rr(res)                           # (nothing of the declined run is in the statistics: they are ks_pack's)
    if len(p.pods) <= 2000:
        assert res.canonical() == O.solve(p).canonical()
    else:                                        # (the reference-shaped oracle needs minutes for thousands of one-pod nodes: ks_pack alone is the yardstick, and the shape)
        monkeypatch.setenv("KS_NO_RR", "1")
        alone, (started2, _) = _solve_with_status(p)
        assert not started2 and res.canonical() == alone.canonical()
        assert len(res.new_nodes) == len(p.pods) and all(len(n.pods) == 1 for n in res.new_nodes)


@pytest.mark.parametrize("seed,code", [(10, 3), (50, 3), (12, 7), (57, 7)])
def test_rr_declines_met_in_the_mid_scale_families(seed, code, monkeypatch):
    """Declines nobody constructed: committed seeds of tests/test_fuzz_mid.py on which ks_pack_rr starts and gives the Solve back mid-run -- 3: a pod with more than 8
    exact-filter exclusions; 7: a class outside its feature set reaches the head of the queue (the wide family's) -- and ks_pack's result is the oracle's (offline fingerprints)."""
    import test_fuzz_mid as T
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = T.mid_problem(seed)
    gold = T._gold()[str(seed)]
    res, (started, why) = _solve_with_status(p)
    assert started and why == code, (started, why)
    assert not_rr(res)
    assert T.fingerprints(res) == {"sha256": gold["sha256"], "reasons_sha256": gold["reasons_sha256"]}


def test_rr_status_says_when_it_took_the_solve(monkeypatch):
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = W.config3(pods=700, sizes=10, seed=7)
    res, (started, why) = _solve_with_status(p)
    assert started and why == 0 and ran_rr(res)
    monkeypatch.setenv("KS_NO_RR", "1")
    res, (started, why) = _solve_with_status(p)
    assert not started and not_rr(res)


def test_kernel_choice_travels_with_the_problem(monkeypatch):
    """ksolve.h KS_FLAG_NO_RR / KS_FLAG_ONE_WAVE / KS_FLAG_NO_LEAN: the choice of pack kernel as a flag of the problem (two threads sharing the library cannot use a
    process-wide environment variable); every choice gives the oracle's result."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = W.config3(pods=3500, sizes=20, seed=44)
    want = O.solve(p).canonical()
    for flags, rr in ((0, True), (S.KS_FLAG_NO_RR, False), (S.KS_FLAG_ONE_WAVE, False), (S.KS_FLAG_NO_LEAN, False)):
        fp = S.FlatProblem(p, flags=flags)
        try:
            res = fp.solve()
            assert fp.rr_status()[0] == (1 if rr else 0), flags
            assert res.canonical() == want, flags
        finally:
            fp.close()


def test_head_window_is_exercised(monkeypatch):
    """Round 5: most pods of the config #3 shape that are no plain replicas are placed by the leader's head window (stats slot 27), the plain stretches in RUN rounds (22)."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    res = S.solve_problem(W.config3(pods=20000, sizes=50, seed=45))
    assert res.stats.get("cyc_kind0", 0) > 4000 and res.stats.get("p22", 0) > 4000


@pytest.mark.parametrize("pods,labels,seed", [(2000, 3, 3), (3000, 6, 9)])
def test_census_of_zero_counters_answers_for_the_workers(pods, labels, seed, monkeypatch):
    """Round 6: hostname-keyed groups crowded until NO node takes the next anti-affinity pod -- the leader's census of zero counters (statistics slot 26) opens the
    machine without asking the workers; counters leave 0 in the head window, in run steps, in rounds and on fresh machines.  The result is the oracle's."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = W.hostname_herd(pods=pods, labels=labels, seed=seed)
    res, (started, why) = _solve_with_status(p)
    assert started and why == 0, (started, why)
    assert res.stats.get("p26", 0) > 0
    assert res.canonical() == O.solve(p).canonical()


def test_prepared_pods_reach_the_head_window(monkeypatch):
    """Round 6: the window's pods are prepared by the worker waves (RRPx); why its phases end is in statistics slot 25 (12 bits each: a plain stretch | the loop's own
    stop | the spare places taken | a topology pod nothing in the window takes | no template)."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    res = S.solve_problem(W.config3(pods=20000, sizes=50, seed=45))
    b = res.stats.get("p25", 0)
    ended = [(b >> (12 * i)) & 4095 for i in range(5)]
    assert res.stats.get("cyc_kind0", 0) > 4000 and sum(ended) == res.stats.get("cyc_kind1", 0), (ended, res.stats.get("cyc_kind1"))


def test_result_arrays_say_what_the_text_says(monkeypatch):
    """The binary result door (ksh_result_arrays_get): Node.Pods in commit order, InstanceTypeOptions, requests, requirement records, stages, unscheduled queue -- field for
    field what ksh_result_text decodes (that one is compared with the oracle everywhere else)."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = W.config3(pods=3500, sizes=20, seed=44)
    fp = S.FlatProblem(p)
    try:
        res = fp.solve()
        ra = fp.result_arrays()
        names = [it.name for it in p.instance_types]
        assert ra["n_new"] == len(res.new_nodes) and ra["n_existing"] == 0
        off, pods = ra["node_pods_off"], ra["node_pods"]
        for j, n in enumerate(res.new_nodes):
            assert list(pods[off[j]:off[j + 1]]) == list(n.pods)
            mask = ra["node_types"][j]
            assert {names[i] for i in range(len(names)) if (int(mask[i // 64]) >> (i % 64)) & 1} == set(n.instance_types)
            req = {ra["resource_names"][r]: int(ra["node_requests"][j][r]) for r in range(len(ra["resource_names"])) if (int(ra["node_requests_present"][j]) >> r) & 1}
            assert req == dict(n.requests)
            for k, key in enumerate(ra["key_names"]):
                if (int(ra["node_present"][j]) >> k) & 1:
                    q = n.requirements[key]
                    assert bool((int(ra["node_complement"][j]) >> k) & 1) == bool(q.complement)
                    assert {ra["key_value"](k, v) for v in range(64) if (int(ra["node_mask"][j][k]) >> v) & 1} == set(q.values)
                else:
                    assert key not in n.requirements
        assert list(ra["unscheduled"]) == list(res.unscheduled) and list(ra["pod_stage"]) == list(res.final_stage)
    finally:
        fp.close()
