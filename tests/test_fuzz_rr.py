"""A randomised family for the kernel the HEADLINE runs on (round 6; VERDICT r05: "~16 % of the randomised evidence exercises the kernel the headline runs on").  Every
problem here stays inside what `ks_pack_rr` covers -- LEAN, no Gt/Lt, no hostname-keyed affinity, at most 64 in-flight nodes, classes of the head window's straight-line
loop -- and the test ASSERTS that `ks_pack_rr` took the Solve (started, not declined): the prepared-pod window (zonal spreads at maxSkew 1-3 through the per-domain
answers, hostname spreads and anti-affinity through the lanes' counters), the census of zero counters, RUN rounds, machines opened inside the window, questions and
hand-overs, all under one random mix.  GPU == oracle bit for bit (fingerprints from the oracle, tests/golden/make_rr_hashes.py; the quick seeds also live, field by field)."""
import numpy as np
import pytest

from karpenter_core_amd import fake, workloads as W
from karpenter_core_amd.model import (Container, DO_NOT_SCHEDULE, LABEL_ARCH, LABEL_HOSTNAME, LABEL_ZONE, LabelSelector, Pod, PodAffinityTerm, Problem, TopologySpreadConstraint)
from oracle import oracle_py as O

RR_SEEDS = list(range(32)) + [9013]      # (9013: found by round 6's last campaign -- a stale "whatever its key" read by the second window function, tests/test_rr_emulated.py)


def rr_problem(seed: int) -> Problem:
    rs = np.random.RandomState(91000 + seed)
    sizes = int(rs.randint(3, 12))
    zone_sets = [[W.ZONES[0]], [W.ZONES[1]], [W.ZONES[2]], W.ZONES[:2], W.ZONES][int(rs.randint(0, 3)):]
    its = W._taint_catalogue(sizes, zone_sets, [["spot", "on-demand"], ["on-demand"]][: 1 + seed % 2])
    npods = int(rs.randint(1500, 7001))
    nlab = int(rs.randint(2, 8))
    labels = W.LABEL_VALUES[:nlab]
    mix = rs.dirichlet(np.ones(6) * (0.6 if seed % 3 else 3.0))       # kind weights: some seeds dominated by one kind, some even
    cpus = [100, 250, 500, 1000, 1500][: int(rs.randint(2, 6))]
    mems = [100, 256, 512, 1024, 2048, 4096][: int(rs.randint(2, 7))]
    by_kind = seed % 2 == 0                                             # BASELINE's generators emit the kinds in blocks (long zonal stretches in the queue); the others interleave them
    kinds = np.sort(rs.choice(6, size=npods, p=mix)) if by_kind else rs.choice(6, size=npods, p=mix)
    pods = []
    for i in range(npods):
        own = labels[rs.randint(nlab)]
        c = Container(requests={"cpu": f"{cpus[rs.randint(len(cpus))]}m", "memory": f"{mems[rs.randint(len(mems))]}Mi"})
        p = Pod(uid=f"pod-{i:06d}", labels={"my-label": own}, containers=[c])
        sel = LabelSelector({"my-label": labels[rs.randint(nlab)]})
        k = int(kinds[i])
        if k == 0:
            p.spread = [TopologySpreadConstraint(1 + int(rs.randint(3)), LABEL_ZONE, DO_NOT_SCHEDULE, sel)]
        elif k == 1:
            p.spread = [TopologySpreadConstraint(1 + int(rs.randint(2)), LABEL_HOSTNAME, DO_NOT_SCHEDULE, sel)]
        elif k == 2:
            p.anti_required = [PodAffinityTerm(LABEL_HOSTNAME, LabelSelector({"my-label": own}))]
        elif k == 3:
            p.anti_required = [PodAffinityTerm(LABEL_HOSTNAME, sel)]
        elif k == 4:
            p.node_selector = {LABEL_ARCH: ["amd64", "arm64"][rs.randint(2)]}
        # k == 5: generic
        pods.append(p)
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=pods, extra_well_known=fake.EXTRA_WELL_KNOWN)


def fingerprints(res):
    import hashlib
    import json
    return {"sha256": hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest(),
            "reasons_sha256": hashlib.sha256(json.dumps(sorted((int(k), int(v)) for k, v in res.reasons.items())).encode()).hexdigest()}


def _gold():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rr_hashes.json")))


def test_rr_family_is_what_it_claims():
    """(CPU) the family opens machines by the hundred, mixes the kinds the head window prepares, and its goldens are there for every seed."""
    g = _gold()
    assert sorted(int(k) for k in g) == RR_SEEDS
    assert sum(v["new_nodes"] for v in g.values()) > 32 * 40
    p = rr_problem(3)
    assert not any(c.ports for q in p.pods for c in q.containers) and all(LABEL_HOSTNAME not in q.node_selector for q in p.pods)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", RR_SEEDS)
def test_gpu_matches_oracle_on_the_rr_family(seed, monkeypatch):
    from karpenter_core_amd import scheduler as S
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = rr_problem(seed)
    gold = _gold()[str(seed)]
    fp = S.FlatProblem(p)
    try:
        got = fp.solve()
        started, why = fp.rr_status()
        assert started and why == 0, (started, why)                      # ks_pack_rr took it -- the point of the family
        assert len(got.new_nodes) == gold["new_nodes"] and len(got.unscheduled) == gold["unscheduled"]
        assert fingerprints(got) == {"sha256": gold["sha256"], "reasons_sha256": gold["reasons_sha256"]}
        if gold["oracle_seconds"] < 2:
            ref = O.solve(p)
            assert got.canonical() == ref.canonical() and got.reasons == ref.reasons
        if seed % 4 == 0:                                                  # ... and ks_pack on the same problem (the kernel a decline would hand it to)
            fk = S.FlatProblem(p, flags=S.KS_FLAG_NO_RR)
            try:
                assert fingerprints(fk.solve())["sha256"] == gold["sha256"]
            finally:
                fk.close()
    finally:
        fp.close()
