"""Randomised small problems that mix every feature the path handles -- existing nodes, weighted provisioners with limits and
taints, host ports, hostname / instance-type selectors, Gt/Lt, zonal / hostname / capacity-type spread (DoNotSchedule and
ScheduleAnyway), pod affinity and anti-affinity, preferred terms (relaxation chains), unschedulable pods, CSI volume limits.
CPU: the speculation rules hold on every case (oracle.solve_spec).  GPU: bit-identical to the oracle, through the default kernel
and through the single-wave one."""
import numpy as np
import pytest

from karpenter_core_amd import fake, workloads as W
from karpenter_core_amd.model import (Container, DO_NOT_SCHEDULE, Expr, HostPort, LABEL_ARCH, LABEL_CAPACITY_TYPE, LABEL_HOSTNAME,
                                      LABEL_INSTANCE_TYPE, LABEL_ZONE, LabelSelector, NO_SCHEDULE, Pod, PodAffinityTerm, PreferredTerm,
                                      Problem, SCHEDULE_ANYWAY, Taint, Toleration, TopologySpreadConstraint, Volume)
from oracle import oracle_py as O

SEEDS = list(range(64))


def fuzz_problem(seed: int) -> Problem:
    rs = np.random.RandomState(1000 + seed)
    sizes = int(rs.randint(3, 9))
    its, prov, nodes, bound = W.cluster_snapshot(existing=int(rs.randint(0, 14)), sizes=sizes, seed=seed) if rs.rand() < 0.7 else \
        (W.cluster_snapshot(existing=1, sizes=sizes, seed=seed)[0], None, [], [])
    n = len(its)
    provs = [fake.provisioner("default", n, weight=0)]
    if rs.rand() < 0.6:
        provs.insert(0, fake.provisioner("tainted", n, weight=10, taints=[Taint("team", "x", NO_SCHEDULE)],
                                         limits={"cpu": str(int(rs.randint(4, 40)))} if rs.rand() < 0.7 else None))
    if rs.rand() < 0.3:
        provs[-1].limits = {"cpu": str(int(rs.randint(20, 200)))}
    labels = ["a", "b", "c"]
    pods = []
    type_names = [t.name for t in its]
    for i in range(int(rs.randint(25, 160))):
        lab = {"my-label": labels[rs.randint(3)]}
        c = Container(requests={"cpu": f"{[100, 250, 500, 1000, 2000][rs.randint(5)]}m", "memory": f"{[64, 256, 1024][rs.randint(3)]}Mi"})
        p = Pod(uid=f"pod-{i:05d}", labels=lab, containers=[c])
        k = rs.randint(14)
        sel = LabelSelector({"my-label": labels[rs.randint(3)]})
        if k == 0:
            p.spread = [TopologySpreadConstraint(1 + int(rs.randint(2)), LABEL_ZONE, DO_NOT_SCHEDULE, sel)]
        elif k == 1:
            p.spread = [TopologySpreadConstraint(1, LABEL_HOSTNAME, DO_NOT_SCHEDULE, sel)]
        elif k == 2:
            p.spread = [TopologySpreadConstraint(1, LABEL_CAPACITY_TYPE, SCHEDULE_ANYWAY, sel), TopologySpreadConstraint(2, LABEL_ZONE, DO_NOT_SCHEDULE, sel)]
        elif k == 3:
            p.anti_required = [PodAffinityTerm(LABEL_HOSTNAME, sel)]
        elif k == 4:
            p.affinity_required = [PodAffinityTerm(LABEL_ZONE, sel)]
        elif k == 5:
            p.affinity_required = [PodAffinityTerm(LABEL_HOSTNAME, LabelSelector({"my-label": lab["my-label"]}))]
        elif k == 6:
            p.required_affinity = [[Expr(fake.LABEL_INTEGER, "Gt", [str(int(rs.randint(1, sizes)))])], [Expr(LABEL_ARCH, "In", ["arm64"])]]
        elif k == 7:
            p.preferred_affinity = [PreferredTerm(10, [Expr(LABEL_ZONE, "In", ["no-such-zone"])]), PreferredTerm(5, [Expr(LABEL_ARCH, "In", ["amd64"])])]
        elif k == 8:
            p.node_selector = {LABEL_INSTANCE_TYPE: type_names[rs.randint(len(type_names))]}
        elif k == 9 and nodes:
            p.node_selector = {LABEL_HOSTNAME: nodes[rs.randint(len(nodes))].name}
        elif k == 10:
            p.containers[0].ports = [HostPort(port=8000 + int(rs.randint(3)))]
        elif k == 11:
            p.containers[0].requests["cpu"] = "1000"           # fits nothing: stays unschedulable
        if rs.rand() < 0.4:
            p.tolerations = [Toleration(key="team", operator="Exists")]
        if rs.rand() < 0.2:
            p.node_selector = dict(p.node_selector, **{LABEL_ZONE: W.ZONES[rs.randint(3)]})
        pods.append(p)
    cps = []
    if nodes:
        from karpenter_core_amd.model import ClusterPod
        cps = [ClusterPod(uid=q.uid, namespace=q.namespace, node_name=nodes[i].name, labels=q.labels) for i in range(len(nodes)) for q in bound[i]]
    # volume limits (a stream of its own: the draws above stay what they were): limits on some nodes, claims already mounted, pods with
    # shared / own / ephemeral-style claims on two drivers, now and then a pod whose claim lookup failed
    rv = np.random.RandomState(7000 + seed)
    if nodes and rv.rand() < 0.6:
        nodes = [__import__("copy").deepcopy(nd) for nd in nodes]
        drivers = ["ebs.csi", "efs.csi"]
        for nd in nodes:
            if rv.rand() < 0.8:
                nd.volume_limits = {d: int(rv.randint(0, 5)) for d in drivers if rv.rand() < 0.7}
            nd.volumes = [Volume(drivers[rv.randint(2)], f"default/shared-{rv.randint(6)}") for _ in range(int(rv.randint(0, 4)))]
        for p in pods:
            r = rv.rand()
            if r < 0.35:
                p.volumes = [Volume(drivers[rv.randint(2)], f"default/shared-{rv.randint(6)}") for _ in range(int(rv.randint(1, 3)))]
                if rv.rand() < 0.5:
                    p.volumes.append(Volume(drivers[rv.randint(2)], f"default/{p.uid}-scratch"))
                if rv.rand() < 0.2:
                    p.volumes.append(Volume("unlimited.csi", "default/whatever"))
            elif r < 0.4:
                p.volume_error = True
    return Problem(instance_types=its, provisioners=provs, pods=pods, nodes=list(nodes), cluster_pods=cps, extra_well_known=fake.EXTRA_WELL_KNOWN)


@pytest.mark.parametrize("seed", SEEDS)
def test_speculation_rules_hold(seed):
    p = fuzz_problem(seed)
    want = O.solve(p).canonical()
    got, ctr = O.solve_spec(p, 7)
    assert ctr["violations"] == 0, ctr
    assert got.canonical() == want


def test_fuzz_cases_are_varied():
    res = [O.solve(fuzz_problem(s)) for s in SEEDS]
    codes = {(r.reasons[p] >> (4 * m)) & 15 for r in res for p in r.reasons for m in range(4)}
    assert {2, 4, 7} <= codes, codes            # taints, incompatible requirements, no instance type all occur as failure reasons
    assert any(r.unscheduled for r in res) and any(r.existing for r in res) and any(len(r.new_nodes) > 3 for r in res)
    assert any(max(r.final_stage, default=0) > 0 for r in res)          # some pod had to relax a preference


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_matches_oracle(seed, monkeypatch):
    from karpenter_core_amd import scheduler as S
    p = fuzz_problem(seed)
    ref = O.solve(p)
    want = ref.canonical()
    got = S.solve_problem(p)
    assert got.canonical() == want
    assert got.reasons == ref.reasons           # per-pod failure reasons (ks_result.pod_reason): one KS_WHY_* per provisioner, like scheduler.go:193-217
    monkeypatch.setenv("KS_ONE_WAVE", "1")
    got = S.solve_problem(p)
    assert got.canonical() == want and got.reasons == ref.reasons
