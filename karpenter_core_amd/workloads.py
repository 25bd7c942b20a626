"""Seeded synthetic Solve() problems: the five BASELINE.json configs and the
reference benchmark's pod mix (scheduling_benchmark_test.go:185-288), restated.

Every generator returns a `model.Problem`; all randomness comes from numpy's legacy RandomState so a
seed pins the problem bit-for-bit on both the build container and the GPU box.
"""
from __future__ import annotations

from typing import List

import dataclasses

import numpy as np

from . import fake
from .model import (ClusterPod, Container, Expr, HostPort, LabelSelector, Pod, PodAffinityTerm, Problem, StateNode,
                    Taint, Toleration, TopologySpreadConstraint, DO_NOT_SCHEDULE, LABEL_ARCH, LABEL_CAPACITY_TYPE,
                    LABEL_HOSTNAME, LABEL_INSTANCE_TYPE, LABEL_OS, LABEL_PROVISIONER, LABEL_ZONE, NO_SCHEDULE)

CPU_CHOICES = [100, 250, 500, 1000, 1500]                 # scheduling_benchmark_test.go:285
MEM_CHOICES = [100, 256, 512, 1024, 2048, 4096]           # :280
LABEL_VALUES = ["a", "b", "c", "d", "e", "f", "g"]        # :275
ZONES = ["test-zone-1", "test-zone-2", "test-zone-3"]


def _container(rs) -> Container:
    return Container(requests={"cpu": f"{CPU_CHOICES[rs.randint(len(CPU_CHOICES))]}m",
                               "memory": f"{MEM_CHOICES[rs.randint(len(MEM_CHOICES))]}Mi"})


def _lab(rs) -> str:
    return LABEL_VALUES[rs.randint(len(LABEL_VALUES))]


def generic_pod(rs, uid) -> Pod:                           # makeGenericPods :235-250
    return Pod(uid=uid, labels={"my-label": _lab(rs)}, containers=[_container(rs)])


def spread_pod(rs, uid, key) -> Pod:                       # makeTopologySpreadPods :210-233
    return Pod(uid=uid, labels={"my-label": _lab(rs)}, containers=[_container(rs)],
               spread=[TopologySpreadConstraint(1, key, DO_NOT_SCHEDULE, LabelSelector({"my-label": _lab(rs)}))])


def affinity_pod(rs, uid, key) -> Pod:                     # makePodAffinityPods :199-208
    return Pod(uid=uid, labels={"my-affininity": _lab(rs)}, containers=[_container(rs)],
               affinity_required=[PodAffinityTerm(key, LabelSelector({"my-affininity": _lab(rs)}))])


def anti_affinity_pod(rs, uid, key, self_selecting=True) -> Pod:
    v = _lab(rs)
    sel = v if self_selecting else _lab(rs)
    return Pod(uid=uid, labels={"my-affininity": v}, containers=[_container(rs)],
               anti_required=[PodAffinityTerm(key, LabelSelector({"my-affininity": sel}))])


def diverse_pods(rs, count: int, uid_prefix="pod") -> List[Pod]:
    """makeDiversePods (:185-197): 1/7 generic, 1/7 zonal spread, 1/7 hostname spread, 1/7 hostname
    affinity, 1/7 zonal affinity, remainder generic.  UIDs are unique (the reference leaves them
    empty, which makes its queue order sort-implementation dependent -- SURVEY App. C.2)."""
    pods: List[Pod] = []
    n = count // 7
    k = [0]

    def uid():
        k[0] += 1
        return f"{uid_prefix}-{k[0]:07d}"
    pods += [generic_pod(rs, uid()) for _ in range(n)]
    pods += [spread_pod(rs, uid(), LABEL_ZONE) for _ in range(n)]
    pods += [spread_pod(rs, uid(), LABEL_HOSTNAME) for _ in range(n)]
    pods += [affinity_pod(rs, uid(), LABEL_HOSTNAME) for _ in range(n)]
    pods += [affinity_pod(rs, uid(), LABEL_ZONE) for _ in range(n)]
    pods += [generic_pod(rs, uid()) for _ in range(count - len(pods))]
    return pods


def reference_benchmark(pod_count: int, instance_count: int = 400, seed: int = 42) -> Problem:
    """benchmarkScheduler (:113-133): 1 provisioner, fake.InstanceTypes(n), makeDiversePods; run the
    oracle with inert_topology=True to mirror the reference's `&scheduling.Topology{}` (:123)."""
    rs = np.random.RandomState(seed)
    its = fake.instance_types(instance_count)
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its), limits={}, discovery_label=True)],
                   pods=diverse_pods(rs, pod_count), extra_well_known=fake.EXTRA_WELL_KNOWN)


# ---- config #1: 1k pods, 50 instance types, no affinity/topology ----
def config1(pods: int = 1000, types: int = 50, seed: int = 42) -> Problem:
    rs = np.random.RandomState(seed)
    its = fake.instance_types(types)
    ps = [Pod(uid=f"pod-{i:07d}", containers=[_container(rs)]) for i in range(pods)]
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=ps,
                   extra_well_known=fake.EXTRA_WELL_KNOWN)


TAINT_KEYS = [f"taint-{i}" for i in range(8)]


def _taint_catalogue(sizes, zone_sets, ct_sets):
    return fake.assorted_ladder(sizes, ["amd64", "arm64"], ["linux", "windows"], zone_sets, ct_sets)


# ---- config #2: 10k pods, 500 instance types, taints + nodeSelector ----
def config2(pods: int = 10_000, sizes: int = 25, seed: int = 43) -> Problem:
    rs = np.random.RandomState(seed)
    zone_sets = [[ZONES[0]], [ZONES[1]], [ZONES[2]], ZONES[:2], ZONES]
    its = _taint_catalogue(sizes, zone_sets, [["spot", "on-demand"]])        # sizes*2*2*5 types (500 at 25)
    n = len(its)
    # four tainted provisioners (weights 40..10) and one untainted catch-all; 4 of the 8 taint keys are used
    used = [int(x) for x in rs.choice(8, size=4, replace=False)]
    provs = []
    for i, tk in enumerate(used):
        taints = [Taint(TAINT_KEYS[tk], "true", NO_SCHEDULE)]
        if i == 0:
            taints.append(Taint(TAINT_KEYS[used[1]], "true", NO_SCHEDULE))
        provs.append(fake.provisioner(f"tainted-{i}", n, weight=40 - 10 * i, taints=taints))
    provs.append(fake.provisioner("default", n, weight=0))
    ps = []
    for i in range(pods):
        tol = [Toleration(key=TAINT_KEYS[k], operator="Exists", effect=NO_SCHEDULE) for k in range(8) if rs.rand() < 0.5]
        sel = {}
        if rs.rand() < 0.5:
            sel[LABEL_ARCH] = ["amd64", "arm64"][rs.randint(2)]
        if rs.rand() < 0.3:
            sel[LABEL_ZONE] = ZONES[rs.randint(3)]
        if rs.rand() < 0.2:
            sel[LABEL_CAPACITY_TYPE] = ["spot", "on-demand"][rs.randint(2)]
        ps.append(Pod(uid=f"pod-{i:07d}", labels={"my-label": _lab(rs)}, node_selector=sel, tolerations=tol,
                      containers=[_container(rs)]))
    return Problem(instance_types=its, provisioners=provs, pods=ps, extra_well_known=fake.EXTRA_WELL_KNOWN)


# ---- config #3: 100k pods, 2k instance types, topology spread + pod anti-affinity ----
def config3(pods: int = 100_000, sizes: int = 50, seed: int = 44) -> Problem:
    rs = np.random.RandomState(seed)
    zone_sets = [[ZONES[0]], [ZONES[1]], [ZONES[2]], ZONES[:2], ZONES]
    its = _taint_catalogue(sizes, zone_sets, [["spot", "on-demand"], ["on-demand"]])   # sizes*2*2*5*2 (2000 at 50)
    n = pods // 7
    ps: List[Pod] = []
    k = [0]

    def uid():
        k[0] += 1
        return f"pod-{k[0]:07d}"
    ps += [generic_pod(rs, uid()) for _ in range(n)]
    ps += [spread_pod(rs, uid(), LABEL_ZONE) for _ in range(n)]
    ps += [spread_pod(rs, uid(), LABEL_HOSTNAME) for _ in range(n)]
    ps += [anti_affinity_pod(rs, uid(), LABEL_HOSTNAME, True) for _ in range(n)]
    ps += [generic_pod(rs, uid()) for _ in range(pods - len(ps))]
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=ps,
                   extra_well_known=fake.EXTRA_WELL_KNOWN)


def hostname_herd(pods: int = 600, labels: int = 3, seed: int = 3) -> Problem:
    """Small pods that crowd the hostname-keyed groups: a third carry self-selecting hostname anti-affinity on one of `labels` values (a node takes one pod per value: the
    machines multiply, and once every node has its pod of a value the next pod of that value is taken by NOBODY), a third a hostname spread with maxSkew 1 on the
    same labels, the rest nothing.  What round 6's census of zero counters (ks_pack_rr: RRLds::hz) is about: counters leave 0 in the head window, in run steps, in
    normal rounds and when a machine opens with its first pod."""
    rs = np.random.RandomState(seed)
    its = fake.instance_types(8)
    ps: List[Pod] = []
    for i in range(pods):
        uid = f"pod-{i:07d}"
        c = Container(requests={"cpu": f"{[50, 100, 200][rs.randint(3)]}m", "memory": f"{[32, 64][rs.randint(2)]}Mi"})
        v = LABEL_VALUES[rs.randint(labels)]
        k = rs.randint(3)
        if k == 0:
            ps.append(Pod(uid=uid, labels={"my-affininity": v}, containers=[c], anti_required=[PodAffinityTerm(LABEL_HOSTNAME, LabelSelector({"my-affininity": v}))]))
        elif k == 1:
            ps.append(Pod(uid=uid, labels={"my-label": v}, containers=[c], spread=[TopologySpreadConstraint(1, LABEL_HOSTNAME, DO_NOT_SCHEDULE, LabelSelector({"my-label": v}))]))
        else:
            ps.append(Pod(uid=uid, labels={"my-label": v}, containers=[c]))
    return Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=ps, extra_well_known=fake.EXTRA_WELL_KNOWN)


# ---- config #5: 1M pods, 5k instance types, full constraint set ----
def config5(pods: int = 1_000_000, sizes: int = 50, seed: int = 46) -> Problem:
    rs = np.random.RandomState(seed)
    zone_sets = [[ZONES[0]], [ZONES[1]], [ZONES[2]], ZONES[:2], ZONES]
    its = fake.assorted_ladder(sizes, ["amd64", "arm64"], ["linux", "windows"], zone_sets,
                               [["spot", "on-demand"], ["on-demand"], ["spot"], ["on-demand", "spot"], ["on-demand"]][: max(1, 5000 // (sizes * 20))])
    n = len(its)
    provs = [fake.provisioner("high", n, weight=10, limits={"cpu": str(max(1000, pods // 8))},
                              taints=[Taint(TAINT_KEYS[0], "true", NO_SCHEDULE)]),
             fake.provisioner("default", n, weight=0)]
    ps: List[Pod] = []
    for i in range(pods):
        kind = i % 10
        uid = f"pod-{i:07d}"
        tol = [Toleration(key=TAINT_KEYS[0], operator="Exists")] if rs.rand() < 0.3 else []
        c = _container(rs)
        if kind == 0:
            p = spread_pod(rs, uid, LABEL_ZONE)
        elif kind == 1:
            p = spread_pod(rs, uid, LABEL_HOSTNAME)
        elif kind == 2:
            p = spread_pod(rs, uid, LABEL_CAPACITY_TYPE)
        elif kind == 3:
            p = anti_affinity_pod(rs, uid, LABEL_HOSTNAME, True)
        elif kind == 4:
            p = affinity_pod(rs, uid, LABEL_ZONE)
        elif kind == 5:
            p = Pod(uid=uid, labels={"my-label": _lab(rs)}, containers=[c],
                    required_affinity=[[Expr(fake.LABEL_INTEGER, "Gt", [str(2 * (1 + rs.randint(8)))])]])
        else:
            p = generic_pod(rs, uid)
        p.tolerations = tol
        if rs.rand() < 0.01:
            p.containers[0].ports = [HostPort(port=8000 + int(rs.randint(4)))]
        if kind >= 6 and rs.rand() < 0.3:
            p.node_selector = {LABEL_ARCH: ["amd64", "arm64"][rs.randint(2)]}
        ps.append(p)
    return Problem(instance_types=its, provisioners=provs, pods=ps, extra_well_known=fake.EXTRA_WELL_KNOWN)


# ---- config #4: consolidation what-ifs over one cluster snapshot ----
def cluster_snapshot(existing: int = 2048, sizes: int = 50, seed: int = 45, spare_pod_slots: int = -1):
    """E existing (owned, initialised) nodes running 8-40 pods each at 30-70 % utilisation; returns
    (instance_types, provisioner, nodes, per-node bound pods as `Pod` objects).
    spare_pod_slots >= 0: a cluster that is full by pod COUNT (max-pods): only that many nodes, picked at random, keep 1-3 free pod slots, every
    other node has none -- the pods of a removed node then mostly need a NEW node (consolidation's "replace" outcome, consolidation.go:230-274)."""
    rs = np.random.RandomState(seed)
    zone_sets = [[ZONES[0]], [ZONES[1]], [ZONES[2]], ZONES[:2], ZONES]
    its = _taint_catalogue(sizes, zone_sets, [["spot", "on-demand"], ["on-demand"]])
    prov = fake.provisioner("default", len(its))
    nodes, bound = [], []
    uid = 0
    for e in range(existing):
        ti = int(rs.randint(len(its)))
        it = its[ti]
        cpu_m = int(it.capacity["cpu"]) * 1000 - 100
        mem_mi = int(it.capacity["memory"][:-2]) * 1024 - 10
        zone = it.offerings[rs.randint(len(it.offerings))]
        arch = [r for r in it.requirements if r.key == LABEL_ARCH][0].values[0]
        os_ = [r for r in it.requirements if r.key == LABEL_OS][0].values[0]
        target = rs.uniform(0.3, 0.7)
        npods = int(rs.randint(8, 41))
        pods_here, used_cpu, used_mem = [], 0, 0
        for _ in range(npods):
            p = generic_pod(rs, f"bound-{uid:07d}")
            uid += 1
            c = int(p.containers[0].requests["cpu"][:-1])
            m = int(p.containers[0].requests["memory"][:-2])
            if used_cpu + c > target * cpu_m or used_mem + m > target * mem_mi:
                break
            used_cpu += c
            used_mem += m
            pods_here.append(p)
        name = f"node-{e:05d}"
        labels = {LABEL_PROVISIONER: "default", LABEL_INSTANCE_TYPE: it.name, LABEL_ZONE: zone.zone,
                  LABEL_CAPACITY_TYPE: zone.capacity_type, LABEL_ARCH: arch, LABEL_OS: os_, LABEL_HOSTNAME: name,
                  "karpenter.sh/initialized": "true"}
        alloc_pods = int(it.capacity["pods"])
        nodes.append(StateNode(name=name, labels=labels,
                               available={"cpu": f"{cpu_m - used_cpu}m", "memory": f"{mem_mi - used_mem}Mi",
                                          "pods": str(alloc_pods - len(pods_here))},
                               capacity=dict(it.capacity)))
        bound.append(pods_here)
    if spare_pod_slots >= 0:
        rs2 = np.random.RandomState(seed + 1000)
        keep = set(int(x) for x in rs2.choice(existing, size=min(spare_pod_slots, existing), replace=False))
        for e, n in enumerate(nodes):
            n.available = dict(n.available, pods=str(int(rs2.randint(1, 4))) if e in keep else "0")
    return its, prov, nodes, bound


def whatif(its, prov, nodes, bound, candidates: List[int], with_cluster_pods: bool = True) -> Problem:
    """simulateScheduling (deprovisioning/helpers.go:42-115): candidate nodes leave the state-node list,
    their pods become the pending batch; the cluster still holds the bound pods (excluded by UID,
    topology.go:66-70,249)."""
    cand = set(candidates)
    ns = [dataclasses.replace(n, in_state=(i not in cand)) for i, n in enumerate(nodes)]
    pods = [p for i in candidates for p in bound[i]]
    # the bound pods only matter to countDomains / inverse anti-affinity; a snapshot whose pods carry no
    # topology terms can skip listing them (nothing would be counted)
    cps = [ClusterPod(uid=p.uid, namespace=p.namespace, node_name=nodes[i].name, labels=p.labels, anti_required=list(p.anti_required))
           for i in range(len(nodes)) for p in bound[i]] if with_cluster_pods else []
    return Problem(instance_types=its, provisioners=[prov], pods=pods, nodes=ns, cluster_pods=cps,
                   extra_well_known=fake.EXTRA_WELL_KNOWN, simulation_mode=True)


def snapshot_problem(its, prov, nodes, bound, with_cluster_pods: bool = True):
    """The whole cluster as ONE problem for `scheduler.open_whatifs`: every node a state node, every bound pod in the pod
    batch (full spec); returns (problem, pod_node).  `whatif()` below builds the same what-if one problem at a time."""
    pods, pod_node = [], []
    for i in range(len(nodes)):
        for p in bound[i]:
            pods.append(p)
            pod_node.append(i)
    ns = [dataclasses.replace(n, in_state=True) for n in nodes]
    cps = [ClusterPod(uid=p.uid, namespace=p.namespace, node_name=nodes[i].name, labels=p.labels, anti_required=list(p.anti_required))
           for i in range(len(nodes)) for p in bound[i]] if with_cluster_pods else []
    return Problem(instance_types=its, provisioners=[prov], pods=pods, nodes=ns, cluster_pods=cps,
                   extra_well_known=fake.EXTRA_WELL_KNOWN, simulation_mode=True), pod_node


def config4_sets(whatifs: int = 512, existing: int = 2048, seed: int = 45) -> List[List[int]]:
    """Candidate sets of config #4: half multi-node prefixes (multinodeconsolidation.go:86-90), half singletons
    (singlenodeconsolidation.go:54)."""
    half = whatifs // 2
    rs = np.random.RandomState(seed + 1)
    return [list(range(0, i + 1)) for i in range(half)] + [[int(rs.randint(existing))] for _ in range(whatifs - half)]


def config4b_snapshot(existing: int = 2048, sizes: int = 50, seed: int = 47):
    """BASELINE configs[3]'s shape over a cluster that is full by pod count: the what-ifs REPLACE (open one node) or fail, instead of all deleting."""
    return cluster_snapshot(existing, sizes, seed, spare_pod_slots=3)      # (a handful of free pod slots in the whole cluster: nearly every what-if opens a node)


def config4(whatifs: int = 512, existing: int = 2048, sizes: int = 50, seed: int = 45, with_cluster_pods: bool = False) -> List[Problem]:
    """512 what-ifs, one Problem each (the per-what-if construction the oracle and the fingerprint tests use)."""
    its, prov, nodes, bound = cluster_snapshot(existing, sizes, seed)
    return [whatif(its, prov, nodes, bound, cs, with_cluster_pods) for cs in config4_sets(whatifs, existing, seed)]


def fresh_node(its, name, rs):
    """An empty owned node of a random instance type / offering of the catalogue, as `cluster_snapshot` makes them (a machine that just joined: cluster.go UpdateNode)."""
    it = its[int(rs.randint(len(its)))]
    off = it.offerings[int(rs.randint(len(it.offerings)))]
    arch = [r for r in it.requirements if r.key == LABEL_ARCH][0].values[0]
    os_ = [r for r in it.requirements if r.key == LABEL_OS][0].values[0]
    labels = {LABEL_PROVISIONER: "default", LABEL_INSTANCE_TYPE: it.name, LABEL_ZONE: off.zone, LABEL_CAPACITY_TYPE: off.capacity_type,
              LABEL_ARCH: arch, LABEL_OS: os_, LABEL_HOSTNAME: name, "karpenter.sh/initialized": "true"}
    return StateNode(name=name, labels=labels, capacity=dict(it.capacity),
                     available={"cpu": f"{int(it.capacity['cpu']) * 1000 - 100}m", "memory": f"{int(it.capacity['memory'][:-2]) * 1024 - 10}Mi", "pods": str(int(it.capacity["pods"]))})


def cluster_after(nodes, bound, events):
    """What a cluster (`cluster_snapshot`'s nodes / per-node bound pods) looks like after `events` (as `model.delta_to_ksd` takes them), the way state.Cluster
    keeps it (cluster.go UpdateNode / DeleteNode / UpdatePod / DeletePod; state/node.go:113,161-182: Available() = Allocatable - the requests of the pods bound):
    returns (nodes, bound, slot) -- fresh lists, removed nodes gone, `slot[i]` = the node's index in the library's snapshot (node slots are never reused: a new
    node takes the next one).  The model the tests hold ParsedProblem.apply against."""
    from .model import format_milli, parse_quantity_milli, pod_requests_milli
    live = [[dataclasses.replace(n, available=dict(n.available)), list(b), i] for i, (n, b) in enumerate(zip(nodes, bound))]
    next_slot = len(nodes)

    def find(name):
        for e in live:
            if e[0].name == name:
                return e
        raise KeyError(name)

    def adjust(n, pod, sign):
        req = pod_requests_milli(pod)
        for k in list(n.available):
            if k in req:
                n.available[k] = format_milli(parse_quantity_milli(n.available[k]) - sign * req[k])

    for ev in events:
        if ev[0] == "node+":
            live.append([dataclasses.replace(ev[1], available=dict(ev[1].available)), [], next_slot])
            next_slot += 1
        elif ev[0] == "node-":
            live.remove(find(ev[1]))
        elif ev[0] == "bind":
            e = find(ev[1])
            adjust(e[0], ev[2], +1)
            e[1].append(ev[2])
        elif ev[0] == "unbind":
            for e in live:
                hit = [p for p in e[1] if p.uid == ev[1]]
                if hit:
                    adjust(e[0], hit[0], -1)
                    e[1].remove(hit[0])
                    break
            else:
                raise KeyError(ev[1])
    return [e[0] for e in live], [e[1] for e in live], [e[2] for e in live]
