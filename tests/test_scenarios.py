"""The reference's envtest scenarios restated as pure Solve() fixtures.  Each test names
the reference lines it restates and asserts the same invariant the reference asserts.  Every scenario
runs against the CPU oracle here, and against the HIP path on the GPU box (-m gpu)."""
import pytest

from helpers import BACKENDS, ClusterSim, mkpod, mkpods
from karpenter_core_amd import fake
from karpenter_core_amd.model import (Container, Expr, HostPort, LabelSelector, PodAffinityTerm, PreferredTerm, Taint,
                                      Toleration, TopologySpreadConstraint, Volume, VolumeLookupError, WeightedPodAffinityTerm,
                                      resolve_pod_volumes, DO_NOT_SCHEDULE,
                                      LABEL_ARCH, LABEL_CAPACITY_TYPE, LABEL_HOSTNAME, LABEL_INSTANCE_TYPE, LABEL_OS,
                                      LABEL_ZONE, SCHEDULE_ANYWAY)

S = "pkg/controllers/provisioning/scheduling/suite_test.go"
T = "pkg/controllers/provisioning/scheduling/topology_test.go"
LABELS = {"test": "test"}
pytestmark = pytest.mark.parametrize("backend", BACKENDS)


def default_prov(**kw):
    kw.setdefault("requirements", [Expr(LABEL_CAPACITY_TYPE, "In", ["spot", "on-demand"])])
    kw.setdefault("limits", {"cpu": "2000"})
    return fake.provisioner("default", 0, discovery_label=True, **kw)


# ---------------- E1/E2/E3: labels & requirements (suite_test.go:113-553) ----------------
def test_custom_labels(backend):
    sim = ClusterSim(backend, provisioners=[default_prov(labels={"test-key": "test-value"})])
    pods = [mkpod(),                                                        # S:115 unconstrained
            mkpod(node_selector={"test-key": "different-value"}),           # S:122 conflicting
            mkpod(node_selector={"undefined": "x"}),                        # S:130 undefined key
            mkpod(required_affinity=[[Expr("test-key", "In", ["test-value", "another-value"])]]),   # S:137 matching
            mkpod(required_affinity=[[Expr("test-key", "In", ["another-value"])]])]                 # S:148 conflicting
    sim.provision(pods)
    assert [sim.scheduled(p) is not None for p in pods] == [True, False, False, True, False]
    assert sim.scheduled(pods[0]).labels["test-key"] == "test-value"


def test_well_known_labels(backend):
    # S:160 provisioner constraints
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-2"])])])
    p = mkpod()
    sim.provision([p])
    assert sim.scheduled(p).labels[LABEL_ZONE] == "test-zone-2"
    # S:168 node selectors; S:178 hostname selector never schedules; S:185 unknown zone; S:213 Gt; S:222 Lt
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1", "test-zone-2"])])])
    pods = [mkpod(node_selector={LABEL_ZONE: "test-zone-2"}),
            mkpod(node_selector={LABEL_HOSTNAME: "red-node"}),
            mkpod(node_selector={LABEL_ZONE: "unknown"}),
            mkpod(node_selector={LABEL_ZONE: "test-zone-3"}),                                      # S:194 outside provisioner
            mkpod(required_affinity=[[Expr(fake.LABEL_INTEGER, "Gt", ["8"])]]),
            mkpod(required_affinity=[[Expr(fake.LABEL_INTEGER, "Lt", ["8"])]]),
            mkpod(required_affinity=[[Expr(LABEL_ZONE, "NotIn", ["test-zone-1", "test-zone-2", "unknown"])]])]   # S:250-ish
    sim.provision(pods)
    assert [sim.scheduled(p) is not None for p in pods] == [True, False, False, False, True, True, False]
    assert sim.scheduled(pods[0]).labels[LABEL_ZONE] == "test-zone-2"
    assert sim.scheduled(pods[4]).labels[fake.LABEL_INTEGER] == "16"      # only the arm type has cpu > 8
    assert sim.scheduled(pods[5]).labels[fake.LABEL_INTEGER] in ("2", "4")


def test_operator_matrix_custom_key(backend):
    # S:400-553
    def alone(pod, **kw):          # every reference It() runs one pod against a fresh cluster
        sim = ClusterSim(backend, **kw)
        sim.provision([pod])
        return sim.scheduled(pod) is not None
    pods = [mkpod(required_affinity=[[Expr("test-key", "In", ["test-value"])]]),      # S:401 undefined key In
            mkpod(required_affinity=[[Expr("test-key", "NotIn", ["test-value"])]]),   # S:409
            mkpod(required_affinity=[[Expr("test-key", "Exists")]]),                   # S:418
            mkpod(required_affinity=[[Expr("test-key", "DoesNotExist")]])]             # S:426
    assert [alone(p) for p in pods] == [False, True, False, True]
    pods = [mkpod(required_affinity=[[Expr("test-key", "In", ["test-value"])]]),      # S:443
            mkpod(required_affinity=[[Expr("test-key", "NotIn", ["test-value"])]]),   # S:454
            mkpod(required_affinity=[[Expr("test-key", "Exists")]]),                   # S:464
            mkpod(required_affinity=[[Expr("test-key", "DoesNotExist")]]),             # S:475
            mkpod(required_affinity=[[Expr("test-key", "In", ["another-value"])]]),   # S:486
            mkpod(required_affinity=[[Expr("test-key", "NotIn", ["another-value"])]])]   # S:496
    assert [alone(p, provisioners=[default_prov(labels={"test-key": "test-value"})]) for p in pods] == [True, False, True, False, False, True]


def test_compatible_pods_share_node(backend):
    # S:507 / S:524: provisioner requirement test-key In [test-value, another-value]
    prov = default_prov(requirements=[Expr("test-key", "In", ["test-value", "another-value"])])
    sim = ClusterSim(backend, provisioners=[prov])
    a = mkpod(required_affinity=[[Expr("test-key", "In", ["test-value"])]])
    b = mkpod(required_affinity=[[Expr("test-key", "NotIn", ["another-value"])]])
    sim.provision([a, b])
    assert sim.scheduled(a).name == sim.scheduled(b).name
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr("test-key", "In", ["test-value", "another-value"])])])
    a = mkpod(required_affinity=[[Expr("test-key", "In", ["test-value"])]])
    b = mkpod(required_affinity=[[Expr("test-key", "In", ["another-value"])]])
    sim.provision([a, b])
    assert sim.scheduled(a).name != sim.scheduled(b).name


# ---------------- E4/E5: relaxation (suite_test.go:555-674) ----------------
def test_relaxation_required_terms(backend):
    prov = default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1"]), Expr(LABEL_INSTANCE_TYPE, "In", ["default-instance-type"])])
    sim = ClusterSim(backend, provisioners=[prov])
    p = mkpod(required_affinity=[[Expr(LABEL_ZONE, "In", ["invalid"])]])       # S:557 final term is never relaxed
    sim.provision([p])
    assert sim.scheduled(p) is None
    sim = ClusterSim(backend)
    p = mkpod(required_affinity=[[Expr(LABEL_ZONE, "In", ["invalid"])], [Expr(LABEL_ZONE, "In", ["invalid"])],
                                 [Expr(LABEL_ZONE, "In", ["test-zone-1"])], [Expr(LABEL_ZONE, "In", ["test-zone-2"])]])   # S:573
    res = sim.provision([p])
    assert sim.scheduled(p).labels[LABEL_ZONE] == "test-zone-1"
    assert res.final_stage == [2]


def test_relaxation_preferred_terms(backend):
    sim = ClusterSim(backend)
    p = mkpod(preferred_affinity=[PreferredTerm(1, [Expr(LABEL_ZONE, "In", ["invalid"])]),
                                  PreferredTerm(1, [Expr(LABEL_INSTANCE_TYPE, "In", ["invalid"])])])   # S:597
    sim.provision([p])
    assert sim.scheduled(p) is not None
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1", "test-zone-2"])])])
    p = mkpod(preferred_affinity=[PreferredTerm(100, [Expr(LABEL_INSTANCE_TYPE, "In", ["test-zone-3"])]),
                                  PreferredTerm(50, [Expr(LABEL_ZONE, "In", ["test-zone-2"])]),
                                  PreferredTerm(1, [Expr(LABEL_ZONE, "In", ["test-zone-1"])])])         # S:616
    sim.provision([p])
    assert sim.scheduled(p).labels[LABEL_ZONE] == "test-zone-2"
    sim = ClusterSim(backend)
    p = mkpod(preferred_affinity=[PreferredTerm(1, [Expr(LABEL_ZONE, "NotIn", ["test-zone-3"])])],
              required_affinity=[[Expr(LABEL_ZONE, "In", ["test-zone-3"])]])                             # S:643
    sim.provision([p])
    assert sim.scheduled(p).labels[LABEL_ZONE] == "test-zone-3"
    sim = ClusterSim(backend)
    p = mkpod(preferred_affinity=[PreferredTerm(1, [Expr(LABEL_ZONE, "In", ["invalid"]), Expr(LABEL_ZONE, "NotIn", ["invalid"])])])   # S:664
    sim.provision([p])
    assert sim.scheduled(p) is not None


def test_prefer_no_schedule_toleration_is_last_resort(backend):
    # provisioning/suite_test.go:1061-1126: a PreferNoSchedule taint is tolerated only after relaxation
    prov = default_prov(taints=[Taint("foo", "bar", "PreferNoSchedule")])
    sim = ClusterSim(backend, provisioners=[prov])
    p = mkpod()
    res = sim.provision([p])
    assert sim.scheduled(p) is not None and res.final_stage == [1]


# ---------------- E6: instance type compatibility (suite_test.go:676-919) ----------------
def test_instance_type_compatibility(backend):
    sim = ClusterSim(backend)
    p = mkpod(requests={"memory": "2Ti"})                                   # S:677
    sim.provision([p])
    assert sim.scheduled(p) is None
    sim = ClusterSim(backend)
    a, b = mkpod(node_selector={LABEL_ARCH: "arm64"}), mkpod(node_selector={LABEL_ARCH: "amd64"})   # S:688
    sim.provision([a, b])
    assert sim.scheduled(a).name != sim.scheduled(b).name
    assert sim.node_types[sim.scheduled(a).name] == "arm-instance-type"
    sim = ClusterSim(backend)
    a = mkpod(requests={fake.RES_GPU_A: "1"}, limits={fake.RES_GPU_A: "1"})
    b = mkpod(requests={fake.RES_GPU_B: "1"}, limits={fake.RES_GPU_B: "1"})        # S:820
    c = mkpod(requests={fake.RES_GPU_A: "1", fake.RES_GPU_B: "1"})                 # S:846 no single type has both
    sim.provision([a, b, c])
    assert sim.scheduled(a).name != sim.scheduled(b).name and sim.scheduled(c) is None
    sim = ClusterSim(backend)
    a = mkpod(node_selector={LABEL_OS: "ios"})
    sim.provision([a])
    assert sim.node_types[sim.scheduled(a).name] == "arm-instance-type"            # S:727


# ---------------- E7: host ports (suite_test.go:921-1076) ----------------
def test_host_ports(backend):
    def two(p1, p2):
        sim = ClusterSim(backend)
        a, b = mkpod(ports=p1), mkpod(ports=p2)
        sim.provision([a, b])
        return sim.scheduled(a).name == sim.scheduled(b).name
    assert not two([HostPort(80)], [HostPort(80)])                                        # S:923
    assert not two([HostPort(80, "UDP")], [HostPort(80, "UDP")])                          # S:940
    assert not two([HostPort(80, "TCP", "1.2.3.4")], [HostPort(80, "TCP", "1.2.3.4")])    # S:958
    assert not two([HostPort(80, "TCP", "1.2.3.4")], [HostPort(80, "TCP", "0.0.0.0")])    # S:976
    assert two([HostPort(80, "TCP")], [HostPort(80, "UDP")])                              # S:1019
    assert two([HostPort(80, "TCP", "1.2.3.4")], [HostPort(80, "TCP", "1.2.3.5")])        # S:1038
    assert two([], [])                                                                    # S:1058
    # S:996 existing node variant: second batch must not land on the node holding 0.0.0.0:80
    sim = ClusterSim(backend)
    a = mkpod(ports=[HostPort(80, "TCP", "1.2.3.4")])
    sim.provision([a])
    b = mkpod(ports=[HostPort(80, "TCP", "0.0.0.0")])
    sim.provision([b])
    assert sim.scheduled(a).name != sim.scheduled(b).name


# ---------------- E8: bin packing (suite_test.go:1078-1341) ----------------
def test_binpacking_small_pods(backend):
    sim = ClusterSim(backend)
    p = mkpod(requests={"memory": "100M"})                                                # S:1079
    sim.provision([p])
    assert sim.node_types[sim.scheduled(p).name] == "small-instance-type"
    sim = ClusterSim(backend)
    pods = mkpods(5, requests={"memory": "10M"})                                          # S:1101
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 1
    assert sim.node_types[sim.scheduled(pods[0]).name] == "small-instance-type"


def test_binpacking_40_large_pods_make_20_nodes(backend):
    sim = ClusterSim(backend)
    pods = mkpods(40, requests={"memory": "1.8G"}, node_selector={LABEL_ARCH: "amd64"})   # S:1119
    res = sim.provision(pods)
    names = {sim.scheduled(p).name for p in pods}
    assert len(names) == 20 and all(sim.node_types[n] == "default-instance-type" for n in names)
    # SURVEY App. F.1 hand trace: every node holds 2 pods and keeps {default, gpu-vendor, gpu-vendor-b}
    for n in res.new_nodes:
        assert len(n.pods) == 2
        assert n.instance_types == ["default-instance-type", "gpu-vendor-instance-type", "gpu-vendor-b-instance-type"]
        assert n.requests == {"memory": 3_600_000_000_000, "pods": 2000}


def test_binpacking_small_and_large_together(backend):
    sim = ClusterSim(backend)
    pods = mkpods(40, requests={"memory": "1.8G"}, node_selector={LABEL_ARCH: "amd64"}) + \
        mkpods(20, requests={"memory": "400M"}, node_selector={LABEL_ARCH: "amd64"})      # S:1138
    sim.provision(pods)
    names = {sim.scheduled(p).name for p in pods}
    assert len(names) == 20 and all(sim.node_types[n] == "default-instance-type" for n in names)


def test_binpacking_tight_and_edge_cases(backend):
    sim = ClusterSim(backend, instance_types=fake.instance_types(5))
    a, b = mkpod(requests={"cpu": "4.5"}), mkpod(requests={"cpu": "1"})                   # S:1170
    sim.provision([a, b])
    assert sim.scheduled(a).name != sim.scheduled(b).name
    assert sim.node_types[sim.scheduled(a).name] != sim.node_types[sim.scheduled(b).name]
    sim = ClusterSim(backend)
    p = mkpod(requests={"foo.com/weird-resources": "0"}, limits={"foo.com/weird-resources": "0"})   # S:1193
    sim.provision([p])
    assert sim.scheduled(p) is not None
    sim = ClusterSim(backend)
    p = mkpod(requests={"memory": "2Ti"})                                                 # S:1205
    sim.provision([p])
    assert sim.scheduled(p) is None
    sim = ClusterSim(backend)
    pods = mkpods(25, requests={"memory": "1m", "cpu": "1m"}, node_selector={LABEL_ARCH: "amd64"})   # S:1215 pods/node cap
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 5


def test_binpacking_init_containers(backend):
    sim = ClusterSim(backend)
    p = mkpod(requests={"cpu": "1", "memory": "1Gi"}, init_containers=[Container(requests={"cpu": "10", "memory": "2Gi"})])   # S:1236
    sim.provision([p])
    assert sim.node_types[sim.scheduled(p).name] == "arm-instance-type"
    sim = ClusterSim(backend)
    p = mkpod(requests={"cpu": "1"}, init_containers=[Container(requests={"cpu": "10000"})])                                # S:1256
    sim.provision([p])
    assert sim.scheduled(p) is None


def test_option_set_is_price_agnostic(backend):
    # S:1275-1341: Solve returns every valid type; price is applied later by the provider
    its = [fake.new_instance_type("small", {"cpu": "1", "memory": "1Gi"}),
           fake.new_instance_type("medium", {"cpu": "2", "memory": "2Gi"}),
           fake.new_instance_type("large", {"cpu": "4", "memory": "4Gi"})]
    for it, price in zip(its, (3.0, 2.0, 1.0)):
        for o in it.offerings:
            o.price = price
    sim = ClusterSim(backend, instance_types=its)
    p = mkpod(requests={"cpu": "1m", "memory": "1Mi"})
    res = sim.provision([p])
    assert res.new_nodes[0].instance_types == ["small", "medium", "large"]
    assert sim.node_types[sim.scheduled(p).name] == "large"


# ---------------- E9/E10: in-flight nodes (suite_test.go:1343-1531) ----------------
def test_inflight_node_reuse(backend):
    sim = ClusterSim(backend)
    a = mkpod(requests={"cpu": "10m"})
    sim.provision([a])                                                                     # S:1344
    b = mkpod(requests={"cpu": "10m"})
    sim.provision([b])
    assert sim.scheduled(a).name == sim.scheduled(b).name
    c = mkpod(requests={"cpu": "10m"}, node_selector={LABEL_ZONE: "test-zone-3"})          # S:1422 incompatible with zone-1 node
    sim.provision([c])
    assert sim.scheduled(c).name != sim.scheduled(a).name
    d = mkpod(requests={"cpu": "1.9"}, node_selector={LABEL_ARCH: "amd64"})                # S:1405 won't fit on what is left of the 2-cpu node
    sim.provision([d])
    assert sim.scheduled(d).name != sim.scheduled(a).name


def test_inflight_zonal_spread(backend):
    # S:1460: pods bound to in-flight nodes count for later batches
    topo = [TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, LabelSelector(dict(LABELS)))]
    sim = ClusterSim(backend)
    sim.provision([mkpod(labels=LABELS, spread=topo, requests={"cpu": "1.1"})])
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1]
    sim.provision([mkpod(labels=LABELS, spread=topo, requests={"cpu": "1.1"})])
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1]
    sim.provision(mkpods(4, labels=LABELS, spread=topo, requests={"cpu": "1.1"}))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [2, 2, 2]


# ---------------- E12: daemonset overhead (suite_test.go:1660-1822, provisioning/suite_test.go:360-528) -------------
def test_daemonset_overhead(backend):
    ds = mkpod(requests={"cpu": "1", "memory": "1Gi"})
    sim = ClusterSim(backend, daemonsets=[ds])
    p = mkpod(requests={"cpu": "1", "memory": "1Gi"})
    res = sim.provision([p])
    assert res.new_nodes[0].requests == {"cpu": 2000, "memory": 2 * 2**30 * 1000, "pods": 2000}
    # a daemonset that does not tolerate the provisioner's taint is ignored (provisioning/suite_test.go:480-528)
    sim = ClusterSim(backend, provisioners=[default_prov(taints=[Taint("foo", "bar", "NoSchedule")])], daemonsets=[ds])
    p = mkpod(requests={"cpu": "1"}, tolerations=[Toleration(operator="Exists")])
    res = sim.provision([p])
    assert res.new_nodes[0].requests == {"cpu": 1000, "pods": 1000}


# ---------------- E13: limits (provisioning/suite_test.go:237-357) ----------------
def test_provisioner_limits(backend):
    sim = ClusterSim(backend, provisioners=[default_prov(limits={"cpu": "2"})])
    p = mkpod(requests={"cpu": "1.75"})                  # only the 2-cpu type stays under the limit, and 1.75 > 1.9 allocatable? no: fits
    sim.provision([p])
    assert sim.scheduled(p) is not None and sim.node_types[sim.scheduled(p).name] == "small-instance-type"
    sim = ClusterSim(backend, provisioners=[default_prov(limits={"cpu": "3"})])
    pods = mkpods(2, requests={"cpu": "1.5"})            # P:310-ish: exactly one schedules (subtractMax pessimism)
    sim.provision(pods)
    assert sum(sim.scheduled(p) is not None for p in pods) == 1
    sim = ClusterSim(backend, provisioners=[default_prov(limits={"cpu": "0"})])
    p = mkpod()
    sim.provision([p])
    assert sim.scheduled(p) is None


# ---------------- E14: zonal spread (topology_test.go:66-377) ----------------
def spread(key, max_skew=1, labels=LABELS, when=DO_NOT_SCHEDULE):
    return [TopologySpreadConstraint(max_skew, key, when, LabelSelector(dict(labels)))]


def test_unknown_topology_key(backend):
    sim = ClusterSim(backend)
    a, b = mkpod(labels=LABELS, spread=spread("unknown")), mkpod()        # T:38
    sim.provision([a, b])
    assert sim.scheduled(a) is None and sim.scheduled(b) is not None


def test_zonal_spread_basic(backend):
    topo = spread(LABEL_ZONE)
    sim = ClusterSim(backend)
    sim.provision(mkpods(4, labels=LABELS, spread=topo))                  # T:66
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1, 2]


def test_zonal_spread_respects_provisioner_zones(backend):
    topo = spread(LABEL_ZONE)
    prov = default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1", "test-zone-2", "test-zone-3"])])   # T:106
    sim = ClusterSim(backend, provisioners=[prov])
    sim.provision(mkpods(4, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1, 2]
    # derived (not a reference assertion): with the provisioner limited to two zones the third zone is
    # still a count-0 domain of the group (universe = instance-type zones, provisioner.go:267-271), the
    # pods allow it, so min stays 0 and only one pod per reachable zone can schedule (topologygroup.go:184-200)
    prov = default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1", "test-zone-2"])])
    sim = ClusterSim(backend, provisioners=[prov])
    sim.provision(mkpods(4, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1]


def test_zonal_spread_recovers_skew(backend):
    topo = spread(LABEL_ZONE)                                             # T:205-241
    sim = ClusterSim(backend)
    pods = mkpods(9, labels=LABELS, spread=topo, requests={"cpu": "1.1"})
    sim.provision(pods)
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [3, 3, 3]
    for p in pods:
        if sim.scheduled(p).labels[LABEL_ZONE] != "test-zone-1":
            sim.delete_pod(p)
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [3]
    sim.provision(mkpods(3, labels=LABELS, spread=topo, requests={"cpu": "1.1"}))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 2, 3]


def test_zonal_spread_do_not_schedule_cap(backend):
    topo = spread(LABEL_ZONE)                                             # T:243-274
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1"])])])
    sim.provision([mkpod(labels=LABELS, spread=topo, requests={"cpu": "1.1"})])
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1]
    sim.provisioners[0].requirements = [Expr(LABEL_ZONE, "In", ["test-zone-2", "test-zone-3"])]
    pods = mkpods(10, labels=LABELS, spread=topo, requests={"cpu": "1.1"})
    sim.provision(pods)
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 2, 2]
    assert sum(sim.scheduled(p) is None for p in pods) == 6


def test_zonal_spread_nil_selector_and_non_self_selecting(backend):
    sim = ClusterSim(backend)
    topo = [TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, None)]      # T:341 nil selector: nothing counts
    pods = mkpods(5, labels=LABELS, spread=topo)
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 1
    sim = ClusterSim(backend)
    topo = spread(LABEL_ZONE)                                                     # T:353: pods do not match their own selector
    pods = mkpods(5, spread=topo)
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 1


# ---------------- E15: hostname spread (topology_test.go:380-489) ----------------
def test_hostname_spread(backend):
    topo = spread(LABEL_HOSTNAME)
    sim = ClusterSim(backend)
    sim.provision(mkpods(4, labels=LABELS, spread=topo))                  # T:380
    assert sim.skew(LABEL_HOSTNAME, topo[0].label_selector) == [1, 1, 1, 1]
    topo = spread(LABEL_HOSTNAME, 4)
    sim = ClusterSim(backend)
    sim.provision(mkpods(4, labels=LABELS, spread=topo))                  # T:396
    assert sim.skew(LABEL_HOSTNAME, topo[0].label_selector) == [4]


# ---------------- E16: capacity-type spread + ScheduleAnyway (topology_test.go:492-782) ----------------
def test_capacity_type_spread(backend):
    topo = spread(LABEL_CAPACITY_TYPE)
    sim = ClusterSim(backend)
    sim.provision(mkpods(4, labels=LABELS, spread=topo))                  # T:493
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [2, 2]


def test_schedule_anyway_relaxes(backend):
    topo = spread(LABEL_CAPACITY_TYPE, when=SCHEDULE_ANYWAY)              # T:561-590
    prov = default_prov(requirements=[Expr(LABEL_CAPACITY_TYPE, "In", ["spot"])])
    sim = ClusterSim(backend, provisioners=[prov])
    sim.provision([mkpod(labels=LABELS, spread=topo, requests={"cpu": "1.1"})])
    sim.provisioners[0].requirements = [Expr(LABEL_CAPACITY_TYPE, "In", ["on-demand"])]
    pods = mkpods(5, labels=LABELS, spread=topo, requests={"cpu": "1.1"})
    sim.provision(pods)
    assert all(sim.scheduled(p) is not None for p in pods)
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [1, 5]


# ---------------- E17: combined hostname + zone (topology_test.go:785-1028) ----------------
def test_combined_hostname_and_zone_spread(backend):
    topo = spread(LABEL_ZONE) + spread(LABEL_HOSTNAME, 3)                 # T:786
    sim = ClusterSim(backend)
    sim.provision(mkpods(2, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1]
    sim.provision(mkpods(3, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 2, 2]
    assert max(sim.skew(LABEL_HOSTNAME, topo[1].label_selector)) <= 3


# ---------------- E18: spread with node affinity (topology_test.go:1031-1193) ----------------
def test_spread_limited_by_node_selector(backend):
    topo = spread(LABEL_ZONE)                                             # T:1032
    sim = ClusterSim(backend)
    sim.provision(mkpods(1, labels=LABELS, spread=topo, node_selector={LABEL_ZONE: "test-zone-1"}) +
                  mkpods(1, labels=LABELS, spread=topo, node_selector={LABEL_ZONE: "test-zone-2"}))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1]
    pods = mkpods(6, labels=LABELS, spread=topo,
                  required_affinity=[[Expr(LABEL_ZONE, "In", ["test-zone-1", "test-zone-2"])]])
    sim.provision(pods)
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [4, 4]


# ---------------- E19: pod affinity (topology_test.go:1205-1443) ----------------
def aff(key, labels, ns=None):
    return [PodAffinityTerm(key, LabelSelector(dict(labels)), list(ns or []))]


def test_pod_affinity_hostname(backend):
    lab = {"security": "s2"}
    sim = ClusterSim(backend)
    target = mkpod(labels=lab)
    follower = mkpod(affinity_required=aff(LABEL_HOSTNAME, lab))          # T:1206
    sim.provision([follower, target])
    assert sim.scheduled(target).name == sim.scheduled(follower).name
    sim = ClusterSim(backend)
    pods = mkpods(3, labels=lab, affinity_required=aff(LABEL_HOSTNAME, lab))   # T:1283 self affinity
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 1


def test_self_affinity_first_empty_domain_only(backend):
    lab = {"security": "s2"}                                               # T:1306-1344
    sim = ClusterSim(backend)
    pods = mkpods(10, labels=lab, affinity_required=aff(LABEL_HOSTNAME, lab))
    sim.provision(pods)
    placed = [p for p in pods if sim.scheduled(p) is not None]
    assert len(placed) == 5 and len({sim.scheduled(p).name for p in placed}) == 1
    more = mkpods(10, labels=lab, affinity_required=aff(LABEL_HOSTNAME, lab))
    sim.provision(more)
    assert all(sim.scheduled(p) is None for p in more)


def test_pod_affinity_zone_and_missing_target(backend):
    lab = {"security": "s2"}
    sim = ClusterSim(backend)
    target = mkpod(labels=lab, node_selector={LABEL_ZONE: "test-zone-3"}, requests={"cpu": "2"})
    followers = mkpods(3, affinity_required=aff(LABEL_ZONE, lab))        # T:1230-ish: followers land in the target's zone
    sim.provision(followers + [target])
    assert all(sim.scheduled(p).labels[LABEL_ZONE] == "test-zone-3" for p in followers)
    sim = ClusterSim(backend)
    p = mkpod(affinity_required=aff(LABEL_ZONE, lab))                     # T:1924 no target anywhere
    sim.provision([p])
    assert sim.scheduled(p) is None


def test_preferred_affinity_is_relaxed(backend):
    lab = {"security": "s2"}                                               # T:1445-1509
    sim = ClusterSim(backend)
    p = mkpod(affinity_preferred=[WeightedPodAffinityTerm(50, PodAffinityTerm(LABEL_HOSTNAME, LabelSelector(lab)))])
    res = sim.provision([p])
    assert sim.scheduled(p) is not None and res.final_stage == [1]


# ---------------- E21/E22: anti-affinity (topology_test.go:1511-1843) ----------------
def test_anti_affinity_hostname_order_independent(backend):
    lab = {"security": "s2"}                                               # T:1511
    for order in (0, 1):
        sim = ClusterSim(backend)
        a = mkpod(labels=lab)
        b = mkpod(anti_required=aff(LABEL_HOSTNAME, lab))
        sim.provision([b, a] if order else [a, b])
        assert sim.scheduled(a).name != sim.scheduled(b).name


def test_anti_affinity_zone_exhausted(backend):
    lab = {"security": "s2"}                                               # T:1531-1570
    sim = ClusterSim(backend)
    zs = [mkpod(labels=lab, requests={"cpu": "2"}, node_selector={LABEL_ZONE: z}) for z in ("test-zone-1", "test-zone-2", "test-zone-3")]
    anti = mkpod(anti_required=aff(LABEL_ZONE, lab))
    sim.provision(zs + [anti])
    assert all(sim.scheduled(p) is not None for p in zs) and sim.scheduled(anti) is None


def test_anti_affinity_schroedinger(backend):
    lab = {"security": "s2"}                                               # T:1713-1743
    sim = ClusterSim(backend)
    anywhere = mkpod(anti_required=aff(LABEL_ZONE, lab), requests={"cpu": "2"})
    target = mkpod(labels=lab)
    sim.provision([anywhere, target])
    assert sim.scheduled(anywhere) is not None and sim.scheduled(target) is None
    sim.provision([target])
    assert sim.scheduled(target).labels[LABEL_ZONE] != sim.scheduled(anywhere).labels[LABEL_ZONE]


def test_inverse_anti_affinity_with_existing_nodes(backend):
    lab = {"security": "s2"}                                               # T:1745-1843
    sim = ClusterSim(backend)
    blockers = [mkpod(anti_required=aff(LABEL_ZONE, lab), requests={"cpu": "2"}, node_selector={LABEL_ZONE: z})
                for z in ("test-zone-1", "test-zone-2", "test-zone-3")]
    sim.provision(blockers)
    p = mkpod(labels=lab)
    sim.provision([p])
    assert sim.scheduled(p) is None


# ---------------- E23: namespaces (topology_test.go:2054-2172) ----------------
def test_affinity_namespaces(backend):
    lab = {"security": "s2"}
    sim = ClusterSim(backend)
    target = mkpod(labels=lab, namespace="other")
    f1 = mkpod(affinity_required=aff(LABEL_HOSTNAME, lab))                 # T:2055: target in another namespace is invisible
    f2 = mkpod(affinity_required=aff(LABEL_HOSTNAME, lab, ["other"]))      # T:2075: namespace list makes it visible
    sim.provision([f1, f2, target])
    assert sim.scheduled(f1) is None
    assert sim.scheduled(f2).name == sim.scheduled(target).name


# ---------------- E24/E25: provisioner weights & taints ----------------
def test_provisioner_weight_order(backend):
    # provisioning/suite_test.go:1129-1204: highest weight first; explicit selection wins
    its = fake.default_instance_types()
    provs = [fake.provisioner("low", len(its), weight=1), fake.provisioner("high", len(its), weight=20),
             fake.provisioner("mid", len(its), weight=10)]
    sim = ClusterSim(backend, instance_types=its, provisioners=provs)
    a = mkpod()
    b = mkpod(node_selector={"karpenter.sh/provisioner-name": "low"})
    res = sim.provision([a, b])
    assert sorted(n.provisioner for n in res.new_nodes) == ["high", "low"]


def test_taints(backend):
    prov = default_prov(taints=[Taint("test-key", "test-value", "NoSchedule")])     # T:2210
    sim = ClusterSim(backend, provisioners=[prov])
    pods = [mkpod(),
            mkpod(tolerations=[Toleration("test-key", "Equal", "test-value", "NoSchedule")]),
            mkpod(tolerations=[Toleration("test-key", "Exists", "", "NoSchedule")]),
            mkpod(tolerations=[Toleration("test-key", "Equal", "other", "NoSchedule")]),
            mkpod(tolerations=[Toleration(operator="Exists")])]                      # T:2243 tolerate everything
    sim.provision(pods)
    assert [sim.scheduled(p) is not None for p in pods] == [False, True, True, False, True]


# ---------------- per-pod failure reasons (scheduler.go:135-172 recordSchedulingResults, one error per provisioner: :193-217) ----------------
def test_failure_reasons(backend):
    """What the reference reports through `errors[pod]` -- for each provisioner, the step of scheduler.add / Node.Add that refused the pod --
    comes back as KS_WHY_* codes (ks_result.pod_reason; 4 bits per provisioner in weight order)."""
    from karpenter_core_amd.model import (REASON_LIMITS, REASON_NO_INSTANCE_TYPE, REASON_REQUIREMENTS, REASON_TAINTS, REASON_TOPOLOGY, reason_codes)

    def why(sim, pod, nprov=1):
        res = sim.last
        idx = [i for i, q in enumerate(sim.last_pods) if q.uid == pod.uid][0]
        assert idx in res.unscheduled
        return reason_codes(res.reasons[idx], nprov)

    # "all available instance types exceed provisioner limits" (scheduler.go:198-201; provisioning/suite_test.go:238-356)
    sim = ClusterSim(backend, provisioners=[default_prov(limits={"cpu": "0"})])
    p = mkpod()
    sim.last_pods = [p]; sim.provision([p])
    assert why(sim, p) == [REASON_LIMITS]
    # taints (node.go:64; topology_test.go:2209-2257) on the first provisioner, a clean second one that is too small (node.go:94-98)
    tainted = fake.provisioner("tainted", 0, weight=10, taints=[Taint("dedicated", "x", "NoSchedule")], discovery_label=True)
    sim = ClusterSim(backend, provisioners=[tainted, default_prov()])
    p = mkpod(requests={"cpu": "10000"})
    sim.last_pods = [p]; sim.provision([p])
    assert why(sim, p, 2) == [REASON_TAINTS, REASON_NO_INSTANCE_TYPE]
    # incompatible requirements (node.go:77; suite_test.go:113-553): the provisioner is pinned to another zone; a zone nobody offers gets
    # past Compatible (the key is well known and the provisioner does not constrain it) and fails at the instance types (node.go:94-98)
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1"])])])
    p = mkpod(node_selector={LABEL_ZONE: "test-zone-2"})
    sim.last_pods = [p]; sim.provision([p])
    assert why(sim, p) == [REASON_REQUIREMENTS]
    sim = ClusterSim(backend)
    p = mkpod(node_selector={LABEL_ZONE: "unknown-zone"})
    sim.last_pods = [p]; sim.provision([p])
    assert why(sim, p) == [REASON_NO_INSTANCE_TYPE]
    # unsatisfiable topology constraint (node.go:83; topology_test.go:35-50)
    sim = ClusterSim(backend)
    p = mkpod(labels=LABELS, spread=spread("unknown"))
    sim.last_pods = [p]; sim.provision([p])
    assert why(sim, p) == [REASON_TOPOLOGY]


# ---------------- E11: in-flight taints (suite_test.go:1533-1658; state/node.go:61-78) ----------------
def _reuse_after(backend, mutate):
    """Provision one pod, delete it (the node stays, empty), let `mutate(node)` change the node, provision a second pod: same node?"""
    sim = ClusterSim(backend)
    first = mkpod(limits={"cpu": "8"})
    sim.provision([first])
    node1 = sim.scheduled(first)
    sim.delete_pod(first)
    mutate(node1)
    second = mkpod()
    sim.provision([second])
    return node1.name, sim.scheduled(second).name


def test_inflight_taints(backend):
    from karpenter_core_amd.model import state_node_taints, TAINT_NODE_NOT_READY, TAINT_NODE_UNREACHABLE
    foreign = Taint("foo.com/taint", "tainted", "NoSchedule")
    not_ready = [Taint(TAINT_NODE_NOT_READY, "", "NoSchedule"), Taint(TAINT_NODE_UNREACHABLE, "", "NoSchedule")]

    def set_taints(raw, startup=(), initialized=False):
        def f(node):
            node.taints = state_node_taints(raw, startup, initialized)
            if not initialized:
                node.labels.pop("karpenter.sh/initialized", None)
        return f

    a, b = _reuse_after(backend, set_taints([]))                                     # S:1534-1553: no taints -> the empty node is reused
    assert a == b
    a, b = _reuse_after(backend, set_taints([foreign], initialized=True))            # S:1554-1579: a foreign taint -> a new node
    assert a != b
    a, b = _reuse_after(backend, set_taints([foreign] + not_ready, startup=[foreign]))   # S:1580-1608: startup + not-ready taints are ignored until initialized
    assert a == b
    a, b = _reuse_after(backend, set_taints([foreign], startup=[foreign], initialized=True))   # S:1609-1633: the startup taint re-appears after initialization -> it counts
    assert a != b
    a, b = _reuse_after(backend, set_taints(not_ready, initialized=True))            # S:1634-1657: NotReady / unreachable never count
    assert a == b


# ---------------- E24: topology counted across provisioners (topology_test.go:2174-2207; T:2174) ----------------
def test_zonal_spread_across_provisioners(backend):
    its = fake.default_instance_types()
    provs = [fake.provisioner("a", len(its), requirements=[Expr(LABEL_ZONE, "In", ["test-zone-1"])], discovery_label=True),
             fake.provisioner("b", len(its), requirements=[Expr(LABEL_ZONE, "In", ["test-zone-2", "test-zone-3"])], discovery_label=True)]
    sim = ClusterSim(backend, instance_types=its, provisioners=provs)
    topo = spread(LABEL_ZONE, labels={"foo": "bar"})
    pods = [mkpod(labels={"foo": "bar"}, spread=topo) for _ in range(10)]
    sim.provision(pods)
    assert sorted(sim.skew(LABEL_ZONE, topo[0].label_selector)) == [3, 3, 4]


# ---------------- pins of host-side logic that the oracle and the product both restate (derived from the reference lines, not from either) ----------------
def test_spread_group_identity_ignores_node_selector_values(backend):
    """TopologyGroup.Hash (topologygroup.go:137-153) hashes with hashstructure v2, which walks EXPORTED fields only: of the spread group's
    TopologyNodeFilter ([]map[string]*Requirement, topologynodefilter.go:28) that is the label KEYS and Requirement.Key -- not the values.  Two
    pods whose hostname spread differs only in the VALUE of their zone node selector therefore share ONE group, and the group keeps the filter
    of the pod that created it (Topology.Update reuses the existing group, topology.go:99-108).  Consequence: the nodes of the second pod's
    zone do not match the group's filter (TopologyGroup.Counts, topologygroup.go:109-111), are never counted, and its replicas may pile onto one
    node although maxSkew is 1 -- with value-sensitive identities each selector would get its own group and the pods would be spread."""
    sim = ClusterSim(backend)
    topo = spread(LABEL_HOSTNAME)
    a = [mkpod(labels=LABELS, spread=topo, node_selector={LABEL_ZONE: "test-zone-1"}, requests={"cpu": "0.1"}) for _ in range(2)]
    b = [mkpod(labels=LABELS, spread=topo, node_selector={LABEL_ZONE: "test-zone-2"}, requests={"cpu": "0.1"}) for _ in range(2)]
    sim.provision(a + b)
    assert len({sim.scheduled(p).name for p in a}) == 2            # counted by the group's (their own) filter: spread over two nodes
    assert len({sim.scheduled(p).name for p in b}) == 1            # not counted by the group they share: both on one node


def test_relaxation_order(backend):
    """Preferences.Relax (preferences.go:36-56) tries, in this order and one per failed attempt: extra required node-affinity terms, preferred pod
    affinity, preferred pod ANTI-affinity, preferred NODE affinity, ScheduleAnyway spreads, the PreferNoSchedule toleration.  The number of
    relaxations a pod needs (`final_stage`) exposes the order: a harmless preference that comes EARLIER in the list is dropped before the one that
    actually blocks."""
    from karpenter_core_amd.model import PodAffinityTerm
    sim = ClusterSim(backend)
    harmless_anti = [WeightedPodAffinityTerm(1, PodAffinityTerm(LABEL_HOSTNAME, LabelSelector({"nobody": "has-this"})))]
    blocking_pref = [PreferredTerm(1, [Expr(LABEL_ZONE, "In", ["no-such-zone"])])]
    p = mkpod(anti_preferred=harmless_anti, preferred_affinity=blocking_pref)
    res = sim.provision([p])
    assert sim.scheduled(p) is not None and res.final_stage[0] == 2          # anti-affinity preference first (:45-47), node-affinity preference second (:48-50)
    sim = ClusterSim(backend)
    harmless_pref = [PreferredTerm(1, [Expr(LABEL_ZONE, "In", ["test-zone-1"])])]
    blocking_spread = spread("no-such-topology-key", when=SCHEDULE_ANYWAY)
    q = mkpod(labels=LABELS, preferred_affinity=harmless_pref, spread=blocking_spread)
    res = sim.provision([q])
    assert sim.scheduled(q) is not None and res.final_stage[0] == 2          # node-affinity preference (:48-50) before the ScheduleAnyway spread (:51-53)
    zone = res.new_nodes[0].requirements.get(LABEL_ZONE)
    assert zone is None or len(zone.values) != 1 or zone.complement         # ... so the zone preference is gone from the node


def test_host_port_ip_forms(backend):
    """entry.matches (hostportusage.go:45-57): same protocol and port conflict iff the IPs are equal or EITHER is unspecified -- 0.0.0.0, the empty
    host IP (defaulted to 0.0.0.0, :133-136) and the IPv6 unspecified address in any spelling (net.IP.IsUnspecified)."""
    def hp(ip):
        return [HostPort(port=8080, protocol="TCP", host_ip=ip)]
    for ip1, ip2, together in (("10.0.0.1", "10.0.0.2", True), ("10.0.0.1", "10.0.0.1", False), ("10.0.0.1", "::", False),
                               ("10.0.0.1", "0:0:0:0:0:0:0:0", False), ("", "10.0.0.9", False), ("fe80::1", "FE80::1", False)):
        sim = ClusterSim(backend)
        a, b = mkpod(ports=hp(ip1)), mkpod(ports=hp(ip2))
        sim.provision([a, b])
        assert (sim.scheduled(a).name == sim.scheduled(b).name) == together, (ip1, ip2)


# ---------------- VolumeUsage (suite_test.go:1994-2270; existingnode.go:87-94, volumeusage.go:102-195) ----------------
CSI = "fake.csi.provider"


def _volume_cluster(backend, limit=10):
    """S:1997-2029: one huge instance type, an in-flight node from a first pod, then a CSINode giving it `limit` volumes of CSI."""
    sim = ClusterSim(backend, instance_types=[fake.new_instance_type("instance-type", {"cpu": "1024", "pods": "1024"})],
                     provisioners=[default_prov(limits=None)])
    first = mkpod()
    sim.provision([first])
    node = sim.scheduled(first)
    node.volume_limits = {CSI: limit}
    return sim, node


def _claims(pod_name, claims, pvcs, pvs=None):
    return resolve_pod_volumes("default", pod_name, [{"name": f"v{i}", "pvc": c} for i, c in enumerate(claims)], pvcs,
                               {"my-storage-class": CSI}, pvs or {})


def test_volume_limits_force_a_second_node(backend):
    # S:1995-2057: 6 pods x 2 dynamic claims each against a limit of 10 -> the in-flight node takes 5, one new node is launched
    sim, node = _volume_cluster(backend)
    pvcs = {f"default/my-claim-{ab}-{i}": {"storage_class": "my-storage-class"} for ab in "ab" for i in range(6)}
    pods = [mkpod(volumes=_claims(f"p{i}", [f"my-claim-a-{i}", f"my-claim-b-{i}"], pvcs)) for i in range(6)]
    res = sim.provision(pods)
    assert len(sim.nodes) == 2
    assert len(res.existing[node.name]) == 5 and len(res.new_nodes) == 1 and len(res.new_nodes[0].pods) == 1
    assert len(node.volumes) == 10


def test_volume_limits_same_claim_counts_once(backend):
    # S:2058-2123: 100 pods mounting one (bound) claim twice all fit the in-flight node: a claim id counts once per node
    sim, node = _volume_cluster(backend)
    pvcs = {"default/my-claim": {"storage_class": "my-storage-class", "volume_name": "my-volume"}}
    pods = [mkpod(volumes=_claims(f"p{i}", ["my-claim", "my-claim"], pvcs, pvs={"my-volume": CSI})) for i in range(100)]
    assert all(p.volumes == [Volume(CSI, "default/my-claim")] for p in pods)
    sim.provision(pods)
    assert len(sim.nodes) == 1 and len(node.volumes) == 1


def test_volume_limits_static_and_non_csi_volumes(backend):
    # S:2124-2189 non-dynamic PVC (storage class "", driver from the PV); S:2190-2240 an NFS volume has no CSI driver and is not tracked
    for pv_driver in (CSI, None):
        sim, node = _volume_cluster(backend)
        pvcs = {"default/my-claim": {"storage_class": "", "volume_name": "my-volume"}}
        pods = [mkpod(volumes=_claims(f"p{i}", ["my-claim", "my-claim"], pvcs, pvs={"my-volume": pv_driver})) for i in range(5)]
        assert all(len(p.volumes) == (1 if pv_driver else 0) for p in pods)
        sim.provision(pods)
        assert len(sim.nodes) == 1


def test_volume_lookup_errors(backend):
    # volumeusage.go:152-154,175-184: a failed Get is returned by Validate -> ExistingNode.Add fails on every existing node, new nodes do not look
    with pytest.raises(VolumeLookupError):
        resolve_pod_volumes("default", "p", [{"name": "v", "pvc": "missing"}], {}, {}, {})
    with pytest.raises(VolumeLookupError):       # S:2241-2270's pod: an ephemeral volume of a storage class that does not exist
        resolve_pod_volumes("default", "p", [{"name": "tmp-ephemeral", "ephemeral": {"storage_class": "non-existent"}}], {}, {}, {})
    sim, node = _volume_cluster(backend)
    bad, good = mkpod(volume_error=True), mkpod()
    res = sim.provision([bad, good])
    assert sim.scheduled(good).name == node.name
    assert sim.scheduled(bad) is not None and sim.scheduled(bad).name != node.name and len(res.new_nodes) == 1


def test_volume_limits_mixed_claims(backend):
    # claims shared between pods, claims already on the node, generic ephemeral volumes (unique per pod), two drivers, an unlimited
    # driver, and a node that is already over its limit (refuses every pod, even one without volumes: Exceeds ranges over the union)
    other = "other.csi.provider"
    sim, node = _volume_cluster(backend, limit=4)
    node.volume_limits[other] = 1
    warm = mkpod(volumes=[Volume(CSI, "default/shared-0")])
    sim.provision([warm])
    assert sim.scheduled(warm).name == node.name and node.volumes == [Volume(CSI, "default/shared-0")]
    eph = lambda name: resolve_pod_volumes("default", name, [{"name": "scratch", "ephemeral": {"storage_class": "my-storage-class"}}], {}, {"my-storage-class": CSI}, {})
    pods = [mkpod(volumes=[Volume(CSI, "default/shared-0"), Volume(CSI, "default/shared-1")]),      # +1 (shared-0 is mounted already)
            mkpod(volumes=[Volume(CSI, "default/shared-1"), Volume("unlimited.csi", "default/x")]),   # +0
            mkpod(volumes=eph("e1")),                                                                 # +1
            mkpod(volumes=eph("e2") + [Volume(other, "default/o1")]),                                 # +1, other 1/1
            mkpod(volumes=[Volume(other, "default/o2")]),                                             # other would be 2/1 -> refused
            mkpod(volumes=eph("e3")),                                                                 # CSI would be 5/4 -> refused
            mkpod(volumes=[Volume(CSI, "default/shared-1")])]                                         # +0 -> fits
    for i, p in enumerate(pods):
        p.containers[0].requests = {"cpu": f"{100 - i}m"}       # queue order == list order
    res = sim.provision(pods)
    assert sorted(res.existing[node.name]) == [0, 1, 2, 3, 6]
    assert sum(len(n.pods) for n in res.new_nodes) == 2
    # over the limit: the CSINode's count drops below what is mounted
    node.volume_limits[CSI] = 2
    p = mkpod()
    sim.provision([p])
    assert sim.scheduled(p).name != node.name


# ---------------- label normalisation (pkg/scheduling/requirement_test.go:44-80, apis/v1alpha5/labels.go:84-110) ----------------
def test_beta_labels_are_normalised(backend):
    """NewLabelRequirements / NewNodeSelectorRequirements / NewPodRequirements map the deprecated label keys to their stable names
    (the reference asserts r.Keys() == {arch, os, instance-type, region, zone} for all three constructors).  Restated through Solve:
    a pod that names the BETA keys in its node selector, in a required node-affinity term and in a preferred one lands on a node whose
    requirements carry the STABLE keys with those values -- and none of the beta keys."""
    beta = {"failure-domain.beta.kubernetes.io/zone": "test-zone-2", "failure-domain.beta.kubernetes.io/region": "test",
            "beta.kubernetes.io/arch": "arm64", "beta.kubernetes.io/os": "linux", "beta.kubernetes.io/instance-type": "arm-instance-type"}
    stable = {LABEL_ZONE: "test-zone-2", "topology.kubernetes.io/region": "test", LABEL_ARCH: "arm64", LABEL_OS: "linux",
              LABEL_INSTANCE_TYPE: "arm-instance-type"}
    exprs = [Expr(k, "In", [v]) for k, v in beta.items()]
    pods = [mkpod(node_selector=dict(beta)),                                                   # NewLabelRequirements
            mkpod(required_affinity=[list(exprs)]),                                            # NewNodeSelectorRequirements (required term)
            mkpod(node_selector=dict(beta), required_affinity=[list(exprs)], preferred_affinity=[PreferredTerm(1, list(exprs))])]
    for p in pods:
        sim = ClusterSim(backend)
        sim.provision([p])
        node = sim.scheduled(p)
        assert node is not None
        nn = [n for n in sim.last.new_nodes if n.pods][0]
        for k, v in stable.items():
            r = nn.requirements[k]
            assert not r.complement and list(r.values) == [v], (k, r)
        assert not any(k in nn.requirements for k in beta)
        assert nn.instance_types == ["arm-instance-type"]
        assert node.labels[LABEL_ZONE] == "test-zone-2" and node.labels[LABEL_ARCH] == "arm64"
