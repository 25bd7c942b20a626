"""More of the reference's envtest scenarios restated as pure Solve() fixtures (round 3): the It() blocks of
pkg/controllers/provisioning/scheduling/topology_test.go and suite_test.go that tests/test_scenarios.py does not cite.  Same rules: every test names
the reference lines it restates and asserts what the reference asserts; every scenario runs against the CPU oracle here and against the HIP path on the
GPU box (-m gpu)."""
import pytest

from helpers import BACKENDS, ClusterSim, mkpod, mkpods
from karpenter_core_amd import fake
from karpenter_core_amd.model import (ClusterPod, Expr, LabelSelector, PodAffinityTerm, PreferredTerm, StateNode, Taint, Toleration,
                                      TopologySpreadConstraint, WeightedPodAffinityTerm, DO_NOT_SCHEDULE, LABEL_ARCH, LABEL_CAPACITY_TYPE,
                                      LABEL_HOSTNAME, LABEL_INSTANCE_TYPE, LABEL_OS, LABEL_ZONE, SCHEDULE_ANYWAY)
from test_scenarios import LABELS, aff, default_prov, spread

pytestmark = pytest.mark.parametrize("backend", BACKENDS)
RR = {"cpu": "1.1"}      # "enough resource requests that the first node we create fills a node and can't act as an in-flight node" (T:126)


def zones_prov(*zones):
    return default_prov(requirements=[Expr(LABEL_ZONE, "In", list(zones))])


def bare_node(sim, name, **labels):
    """test.Node(...) applied straight to the API server: a node no provisioner owns -- never a scheduling target, but its pods count (topology.go:231-276)."""
    n = StateNode(name=name, labels=dict(labels))
    sim.nodes.append(n)
    return n


def running(sim, node, labels=None, namespace="default", uid=None):
    sim._n += 1
    sim.cluster_pods.append(ClusterPod(uid=uid or f"running-{sim._n}", namespace=namespace, node_name=node.name, labels=dict(labels or {})))


# ---------------- topology_test.go: zonal ----------------
def test_spread_with_a_selector_nothing_matches(backend):
    topo = [TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, LabelSelector({"app.kubernetes.io/name": "{{zqfmgb}}"}))]      # T:52 (an invalid label VALUE: the selector selects nothing, topologygroup.go:246-252;
                                                                                                                         #       KSP1 tokens carry no blanks, the braces stay)
    sim = ClusterSim(backend)
    pods = mkpods(2, labels=LABELS, spread=topo)
    sim.provision(pods)
    assert all(sim.scheduled(p) is not None for p in pods)
    assert sim.skew(LABEL_ZONE, LabelSelector(dict(LABELS))) == [2]      # (the reference counts by the pods' own labels here: both share one node)


def test_zonal_spread_match_expressions(backend):
    sel = LabelSelector({}, [Expr("test", "In", ["test"])])                                                                        # T:82
    topo = [TopologySpreadConstraint(1, LABEL_ZONE, DO_NOT_SCHEDULE, sel)]
    sim = ClusterSim(backend)
    sim.provision(mkpods(4, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, sel) == [1, 1, 2]


def test_zonal_spread_provisioner_zones_with_an_existing_pod(backend):
    sim = ClusterSim(backend)                                                                                                      # T:124
    first = mkpod(labels=LABELS, requests=RR, node_selector={LABEL_ZONE: "test-zone-3"})
    sim.provision([first])
    assert sim.scheduled(first) is not None
    sim.provisioners[0].requirements = [Expr(LABEL_ZONE, "In", ["test-zone-1", "test-zone-2"])]
    topo = spread(LABEL_ZONE)
    sim.provision(mkpods(6, labels=LABELS, requests=RR, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 2, 2]      # zone-3's one pod caps the two reachable zones at two each


def test_zonal_spread_non_minimum_domain(backend):
    topo = spread(LABEL_ZONE, 5)                                                                                                   # T:163
    sim = ClusterSim(backend, provisioners=[zones_prov("test-zone-1")])
    sim.provision([mkpod(labels=LABELS, requests=RR, spread=topo)])
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1]
    sim.provisioners[0].requirements = [Expr(LABEL_ZONE, "In", ["test-zone-2"])]
    sim.provision([mkpod(labels=LABELS, requests=RR, spread=topo)])
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1]
    sim.provisioners[0].requirements = [Expr(LABEL_ZONE, "In", ["test-zone-3"])]
    sim.provision(mkpods(10, labels=LABELS, requests=RR, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1, 6]


def test_zonal_spread_discovers_domains_of_bound_pods(backend):
    topo = spread(LABEL_ZONE)                                                                                                      # T:276
    sim = ClusterSim(backend, provisioners=[zones_prov("test-zone-1")])
    sim.provision([mkpod(labels=LABELS, requests=RR)])                   # (no constraint of its own: it only counts)
    sim.provisioners[0].requirements = [Expr(LABEL_ZONE, "In", ["test-zone-2", "test-zone-3"])]
    sim.provision(mkpods(10, labels=LABELS, requests=RR, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 2, 2]


def test_zonal_spread_counts_only_matching_scheduled_pods(backend):
    topo = spread(LABEL_ZONE)                                                                                                      # T:308
    sim = ClusterSim(backend)
    first, second, third = bare_node(sim, "first", **{LABEL_ZONE: "test-zone-1"}), bare_node(sim, "second", **{LABEL_ZONE: "test-zone-2"}), bare_node(sim, "third")
    running(sim, first)                                   # ignored, missing labels
    running(sim, third, LABELS)                           # ignored, no domain on node
    running(sim, first, LABELS, namespace="wrong")        # ignored, wrong namespace
    running(sim, first, LABELS); running(sim, first, LABELS); running(sim, second, LABELS)
    # (pending / terminating / Failed / Succeeded pods are not listed as cluster pods at all: topology.go:404-406 IgnoredForTopology)
    sim.provision(mkpods(2, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 2, 2]


# ---------------- topology_test.go: hostname ----------------
def app_pod(app, arch=None):
    kw = dict(labels={"app": app}, spread=[TopologySpreadConstraint(1, LABEL_HOSTNAME, DO_NOT_SCHEDULE, LabelSelector({"app": app}))])
    if arch:
        kw["required_affinity"] = [[Expr(LABEL_ARCH, "In", [arch])]]
    return mkpod(**kw)


def test_hostname_spread_of_two_deployments_shares_nodes(backend):
    sim = ClusterSim(backend)                                                                                                      # T:412 (issue #1425)
    pods = [app_pod("app1"), app_pod("app1"), app_pod("app2"), app_pod("app2")]
    sim.provision(pods)
    assert all(sim.scheduled(p) is not None for p in pods)
    assert len(sim.nodes) == 2      # "ensures that we launch the minimum number of nodes"
    sim = ClusterSim(backend)                                                                                                      # T:447: different architectures cannot share
    pods = [app_pod("app1", "amd64"), app_pod("app1", "amd64"), app_pod("app2", "arm64"), app_pod("app2", "arm64")]
    sim.provision(pods)
    assert all(sim.scheduled(p) is not None for p in pods)
    assert len(sim.nodes) == 4


# ---------------- topology_test.go: capacity type / arch ----------------
def test_capacity_type_spread_with_provisioner_constraints(backend):
    topo = spread(LABEL_CAPACITY_TYPE)                                                                                             # T:508
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr(LABEL_CAPACITY_TYPE, "In", ["spot", "on-demand"])])])
    sim.provision(mkpods(4, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [2, 2]


def test_capacity_type_spread_do_not_schedule_cap(backend):
    topo = spread(LABEL_CAPACITY_TYPE)                                                                                             # T:526
    sim = ClusterSim(backend, provisioners=[default_prov(requirements=[Expr(LABEL_CAPACITY_TYPE, "In", ["spot"])])])
    sim.provision([mkpod(labels=LABELS, requests=RR, spread=topo)])
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [1]
    sim.provisioners[0].requirements = [Expr(LABEL_CAPACITY_TYPE, "In", ["on-demand"])]
    sim.provision(mkpods(5, labels=LABELS, requests=RR, spread=topo))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [1, 2]


def test_capacity_type_spread_counts_only_matching_scheduled_pods(backend):
    topo = spread(LABEL_CAPACITY_TYPE)                                                                                             # T:592
    sim = ClusterSim(backend)
    first, second, third = bare_node(sim, "first", **{LABEL_CAPACITY_TYPE: "spot"}), bare_node(sim, "second", **{LABEL_CAPACITY_TYPE: "on-demand"}), bare_node(sim, "third")
    running(sim, first); running(sim, third, LABELS); running(sim, first, LABELS, namespace="wrong")
    running(sim, first, LABELS); running(sim, first, LABELS); running(sim, second, LABELS)
    sim.provision(mkpods(2, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [2, 3]


def test_skew_without_selector_counts_every_pod(backend):
    sim = ClusterSim(backend)                                                                                                      # T:625
    sim.provision([mkpod()])
    assert sim.skew(LABEL_CAPACITY_TYPE, LabelSelector({})) == [1]


def test_spread_whose_selector_matches_no_owner(backend):
    topo = spread(LABEL_HOSTNAME)                                                                                                  # T:637 "interdependent selectors": the owners do not
    sim = ClusterSim(backend)                                                                                                      #        count themselves, so one node takes them all
    pods = mkpods(5, spread=topo)
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 1


def test_capacity_type_spread_node_affinity_limits_what_counts(backend):
    sim = ClusterSim(backend)                                                                                                      # T:661
    first = mkpod(labels=LABELS, required_affinity=[[Expr(LABEL_ZONE, "In", ["test-zone-1"]), Expr(LABEL_CAPACITY_TYPE, "In", ["on-demand"])]])
    sim.provision([first])
    assert sim.scheduled(first) is not None
    topo = spread(LABEL_CAPACITY_TYPE)
    sim.provision(mkpods(5, labels=LABELS, spread=topo, required_affinity=[[Expr(LABEL_ZONE, "In", ["test-zone-2"]), Expr(LABEL_CAPACITY_TYPE, "In", ["spot"])]]))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [1, 5]      # the zone-2 filter keeps the on-demand pod of zone-1 out of the count


def test_capacity_type_spread_sees_the_existing_node(backend):
    sim = ClusterSim(backend)                                                                                                      # T:697
    first = mkpod(labels=LABELS, node_selector={LABEL_INSTANCE_TYPE: "single-pod-instance-type"}, required_affinity=[[Expr(LABEL_CAPACITY_TYPE, "In", ["on-demand"])]])
    sim.provision([first])
    assert sim.scheduled(first) is not None
    topo = spread(LABEL_CAPACITY_TYPE)
    sim.provisioners[0].requirements = [Expr(LABEL_CAPACITY_TYPE, "In", ["spot"])]
    sim.provision(mkpods(5, labels=LABELS, requests={"cpu": "2"}, spread=topo))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [1, 2]


def test_arch_spread_sees_the_existing_node(backend):
    sim = ClusterSim(backend)                                                                                                      # T:740
    first = mkpod(labels=LABELS, node_selector={LABEL_INSTANCE_TYPE: "single-pod-instance-type"}, required_affinity=[[Expr(LABEL_ARCH, "In", ["amd64"])]])
    sim.provision([first])
    assert sim.scheduled(first) is not None
    topo = spread(LABEL_ARCH)
    sim.provisioners[0].requirements = [Expr(LABEL_ARCH, "In", ["arm64"])]
    sim.provision(mkpods(5, labels=LABELS, requests={"cpu": "2"}, spread=topo))
    assert sim.skew(LABEL_ARCH, topo[0].label_selector) == [1, 2]


def test_spread_over_a_key_only_provisioners_define(backend):
    its = fake.default_instance_types()                                                                                            # T:825: a 4:1 spot : on-demand split through a custom key
    key = "capacity.spread.4-1"
    provs = [fake.provisioner("spot", len(its), requirements=[Expr(LABEL_CAPACITY_TYPE, "In", ["spot"]), Expr(key, "In", ["2", "3", "4", "5"])], discovery_label=True),
             fake.provisioner("on-demand", len(its), requirements=[Expr(LABEL_CAPACITY_TYPE, "In", ["on-demand"]), Expr(key, "In", ["1"])], discovery_label=True)]
    sim = ClusterSim(backend, instance_types=its, provisioners=provs)
    topo = spread(key)
    pods = mkpods(20, labels=LABELS, spread=topo)
    sim.provision(pods)
    assert all(sim.scheduled(p) is not None for p in pods)
    assert sim.skew(key, topo[0].label_selector) == [4, 4, 4, 4, 4]
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [4, 16]


# ---------------- topology_test.go: combined constraints ----------------
def test_zone_spread_with_schedule_anyway_hostname_spread(backend):
    topo = spread(LABEL_ZONE) + spread(LABEL_HOSTNAME, when=SCHEDULE_ANYWAY)                                                       # T:882
    sim = ClusterSim(backend, provisioners=[zones_prov("test-zone-1", "test-zone-2")])
    sim.provision(mkpods(10, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 1]          # one pod per reachable zone: zone-3 stays at zero
    assert sim.skew(LABEL_HOSTNAME, topo[1].label_selector) == [1, 1]


def test_capacity_type_and_hostname_spread(backend):
    topo = spread(LABEL_CAPACITY_TYPE) + spread(LABEL_HOSTNAME, 3)                                                                 # T:910
    sim = ClusterSim(backend)
    for n, want in ((2, [1, 1]), (3, [2, 3]), (5, [5, 5]), (11, [10, 11])):
        sim.provision(mkpods(n, labels=LABELS, spread=topo))
        assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == want
        assert max(sim.skew(LABEL_HOSTNAME, topo[1].label_selector)) <= 3


def test_capacity_type_and_zone_spread(backend):
    topo = spread(LABEL_CAPACITY_TYPE) + spread(LABEL_ZONE)                                                                        # T:953
    sim = ClusterSim(backend)
    for n, ct_max, zone_max in ((2, 1, 1), (3, 3, 2), (5, 5, 4), (11, 11, 7)):
        sim.provision(mkpods(n, labels=LABELS, spread=topo))
        assert max(sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector)) <= ct_max
        assert max(sim.skew(LABEL_ZONE, topo[1].label_selector)) <= zone_max


def test_capacity_type_zone_and_hostname_spread(backend):
    topo = spread(LABEL_CAPACITY_TYPE) + spread(LABEL_ZONE, 2) + spread(LABEL_HOSTNAME, 3)                                         # T:993
    sim = ClusterSim(backend, instance_types=fake.instance_types_assorted())

    def max_skew(key, sel):      # ExpectMaxSkew, expectations.go:362-377
        s = sim.skew(key, sel)
        return max(s) - min(s)
    for i in range(1, 15):
        pods = mkpods(i, labels=LABELS, spread=topo)
        sim.provision(pods)
        assert max_skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) <= 1
        assert max_skew(LABEL_ZONE, topo[1].label_selector) <= 2
        assert max_skew(LABEL_HOSTNAME, topo[2].label_selector) <= 3
        assert all(sim.scheduled(p) is not None for p in pods)


# ---------------- topology_test.go: spread + node affinity ----------------
def zone_in(*zones):
    return [[Expr(LABEL_ZONE, "In", list(zones))]]


def test_zonal_spread_limited_by_node_requirements(backend):
    topo = spread(LABEL_ZONE)                                                                                                      # T:1057
    sim = ClusterSim(backend)
    sim.provision(mkpods(10, labels=LABELS, spread=topo, required_affinity=zone_in("test-zone-1", "test-zone-2")))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [5, 5]


def test_zonal_spread_limited_by_node_affinity_then_opened(backend):
    topo = spread(LABEL_ZONE)                                                                                                      # T:1079
    sim = ClusterSim(backend)
    sim.provision(mkpods(6, labels=LABELS, spread=topo, required_affinity=zone_in("test-zone-1", "test-zone-2")))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [3, 3]
    sim.provisioners[0].requirements = [Expr(LABEL_ZONE, "In", ["test-zone-1", "test-zone-2", "test-zone-3"])]
    sim.provision(mkpods(1, labels=LABELS, spread=topo, required_affinity=zone_in("test-zone-2", "test-zone-3")))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [1, 3, 3]      # the empty zone-3: it improves the skew
    sim.provision(mkpods(5, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_ZONE, topo[0].label_selector) == [4, 4, 4]


def test_capacity_type_spread_limited_by_node_selector(backend):
    topo = spread(LABEL_CAPACITY_TYPE, when=SCHEDULE_ANYWAY)                                                                       # T:1127
    sim = ClusterSim(backend)
    sim.provision(mkpods(5, labels=LABELS, spread=topo, node_selector={LABEL_CAPACITY_TYPE: "spot"}) +
                  mkpods(5, labels=LABELS, spread=topo, node_selector={LABEL_CAPACITY_TYPE: "on-demand"}))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [5, 5]


def test_capacity_type_spread_limited_by_node_affinity_then_opened(backend):
    topo = spread(LABEL_CAPACITY_TYPE)                                                                                             # T:1151
    sim = ClusterSim(backend)
    sim.provision(mkpods(3, labels=LABELS, spread=topo, required_affinity=[[Expr(LABEL_CAPACITY_TYPE, "In", ["spot"])]]))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [3]
    sim.provision(mkpods(1, labels=LABELS, spread=topo, required_affinity=[[Expr(LABEL_CAPACITY_TYPE, "In", ["on-demand", "spot"])]]))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [1, 3]
    sim.provision(mkpods(5, labels=LABELS, spread=topo))
    assert sim.skew(LABEL_CAPACITY_TYPE, topo[0].label_selector) == [4, 5]


# ---------------- topology_test.go: pod affinity / anti-affinity ----------------
def test_empty_affinity_lists(backend):
    sim = ClusterSim(backend)                                                                                                      # T:1196
    p = mkpod(affinity_required=[], anti_required=[])
    sim.provision([p])
    assert sim.scheduled(p) is not None


def test_pod_affinity_on_arch_with_a_hostname_spread(backend):
    lab = {"security": "s2"}                                                                                                       # T:1239
    tsc = spread(LABEL_HOSTNAME, labels=lab)
    a = mkpod(labels=lab, spread=tsc, requests={"cpu": "2"}, node_selector={LABEL_ARCH: "arm64"})
    b = mkpod(labels=lab, spread=tsc, requests={"cpu": "1"}, affinity_required=aff(LABEL_ARCH, lab))
    sim = ClusterSim(backend)
    sim.provision([a, b])
    n1, n2 = sim.scheduled(a), sim.scheduled(b)
    assert n1.labels[LABEL_ARCH] == n2.labels[LABEL_ARCH]      # same arch ...
    assert n1.name != n2.name                                    # ... but, due to the spread, not the same node


def test_self_affinity_hostname_with_constrained_zones(backend):
    lab = {"security": "s2"}                                                                                                       # T:1346
    sim = ClusterSim(backend)
    sim.provision([mkpod(labels=lab, node_selector={LABEL_ZONE: "test-zone-1"}, affinity_required=aff(LABEL_HOSTNAME, lab))])
    pods = mkpods(10, labels=lab, required_affinity=zone_in("test-zone-2", "test-zone-3"), affinity_required=aff(LABEL_HOSTNAME, lab))
    sim.provision(pods)
    assert all(sim.scheduled(p) is None for p in pods)      # node selectors limit what a SPREAD counts, never an affinity: the only non-empty hostname is in zone-1


def test_self_affinity_zone(backend):
    lab = {"security": "s2"}                                                                                                       # T:1390
    sim = ClusterSim(backend)
    pods = mkpods(3, labels=lab, affinity_required=aff(LABEL_ZONE, lab))
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 1
    sim = ClusterSim(backend)                                                                                                      # T:1414: ... further limited to zone-3
    pods = mkpods(3, labels=lab, affinity_required=aff(LABEL_ZONE, lab), required_affinity=zone_in("test-zone-3"))
    sim.provision(pods)
    assert len({sim.scheduled(p).name for p in pods}) == 1
    assert all(sim.scheduled(p).labels[LABEL_ZONE] == "test-zone-3" for p in pods)


def anti(key, labels, ns=None):
    return [PodAffinityTerm(key, LabelSelector(dict(labels)), list(ns or []))]


def test_anti_affinity_zone_when_the_avoided_pod_goes_first(backend):
    lab = {"security": "s2"}                                                                                                       # T:1572
    target, avoider = mkpod(labels=lab, requests={"cpu": "2"}), mkpod(anti_required=anti(LABEL_ZONE, lab))
    sim = ClusterSim(backend)
    sim.provision([target, avoider])
    assert sim.scheduled(target) is not None      # it schedules first (larger) -- nobody knows in which zone yet ...
    assert sim.scheduled(avoider) is None         # ... so no zone is safe for the pod that must avoid it


def test_anti_affinity_on_arch(backend):
    lab = {"security": "s2"}                                                                                                       # T:1594
    tsc = spread(LABEL_HOSTNAME, labels=lab)
    a = mkpod(labels=lab, spread=tsc, requests={"cpu": "2"}, node_selector={LABEL_ARCH: "arm64"})
    b = mkpod(labels=lab, spread=tsc, requests={"cpu": "1"}, anti_required=anti(LABEL_ARCH, lab))
    sim = ClusterSim(backend)
    sim.provision([a, b])
    assert sim.scheduled(a).labels[LABEL_ARCH] != sim.scheduled(b).labels[LABEL_ARCH]


def test_inverse_anti_affinity_on_zone(backend):
    lab = {"security": "s2"}
    for required, lands in ((False, True), (True, False)):                                                                         # T:1637 preferred (inverse) / T:1677 required (inverse)
        kw = (lambda: dict(anti_required=anti(LABEL_ZONE, lab))) if required else (lambda: dict(anti_preferred=[WeightedPodAffinityTerm(10, anti(LABEL_ZONE, lab)[0])]))
        zoned = [mkpod(requests={"cpu": "2"}, node_selector={LABEL_ZONE: z}, **kw()) for z in ("test-zone-1", "test-zone-2", "test-zone-3")]
        target = mkpod(labels=lab)
        sim = ClusterSim(backend)
        sim.provision(zoned + [target])
        assert all(sim.scheduled(p) is not None for p in zoned)      # first fit-descending: the three larger pods go first
        assert (sim.scheduled(target) is not None) == lands          # every zone holds a pod that refuses its company -- a preference gives way, a requirement does not


EVERYTHING = LabelSelector({})      # ExpectSkew with a constraint that carries no selector lists every pod


def test_preferred_affinity_gives_way_to_a_required_spread(backend):
    lab = {"security": "s2"}                                                                                                       # T:1845
    constraint = spread(LABEL_HOSTNAME)
    target = mkpod(labels=lab)
    pods = mkpods(3, labels=LABELS, spread=constraint, affinity_preferred=[WeightedPodAffinityTerm(50, aff(LABEL_HOSTNAME, lab)[0])])
    sim = ClusterSim(backend)
    sim.provision(pods + [target])
    assert all(sim.scheduled(p) is not None for p in pods + [target])
    assert sim.skew(LABEL_HOSTNAME, constraint[0].label_selector) == [1, 1, 1]


def test_zonal_anti_affinity_settles_over_several_batches(backend):
    lab = {"security": "s2"}                                                                                                       # T:1879 ("one of the downsides of late committal")
    sim = ClusterSim(backend)
    for want in ([1], [1, 1], [1, 1, 1], [1, 1, 1]):
        sim.provision(mkpods(3, labels=lab, anti_required=anti(LABEL_ZONE, lab)))      # (the pods left pending are never bound: ExpectDeleteAllUnscheduledPods)
        assert sim.skew(LABEL_ZONE, EVERYTHING) == want


def test_zonal_affinity_to_an_uncommitted_target(backend):
    lab = {"security": "s2"}                                                                                                       # T:1941
    target = mkpod(labels=lab)
    followers = mkpods(10, affinity_required=aff(LABEL_ZONE, lab))
    sim = ClusterSim(backend)
    sim.provision(followers + [target])
    assert all(sim.scheduled(p) is None for p in followers)      # the target's zone is not decided while its node is only planned
    assert sim.skew(LABEL_ZONE, EVERYTHING) == [1]
    sim.provision(followers)
    assert all(sim.scheduled(p) is not None for p in followers)
    assert sim.skew(LABEL_ZONE, EVERYTHING) == [11]


def test_zonal_affinity_to_a_constrained_target(backend):
    lab = {"security": "s2"}                                                                                                       # T:1974
    target = mkpod(labels=lab, required_affinity=zone_in("test-zone-1"))
    followers = mkpods(10, affinity_required=aff(LABEL_ZONE, lab))
    sim = ClusterSim(backend)
    sim.provision(followers + [target])
    assert sim.skew(LABEL_ZONE, EVERYTHING) == [11]


def test_chain_of_dependent_affinities(backend):
    db, web, cache, ui = ({"type": t, "spread": "spread"} for t in ("db", "web", "cache", "ui"))                                   # T:2003
    for order in range(6):      # (the reference repeats 50 times over random UIDs: here the queue's tie-break is steered through the uid instead)
        pods = [mkpod(labels=db), mkpod(labels=web, affinity_required=aff(LABEL_HOSTNAME, db)), mkpod(labels=cache, affinity_required=aff(LABEL_HOSTNAME, web)),
                mkpod(labels=ui, affinity_required=aff(LABEL_HOSTNAME, cache))]
        perm = [(0, 1, 2, 3), (3, 2, 1, 0), (1, 3, 0, 2), (2, 0, 3, 1), (3, 0, 2, 1), (1, 2, 3, 0)][order]
        for rank, p in zip(perm, pods):
            p.uid = f"chain-{order}-{rank}"
        sim = ClusterSim(backend)
        sim.provision(pods)
        assert all(sim.scheduled(p) is not None for p in pods), order
    sim = ClusterSim(backend)                                                                                                      # T:2037: a dependency nobody satisfies ends, unscheduled
    p = mkpod(labels=db, affinity_required=aff(LABEL_HOSTNAME, web))
    sim.provision([p])
    assert sim.scheduled(p) is None


def test_affinity_across_namespaces(backend):
    lab = {"security": "s2"}
    for ns, term_ns in (("other-ns-list", ["other-ns-list"]),                              # T:2092: the term lists the target's namespace
                        ("empty-ns-selector", ["default", "empty-ns-selector"])):         # T:2131: an empty namespaceSelector = every namespace (topology.go:324-352 resolves it to
        topo = spread(LABEL_HOSTNAME)                                                      #         names through the API server: the caller's side of the boundary)
        target = mkpod(labels=lab, namespace=ns)
        follower = mkpod(affinity_required=aff(LABEL_HOSTNAME, lab, term_ns))
        sim = ClusterSim(backend)
        sim.provision(mkpods(10, labels=LABELS, spread=topo) + [target, follower])         # ten nodes, the target on one of them, the follower on the same
        assert sim.scheduled(target) is not None and sim.scheduled(target).name == sim.scheduled(follower).name


# ---------------- topology_test.go: taints ----------------
def test_provisioner_taints_and_tolerations(backend):
    prov = default_prov(taints=[Taint("test-key", "test-value", "NoSchedule")])                                                    # T:2219
    sim = ClusterSim(backend, provisioners=[prov])
    ok = [mkpod(tolerations=[Toleration("test-key", "Exists", "", "NoSchedule")]), mkpod(tolerations=[Toleration("test-key", "Equal", "test-value", "NoSchedule")])]
    sim.provision(ok)
    assert all(sim.scheduled(p) is not None for p in ok)
    bad = [mkpod(), mkpod(tolerations=[Toleration("invalid", "Exists")]), mkpod(tolerations=[Toleration("test-key", "Equal", "", "NoSchedule")])]
    sim.provision(bad)
    assert all(sim.scheduled(p) is None for p in bad)      # missing toleration / key mismatch / value mismatch


def test_tolerations_generate_no_taints(backend):
    sim = ClusterSim(backend)                                                                                                      # T:2249
    p = mkpod(tolerations=[Toleration("test-key", "Exists", "", "NoExecute")])
    res = sim.provision([p])
    assert sim.scheduled(p) is not None and sim.scheduled(p).taints == []      # (the reference's one taint is the not-ready taint of a fresh Machine)
    assert len(res.new_nodes) == 1


# ---------------- suite_test.go: well-known labels, node selectors + requirements + preferences ----------------
ZONES3 = ["test-zone-1", "test-zone-2", "test-zone-3"]


def _alone(backend, pod, **kw):
    sim = ClusterSim(backend, **kw)
    sim.provision([pod])
    return sim.scheduled(pod)


def test_zone_requirements_and_preferences(backend):
    z = LABEL_ZONE
    assert _alone(backend, mkpod(required_affinity=[[Expr(z, "In", ["test-zone-3"])]])).labels[z] == "test-zone-3"                               # S:203
    assert _alone(backend, mkpod(required_affinity=[[Expr(z, "In", ["unknown"])]])) is None                                                       # S:231
    assert _alone(backend, mkpod(required_affinity=[[Expr(z, "NotIn", ["test-zone-1", "test-zone-2", "unknown"])]])).labels[z] == "test-zone-3"   # S:240
    assert _alone(backend, mkpod(required_affinity=[[Expr(z, "NotIn", ZONES3 + ["unknown"])]])) is None                                           # S:250
    wide = [[Expr(z, "In", ZONES3 + ["unknown"])]]
    assert _alone(backend, mkpod(required_affinity=wide, preferred_affinity=[PreferredTerm(1, [Expr(z, "In", ["test-zone-2", "unknown"])])])).labels[z] == "test-zone-2"       # S:260
    assert _alone(backend, mkpod(required_affinity=wide, preferred_affinity=[PreferredTerm(1, [Expr(z, "In", ["unknown"])])])) is not None                                      # S:273 (relaxed away)
    assert _alone(backend, mkpod(required_affinity=wide, preferred_affinity=[PreferredTerm(1, [Expr(z, "NotIn", ["test-zone-1", "test-zone-3"])])])).labels[z] == "test-zone-2"  # S:285
    assert _alone(backend, mkpod(required_affinity=wide, preferred_affinity=[PreferredTerm(1, [Expr(z, "NotIn", ZONES3)])])) is not None                                        # S:298
    n = _alone(backend, mkpod(node_selector={z: "test-zone-3"}, required_affinity=[[Expr(z, "In", ZONES3)]], preferred_affinity=[PreferredTerm(1, [Expr(z, "In", ZONES3)])]))  # S:310
    assert n.labels[z] == "test-zone-3"
    n = _alone(backend, mkpod(node_selector={z: "test-zone-3", LABEL_INSTANCE_TYPE: "arm-instance-type"},                                                                       # S:324
                              required_affinity=[[Expr(z, "In", ["test-zone-1", "test-zone-3"]), Expr(LABEL_INSTANCE_TYPE, "In", ["default-instance-type", "arm-instance-type"])]],
                              preferred_affinity=[PreferredTerm(1, [Expr(z, "NotIn", ["unknown"]), Expr(LABEL_INSTANCE_TYPE, "NotIn", ["unknown"])])]))
    assert n.labels[z] == "test-zone-3" and n.labels[LABEL_INSTANCE_TYPE] == "arm-instance-type"


def test_restricted_labels_never_schedule(backend):
    for key in ("karpenter.sh/emptiness-timestamp", LABEL_HOSTNAME,                      # S:348 v1alpha5.RestrictedLabels
                "kubernetes.io/test", "k8s.io/test", "karpenter.sh/test"):              # S:358 RestrictedLabelDomains
        # (the reference turns these pods away while validating the batch, provisioner.go:185-215; Solve would refuse them as well: nothing defines the label)
        assert _alone(backend, mkpod(required_affinity=[[Expr(key, "In", ["test"])]])) is None, key


def test_labels_of_excepted_domains_come_from_the_provisioner(backend):
    keys = [d + "/test" for d in ("kops.k8s.io", "node.kubernetes.io", "testing.karpenter.sh")]                                    # S:368 LabelDomainExceptions
    prov = default_prov(requirements=[Expr(k, "In", ["test-value"]) for k in keys])
    n = _alone(backend, mkpod(), provisioners=[prov])
    assert all(n.labels[k] == "test-value" for k in keys)


# ---------------- suite_test.go: instance type compatibility ----------------
def arch_prov(*archs):
    return default_prov(requirements=[Expr(LABEL_ARCH, "In", list(archs))])


def test_instance_types_the_provisioner_excludes(backend):
    amd = lambda: [arch_prov("amd64")]                                                                                             # noqa: E731
    assert _alone(backend, mkpod(required_affinity=[[Expr(LABEL_INSTANCE_TYPE, "In", ["arm-instance-type"])]]), provisioners=amd()) is None      # S:708
    assert _alone(backend, mkpod(required_affinity=[[Expr(LABEL_OS, "In", ["ios"])]]), provisioners=amd()) is None                               # S:725 (only the arm type runs ios)
    assert _alone(backend, mkpod(limits={"cpu": "14"}), provisioners=amd()) is None                                                              # S:747 (only the arm type has 14 cpu; limits stand in for requests)


def test_different_selectors_need_different_nodes(backend):
    for a, b in (({LABEL_OS: "linux"}, {LABEL_OS: "windows"}),                                                                     # S:760
                 ({"beta.kubernetes.io/instance-type": "small-instance-type"}, {LABEL_INSTANCE_TYPE: "default-instance-type"}),    # S:780 (the beta label is normalised)
                 ({LABEL_ZONE: "test-zone-1"}, {LABEL_ZONE: "test-zone-2"})):                                                      # S:800
        sim = ClusterSim(backend, provisioners=[arch_prov("arm64", "amd64")])
        pa, pb = mkpod(node_selector=a), mkpod(node_selector=b)
        sim.provision([pa, pb])
        assert sim.scheduled(pa).name != sim.scheduled(pb).name


def test_provider_specific_labels(backend):
    def sim5():
        return ClusterSim(backend, instance_types=fake.instance_types(5))
    sim = sim5()                                                                                                                   # S:865
    large, small = mkpod(node_selector={fake.LABEL_INSTANCE_SIZE: "large"}), mkpod(node_selector={fake.LABEL_INSTANCE_SIZE: "small"})
    sim.provision([large, small])
    assert sim.node_types[sim.scheduled(large).name] == "fake-it-4" and sim.node_types[sim.scheduled(small).name] == "fake-it-0"
    sim = sim5()                                                                                                                   # S:877
    pods = [mkpod(node_selector={fake.LABEL_INSTANCE_SIZE: "large", LABEL_INSTANCE_TYPE: "fake-it-0"}), mkpod(node_selector={fake.LABEL_INSTANCE_SIZE: "small", LABEL_INSTANCE_TYPE: "fake-it-4"})]
    sim.provision(pods)
    assert all(sim.scheduled(p) is None for p in pods)
    sim = sim5()                                                                                                                   # S:893: only some types carry the key
    p = mkpod(required_affinity=[[Expr(fake.LABEL_EXOTIC, "Exists")]])
    sim.provision([p])
    assert fake.LABEL_EXOTIC in sim.scheduled(p).labels and sim.node_types[sim.scheduled(p).name] == "fake-it-4"
    sim = ClusterSim(backend, instance_types=fake.instance_types(5), provisioners=[fake.provisioner("default", 5, discovery_label=True)])        # S:906
    p = mkpod(required_affinity=[[Expr(fake.LABEL_EXOTIC, "DoesNotExist")]])
    sim.provision([p])
    assert sim.scheduled(p) is not None and fake.LABEL_EXOTIC not in sim.scheduled(p).labels


# ---------------- suite_test.go: bin packing, in-flight nodes ----------------
from helpers import format_milli, parse_quantity_milli      # noqa: E402


def test_small_pod_takes_the_smallest_type(backend):
    sim = ClusterSim(backend)                                                                                                      # S:1090
    p = mkpod(requests={"memory": "2000M"})
    sim.provision([p])
    assert sim.node_types[sim.scheduled(p).name] == "small-instance-type"


def test_inflight_node_reuse_by_zone_intersection(backend):
    sim = ClusterSim(backend)                                                                                                      # S:1359
    a = mkpod(limits={"cpu": "10m"}, required_affinity=zone_in("test-zone-2"))
    sim.provision([a])
    b = mkpod(limits={"cpu": "10m"}, required_affinity=zone_in("test-zone-1", "test-zone-2"))
    sim.provision([b])
    assert sim.scheduled(a).name == sim.scheduled(b).name      # zone-2 is in the intersection and the node has room
    c = mkpod(limits={"cpu": "10m"}, required_affinity=zone_in("test-zone-1", "test-zone-3"))
    sim.provision([c])
    assert sim.scheduled(c).name != sim.scheduled(a).name


def test_a_terminating_inflight_node_is_not_reused(backend):
    sim = ClusterSim(backend)                                                                                                      # S:1438
    a = mkpod(limits={"cpu": "10m"})
    sim.provision([a])
    first = sim.scheduled(a)
    first.in_state = False      # deleted: cluster state marks it for deletion and Solve never sees it (scheduler.go:260-264 via provisioner.go:249-254)
    b = mkpod(limits={"cpu": "10m"})
    sim.provision([b])
    assert sim.scheduled(b).name != first.name


def test_hostname_spread_prefers_new_nodes_over_inflight_ones(backend):
    lab = {"foo": "bar"}                                                                                                           # S:1498
    topo = spread(LABEL_HOSTNAME, labels=lab)
    sim = ClusterSim(backend)
    sim.provision(mkpods(4, labels=lab, spread=topo))
    assert sim.skew(LABEL_HOSTNAME, topo[0].label_selector) == [1, 1, 1, 1]
    sim.provision(mkpods(5, labels=lab, spread=topo))
    assert sim.skew(LABEL_HOSTNAME, topo[0].label_selector) == [1] * 9


def _emptied_node_with_bound_daemon(sim, first, ds_cpu, ds_mem):
    """the first pod is deleted again and a daemonset pod of (ds_cpu, ds_mem) is bound by hand: state.Node then reports it under DaemonSetRequests() and
    takes it off Available() (state/node.go:172-190; S:1697-1722 asserts 15.9 -> 14.9 cpu)"""
    node = sim.scheduled(first)
    sim.delete_pod(first)
    avail = dict(node._alloc)
    avail["cpu"] -= parse_quantity_milli(ds_cpu); avail["memory"] -= parse_quantity_milli(ds_mem); avail["pods"] -= 1000
    node.available = {k: format_milli(v) for k, v in avail.items()}
    node.daemonset_requests = {"cpu": ds_cpu, "memory": ds_mem, "pods": "1"}
    return node


def test_bound_daemonset_pods_are_not_reserved_twice(backend):
    ds = mkpod(requests={"cpu": "1", "memory": "1Gi"})                                                                             # S:1660
    sim = ClusterSim(backend, daemonsets=[ds])
    first = mkpod(limits={"cpu": "8"})
    sim.provision([first])
    node = _emptied_node_with_bound_daemon(sim, first, "1", "2Gi")
    assert node.available["cpu"] == "14900m"
    second = mkpod(limits={"cpu": "14.9"})
    sim.provision([second])
    assert sim.scheduled(second).name == node.name      # the daemonset pod has bound: nothing of it remains to be reserved, 14.9 cpu are free


def test_unexpected_daemonset_pods_do_not_free_capacity(backend):
    ds1 = mkpod(requests={"cpu": "1", "memory": "1Gi"}, node_selector={"my-node-label": "value"})                                  # S:1732
    ds2 = mkpod(requests={"cpu": "1m"})
    sim = ClusterSim(backend, daemonsets=[ds1, ds2])
    first = mkpod(limits={"cpu": "8"})
    sim.provision([first])
    node = _emptied_node_with_bound_daemon(sim, first, "1", "2Gi")
    node.labels["my-node-label"] = "value"      # "this label appears on the node for some reason that Karpenter can't track"
    second = mkpod(limits={"cpu": "15.5"})
    sim.provision([second])
    assert sim.scheduled(second) is not None and sim.scheduled(second).name != node.name      # a NEGATIVE remainder of daemonset resources would have made room for it


def test_inflight_nodes_are_packed_before_new_ones_open(backend):
    import numpy as np
    its = [fake.new_instance_type("medium", {"cpu": "4.25", "pods": "4"})]                                                         # S:1824 (rand.Intn(10) batches: seeded here)
    rs = np.random.RandomState(4)
    sim = ClusterSim(backend, instance_types=its)
    for _ in range(10):
        pods = mkpods(int(rs.randint(10)), limits={"cpu": "1"})
        sim.provision(pods)
        assert all(sim.scheduled(p) is not None for p in pods)
    free = sum(parse_quantity_milli(n.available["cpu"]) >= 1000 for n in sim.nodes)
    assert free <= 1      # only the final node may have room for another pod


def _provision_no_binding(sim, pods):
    """ExpectProvisionedNoBinding (expectations.go:234-262): the nodes are launched, the pods stay pending"""
    res = sim.provision(pods, bind=False)
    provs = {p.name: p for p in sim.provisioners}
    for nn in res.new_nodes:
        sim.nodes.append(sim._launch(nn, provs[nn.provisioner]))
    return res


def test_no_pre_binding(backend):
    sim = ClusterSim(backend)                                                                                                      # S:1895 (and S:1864, issue #2011: the same flow through a ProviderRef)
    a = mkpod(limits={"cpu": "10m"})
    _provision_no_binding(sim, [a])
    assert len(sim.nodes) == 1 and sim.scheduled(a) is None
    res = _provision_no_binding(sim, [mkpod(limits={"cpu": "10m"})])
    assert len(sim.nodes) == 1 and not res.new_nodes      # the in-flight node takes it: no second node


def test_self_affinity_prefers_the_inflight_nodes_zone(backend):
    lab = {"security": "s2"}                                                                                                       # S:1963 (issue #1975)
    pods = mkpods(2, labels=lab, affinity_required=aff(LABEL_ZONE, lab))
    sim = ClusterSim(backend)
    _provision_no_binding(sim, [pods[0]])
    _provision_no_binding(sim, [pods[1]])
    assert len(sim.nodes) == 1      # nothing is counted anywhere yet: the existing node's zone must win over "any viable domain"


def test_extended_resources_zeroed_by_the_kubelet_at_startup(backend):
    sim = ClusterSim(backend)                                                                                                      # S:1923 (issue #1459)
    a = mkpod(limits={"cpu": "10m", fake.RES_GPU_A: "1"})
    _provision_no_binding(sim, [a])
    assert len(sim.nodes) == 1
    # the kubelet reports the extended resource as 0 until the device plugin registers; state.Node.Available() of a node that is not initialised yet
    # keeps the instance type's figures (state/node.go:131-146) -- the input Solve sees is unchanged, and the second pod must still fit the in-flight node
    res = _provision_no_binding(sim, [mkpod(limits={"cpu": "10m", fake.RES_GPU_A: "1"})])
    assert len(sim.nodes) == 1 and not res.new_nodes
