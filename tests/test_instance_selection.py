"""E26 -- pkg/controllers/provisioning/scheduling/instance_selection_test.go:72-588 restated as pure Solve() fixtures: the 1 344-type assorted
catalogue (fake.InstanceTypesAssorted), SHUFFLED like the reference does (:62-66), with constraints placed on the provisioner and / or the pod.
Invariants the reference asserts: the node is launched at the global minimum price (`nodePrice(node) == minPrice`, every constraint combination
admits a cheapest-priced type), and EVERY instance type handed to the cloud provider satisfies the constraint (ExpectInstancesWithLabel /
ExpectInstancesWithOffering, :605-650).  Runs on the CPU oracle and, on the GPU box, on the HIP path."""
import numpy as np
import pytest

from helpers import BACKENDS, ClusterSim, mkpod
from karpenter_core_amd import fake
from karpenter_core_amd.model import Expr, Offering, LABEL_ARCH, LABEL_CAPACITY_TYPE, LABEL_OS, LABEL_ZONE

pytestmark = pytest.mark.parametrize("backend", BACKENDS)


def catalogue(seed=0):
    its = fake.instance_types_assorted()
    np.random.RandomState(seed).shuffle(its)                 # instance_selection_test.go:62-66
    return its


def min_price(its):                                          # getMinPrice :583-592
    return min(o.price for it in its for o in it.offerings)


def launched_price(sim, pod):                                # nodePrice :563-571
    node = sim.scheduled(pod)
    it = {i.name: i for i in sim.instance_types}[sim.node_types[node.name]]
    return [o.price for o in it.offerings if o.capacity_type == node.labels[LABEL_CAPACITY_TYPE] and o.zone == node.labels[LABEL_ZONE]][0]


def requirement(key, value):
    return [Expr(key, "In", [value])]


def check(its, label, value):
    for it in its:
        if label in (LABEL_ARCH, LABEL_OS):
            e = [r for r in it.requirements if r.key == label][0]
            assert value in e.values, (it.name, label, value)
        elif label == LABEL_ZONE:
            assert any(o.zone == value for o in it.offerings), (it.name, value)
        else:
            assert any(o.capacity_type == value for o in it.offerings), (it.name, value)


# (provisioner requirements, pod node-selector requirements) -- one row per `It` of :72-396
CASES = [
    ([], []),
    ([], [(LABEL_ARCH, "amd64")]), ([], [(LABEL_ARCH, "arm64")]), ([(LABEL_ARCH, "amd64")], []), ([(LABEL_ARCH, "arm64")], []),
    ([(LABEL_OS, "windows")], []), ([], [(LABEL_OS, "windows")]), ([], [(LABEL_OS, "linux")]),
    ([(LABEL_ZONE, "test-zone-2")], []), ([], [(LABEL_ZONE, "test-zone-2")]),
    ([(LABEL_CAPACITY_TYPE, "spot")], []), ([], [(LABEL_CAPACITY_TYPE, "spot")]),
    ([(LABEL_CAPACITY_TYPE, "on-demand"), (LABEL_ZONE, "test-zone-1")], []),
    ([], [(LABEL_CAPACITY_TYPE, "spot"), (LABEL_ZONE, "test-zone-1")]),
    ([(LABEL_CAPACITY_TYPE, "spot")], [(LABEL_ZONE, "test-zone-2")]),
    ([(LABEL_CAPACITY_TYPE, "on-demand"), (LABEL_ZONE, "test-zone-1"), (LABEL_ARCH, "arm64"), (LABEL_OS, "windows")], []),
    ([(LABEL_CAPACITY_TYPE, "spot"), (LABEL_ZONE, "test-zone-2")], [(LABEL_ARCH, "amd64"), (LABEL_OS, "linux")]),
    ([], [(LABEL_CAPACITY_TYPE, "spot"), (LABEL_ZONE, "test-zone-2"), (LABEL_ARCH, "amd64"), (LABEL_OS, "linux")]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cheapest_valid_instance(backend, case):
    prov_reqs, pod_reqs = CASES[case]
    its = catalogue(case)
    prov = fake.provisioner("default", len(its), requirements=[Expr(k, "In", [v]) for k, v in prov_reqs], discovery_label=True)
    sim = ClusterSim(backend, instance_types=its, provisioners=[prov])
    pod = mkpod(required_affinity=[[Expr(k, "In", [v]) for k, v in pod_reqs]] if pod_reqs else [])
    res = sim.provision([pod])
    assert sim.scheduled(pod) is not None
    assert launched_price(sim, pod) == min_price(its)
    by_name = {i.name: i for i in its}
    options = [by_name[n] for n in res.new_nodes[0].instance_types]       # supportedInstanceTypes(cloudProv.CreateCalls[0])
    assert options
    for k, v in prov_reqs + pod_reqs:
        check(options, k, v)
    if dict(prov_reqs + pod_reqs).keys() >= {LABEL_CAPACITY_TYPE, LABEL_ZONE}:       # ExpectInstancesWithOffering
        d = dict(prov_reqs + pod_reqs)
        for it in options:
            assert any(o.capacity_type == d[LABEL_CAPACITY_TYPE] and o.zone == d[LABEL_ZONE] for o in it.offerings)


@pytest.mark.parametrize("prov_reqs,pod_reqs", [
    ([], [(LABEL_ARCH, "arm")]),                                           # :398-415  no such architecture
    ([], [(LABEL_ARCH, "arm"), (LABEL_ZONE, "test-zone-2")]),              # :417-443
    ([(LABEL_ARCH, "arm")], [(LABEL_ZONE, "test-zone-2")]),                # :445-474
])
def test_no_instance_type_matches(backend, prov_reqs, pod_reqs):
    its = catalogue(99)
    prov = fake.provisioner("default", len(its), requirements=[Expr(k, "In", [v]) for k, v in prov_reqs], discovery_label=True)
    sim = ClusterSim(backend, instance_types=its, provisioners=[prov])
    pod = mkpod(required_affinity=[[Expr(k, "In", [v]) for k, v in pod_reqs]])
    sim.provision([pod])
    assert sim.scheduled(pod) is None


def test_enough_resources(backend):
    """:476-526 -- three equal pods always share ONE node, and every instance type offered to the provider holds them plus its overhead with room to
    spare (strictly less than capacity in cpu and memory)."""
    from karpenter_core_amd.model import parse_quantity_milli
    its = catalogue(7)
    for cpu in (0.1, 1.0, 2, 2.5, 4, 8, 16):
        for mem in (0.1, 2, 8, 32):                                        # (a sub-grid of the reference's 7 x 7: same corners)
            sim = ClusterSim(backend, instance_types=its, provisioners=[fake.provisioner("default", len(its), discovery_label=True)])
            pods = [mkpod(requests={"cpu": f"{cpu:.1f}", "memory": f"{mem:.1f}Gi"}) for _ in range(3)]
            res = sim.provision(pods)
            assert len({sim.scheduled(p).name for p in pods}) == 1
            by_name = {i.name: i for i in its}
            need_cpu, need_mem = 3 * parse_quantity_milli(f"{cpu:.1f}"), 3 * parse_quantity_milli(f"{mem:.1f}Gi")
            for n in res.new_nodes[0].instance_types:
                it = by_name[n]
                assert need_cpu + parse_quantity_milli(it.overhead["cpu"]) < parse_quantity_milli(it.capacity["cpu"])
                assert need_mem + parse_quantity_milli(it.overhead["memory"]) < parse_quantity_milli(it.capacity["memory"])


def test_on_demand_price_decides_when_spot_would_order_differently(backend):
    """:528-573 -- two types whose spot prices order one way and on-demand prices the other; the provisioner only allows on-demand."""
    res = {"cpu": "1", "memory": "1Gi"}
    a = fake.new_instance_type("test-instance1", res, [Offering("on-demand", "test-zone-1a", 1.0), Offering("spot", "test-zone-1a", 0.2)], "amd64", ["linux"])
    b = fake.new_instance_type("test-instance2", res, [Offering("on-demand", "test-zone-1a", 1.3), Offering("spot", "test-zone-1a", 0.1)], "amd64", ["linux"])
    prov = fake.provisioner("default", 2, requirements=[Expr(LABEL_CAPACITY_TYPE, "In", ["on-demand"])], discovery_label=True)
    sim = ClusterSim(backend, instance_types=[a, b], provisioners=[prov])
    pod = mkpod()
    sim.provision([pod])
    assert sim.node_types[sim.scheduled(pod).name] == "test-instance1"
