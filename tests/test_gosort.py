"""The reference orders the open nodes with Go's UNSTABLE sort.Slice (scheduler.go:183, pdqsort_func since go 1.19); parity is defined against
the canonical stable order.  `oracle.solve(..., gosort=True)` swaps in a restatement of pdqsort_func (SURVEY App. C.1,
unverifiable here: no Go toolchain) so that the unpinned region can be measured instead of asserted.  These tests pin what the restatement must
share with a stable sort; `tools/`-free, CPU only."""
import numpy as np

from karpenter_core_amd import workloads as W
from oracle import oracle_py as O


def stable(keys):
    return sorted(range(len(keys)), key=lambda i: keys[i])


def test_up_to_12_elements_it_is_insertion_sort():
    rs = np.random.RandomState(3)
    for n in range(0, 13):
        for _ in range(20):
            keys = [int(x) for x in rs.randint(0, 4, size=n)]
            assert O.gosort_order(keys) == stable(keys)


def test_a_sorted_slice_with_one_incremented_key_keeps_the_stable_order():
    """What scheduler.add sees: last call's sorted slice with ONE node's pod count raised by one.  For >= 50 elements the "looks sorted" path
    (partialInsertionSort) shifts that element to the front of the next run, like a stable sort -- when the pivot sampling does not touch it."""
    rs = np.random.RandomState(4)
    same = total = 0
    for n in (50, 64, 200, 2111):
        for _ in range(30):
            keys = sorted(int(x) for x in rs.randint(1, 6, size=n))
            i = int(rs.randint(n))
            keys[i] += 1
            total += 1
            same += O.gosort_order(keys) == stable(keys)
    assert same >= total * 0.5, (same, total)          # (the rest is exactly the unpinned region the next test measures)


def test_it_always_sorts():
    rs = np.random.RandomState(5)
    for n in (13, 49, 50, 300, 5000):
        keys = [int(x) for x in rs.randint(0, 50, size=n)]
        perm = O.gosort_order(keys)
        assert sorted(perm) == list(range(n))
        assert [keys[i] for i in perm] == sorted(keys)


def test_solves_agree_while_at_most_12_nodes_are_open():
    p = W.config1(pods=200, types=30, seed=2)
    a, b = O.solve(p), O.solve(p, gosort=True)
    assert len(a.new_nodes) <= 12
    assert a.canonical() == b.canonical()


def test_the_unpinned_region_is_real():
    """Beyond 12 open nodes the restated pdqsort and the stable order give different (both legitimate) schedules: same pods scheduled, possibly
    different nodes.  profiles/r02_gosort_config3.json records the size of the difference on BASELINE configs[2]."""
    p = W.config3(pods=2000, sizes=10, seed=7)
    a, b = O.solve(p), O.solve(p, gosort=True)
    assert len(a.new_nodes) > 12 and sorted(a.unscheduled) == sorted(b.unscheduled)
    assert abs(len(a.new_nodes) - len(b.new_nodes)) <= 0.05 * len(a.new_nodes)


def test_go_harness_exporter_round_trips():
    """tools/go_harness/export_fixture.py (the fixture the build-tagged Go test reads the day a toolchain exists): the JSON holds every field of the problem's pods, instance
    types and provisioners (EVERY entry equals the dataclass's dump: nothing the Go side would need is lost on the way out), and the lines `--want` prints are the oracle's
    result for the generator's problem -- node count, the first node's pods in commit order and its InstanceTypeOptions.  (The Go test rebuilds the objects from the JSON; no
    rebuild happens here.)"""
    import dataclasses
    import json
    import os
    import subprocess
    import sys
    from karpenter_core_amd import model as Mo, workloads as W
    from oracle import oracle_py as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = os.path.join(root, "tools", "go_harness", "export_fixture.py")
    js = json.loads(subprocess.check_output([sys.executable, exp, "config3", "--pods", "300"], text=True))
    want = subprocess.check_output([sys.executable, exp, "config3", "--pods", "300", "--want"], text=True).strip().splitlines()
    p = W.config3(300)
    assert len(js["pods"]) == len(p.pods) and len(js["instance_types"]) == len(p.instance_types) and len(js["provisioners"]) == len(p.provisioners)
    assert js["pods"] == json.loads(json.dumps([dataclasses.asdict(x) for x in p.pods])) and js["instance_types"] == json.loads(json.dumps([dataclasses.asdict(x) for x in p.instance_types]))
    res = O.solve(p)
    assert len(want) == len(res.new_nodes) + 1 and want[-1].startswith("UNSCHEDULED")
    n0 = res.new_nodes[0]
    assert want[0].split(" | ")[1].split() == [p.pods[i].uid for i in n0.pods] and want[0].split(" | ")[2].split() == sorted(n0.instance_types)
