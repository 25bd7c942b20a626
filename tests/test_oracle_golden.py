"""Pin the CPU oracle's Requirement algebra against the reference's own truth tables
(fixtures transcribed from pkg/scheduling/requirement_test.go / requirements_test.go by
tests/golden/make_requirement_tables.py)."""
import json
import os

import pytest

from oracle import oracle_py as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "requirement_tables.json")))
OPS = {k: (v["op"], v["values"]) for k, v in G["operands"].items()}


def _render(e):
    return "c=%d vals=[%s] gt=%s lt=%s" % (1 if e["complement"] else 0, ",".join(sorted(e["values"])),
                                          "-" if e["gt"] is None else e["gt"], "-" if e["lt"] is None else e["lt"])


def test_intersection_196_cells():
    assert len(G["intersection"]) == 196
    for c in G["intersection"]:
        assert O.req_intersection(OPS[c["a"]], OPS[c["b"]]) == _render(c["expect"]), c


def test_has_70_cells():
    assert len(G["has"]) == 70
    for c in G["has"]:
        assert O.req_has(OPS[c["a"]], c["value"]) == c["expect"], c


def test_operator_and_len():
    for c in G["operator"]:
        assert O.req_operator(OPS[c["a"]]) == c["expect"], c
    for c in G["len"]:
        assert O.req_len(OPS[c["a"]]) == c["expect"], c


def test_compatible_225_cells():
    assert len(G["compatible"]) == 225
    key = G["compatible_wellknown_key"]
    for c in G["compatible"]:
        a = None if c["a"] == "unconstrained" else OPS[c["a"]]
        b = None if c["b"] == "unconstrained" else OPS[c["b"]]
        assert O.reqs_compatible(key, True, a, b) == c["expect"], c


def test_custom_label_rule():
    # requirements.go:125-130: an incoming custom key must be defined on the receiver unless the
    # incoming operator is NotIn / DoesNotExist (scenarios suite_test.go:400-553).
    inA, notInA, exists, dne = ("In", ["A"]), ("NotIn", ["A"]), ("Exists", []), ("DoesNotExist", [])
    assert not O.reqs_compatible("custom", False, None, inA)
    assert not O.reqs_compatible("custom", False, None, exists)
    assert O.reqs_compatible("custom", False, None, notInA)
    assert O.reqs_compatible("custom", False, None, dne)
    assert O.reqs_compatible("custom", False, inA, inA)
    assert not O.reqs_compatible("custom", False, inA, ("In", ["B"]))
    assert O.reqs_compatible("custom", True, None, inA)


@pytest.mark.parametrize("text,milli", [
    ("100m", 100), ("1", 1000), ("1.1", 1100), ("1.8G", 1_800_000_000_000), ("100M", 100_000_000_000),
    ("10Mi", 10 * 1024 * 1024 * 1000), ("4Gi", 4 * 1024**3 * 1000), ("2Ti", 2 * 1024**4 * 1000), ("0", 0),
    ("1e3", 1_000_000), ("1500m", 1500), ("0.5", 500), ("128Gi", 128 * 1024**3 * 1000),
])
def test_quantity_exact(text, milli):
    assert O.parse_quantity_milli(text) == milli


def test_quantity_rejects_sub_milli():
    with pytest.raises(ValueError):
        O.parse_quantity_milli("0.0001")


def test_node_selector_requirement_conversion():
    """Requirement.NodeSelectorRequirement, pinned by requirement_test.go:446-463 (the wire form ToMachine emits)."""
    from karpenter_core_amd.model import RequirementOut as R
    want = {
        "exists": (R("key", True, (), None, None), ("key", "Exists", ())),
        "doesNotExist": (R("key", False, (), None, None), ("key", "DoesNotExist", ())),
        "inA": (R("key", False, ("A",), None, None), ("key", "In", ("A",))),
        "inB": (R("key", False, ("B",), None, None), ("key", "In", ("B",))),
        "inAB": (R("key", False, ("B", "A"), None, None), ("key", "In", ("A", "B"))),
        "notInA": (R("key", True, ("A",), None, None), ("key", "NotIn", ("A",))),
        "in1": (R("key", False, ("1",), None, None), ("key", "In", ("1",))),
        "in9": (R("key", False, ("9",), None, None), ("key", "In", ("9",))),
        "in19": (R("key", False, ("9", "1"), None, None), ("key", "In", ("1", "9"))),
        "notIn12": (R("key", True, ("2", "1"), None, None), ("key", "NotIn", ("1", "2"))),
        "greaterThan1": (R("key", True, (), 1, None), ("key", "Gt", ("1",))),
        "greaterThan9": (R("key", True, (), 9, None), ("key", "Gt", ("9",))),
        "lessThan1": (R("key", True, (), None, 1), ("key", "Lt", ("1",))),
        "lessThan9": (R("key", True, (), None, 9), ("key", "Lt", ("9",))),
    }
    for name, (req, expect) in want.items():
        assert req.node_selector_requirement() == expect, name


def test_requirement_any():
    """Requirement.Any as Requirements.Labels() uses it (requirement.go:152-168), pinned by requirement_test.go:408-425.  The reference draws at random where
    it has a choice; the mirror (scheduler.requirements_labels) takes the smallest admissible value -- every assertion of the reference's test must hold."""
    from karpenter_core_amd.model import RequirementOut as R
    from karpenter_core_amd.scheduler import requirements_labels

    def any_of(req):
        return requirements_labels({"key": req}).get("key", "")
    assert any_of(R("key", True, (), None, None)) != ""                     # exists
    assert any_of(R("key", False, (), None, None)) == ""                    # doesNotExist
    assert any_of(R("key", False, ("A",), None, None)) == "A" and any_of(R("key", False, ("B",), None, None)) == "B"
    assert any_of(R("key", False, ("B", "A"), None, None)) in ("A", "B")
    assert any_of(R("key", True, ("A",), None, None)) not in ("", "A")      # notInA
    assert any_of(R("key", False, ("1",), None, None)) == "1" and any_of(R("key", False, ("9",), None, None)) == "9"
    assert any_of(R("key", False, ("9", "1"), None, None)) in ("1", "9")
    assert any_of(R("key", True, ("2", "1"), None, None)) not in ("", "1", "2")      # notIn12
    assert int(any_of(R("key", True, (), 1, None))) >= 1 and 9 <= int(any_of(R("key", True, (), 9, None))) < 2**63 - 1
    assert any_of(R("key", True, (), None, 1)) == "0" and 0 <= int(any_of(R("key", True, (), None, 9))) < 9
    # a restricted key never becomes a label (Requirements.Labels, requirements.go:208-218)
    assert requirements_labels({"kubernetes.io/hostname": R("kubernetes.io/hostname", False, ("n1",), None, None), "topology.kubernetes.io/zone": R("z", False, ("z1",), None, None),
                                "node.kubernetes.io/custom": R("c", False, ("v",), None, None)}) == {"node.kubernetes.io/custom": "v"}
