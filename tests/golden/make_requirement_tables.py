#!/usr/bin/env python3
"""Transcribe the reference's own truth tables for the Requirement algebra into JSON fixtures.

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_requirement_tables.py

Sources (reference, read-only):
  pkg/scheduling/requirement_test.go:29-42   operand definitions
  pkg/scheduling/requirement_test.go:81-293  Intersection 14x14
  pkg/scheduling/requirement_test.go:294-371 Has 14x{A,B,1,2,9}
  pkg/scheduling/requirement_test.go:372-389 Operator; :390-407 Len
  pkg/scheduling/requirements_test.go:50-290 Compatible 15x15 (well-known key)
Output: tests/golden/requirement_tables.json (committed).
"""
import json
import os
import re
import sys

REF = "/root/reference/pkg/scheduling"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "requirement_tables.json")

OPMAP = {"NodeSelectorOpIn": "In", "NodeSelectorOpNotIn": "NotIn", "NodeSelectorOpExists": "Exists",
         "NodeSelectorOpDoesNotExist": "DoesNotExist", "NodeSelectorOpGt": "Gt", "NodeSelectorOpLt": "Lt"}


def parse_operands(src):
    ops = {}
    for m in re.finditer(r'(\w+) := NewRequirement\("key", v1\.(\w+)((?:, "[^"]*")*)\)', src):
        name, op, vals = m.group(1), OPMAP[m.group(2)], re.findall(r'"([^"]*)"', m.group(3))
        ops[name] = {"op": op, "values": vals}
    return ops


def canon(op, values):
    """Canonical (complement, values, gt, lt) of NewRequirement (requirement.go:44-68)."""
    c = op not in ("In", "DoesNotExist")
    vals = sorted(values) if op in ("In", "NotIn") else []
    gt = int(values[0]) if op == "Gt" else None
    lt = int(values[0]) if op == "Lt" else None
    return {"complement": c, "values": vals, "gt": gt, "lt": lt}


def main():
    src = open(os.path.join(REF, "requirement_test.go")).read()
    operands = parse_operands(src)
    assert len(operands) == 14, operands.keys()
    inter = []
    for m in re.finditer(r'Expect\((\w+)\.Intersection\((\w+)\)\)\.To\(Equal\((.*)\)\)\s*$', src, re.M):
        a, b, rhs = m.group(1), m.group(2), m.group(3)
        if rhs in operands:
            exp = canon(operands[rhs]["op"], operands[rhs]["values"])
        else:
            mm = re.match(r'&Requirement\{Key: "key", complement: (true|false)(.*)\}$', rhs)
            assert mm, rhs
            rest = mm.group(2)
            vals = re.search(r'values: sets\.NewString\(([^)]*)\)', rest)
            values = sorted(re.findall(r'"([^"]*)"', vals.group(1))) if vals else []
            gt = re.search(r'greaterThan: (\w+)\.greaterThan', rest)
            lt = re.search(r'lessThan: (\w+)\.lessThan', rest)
            exp = {"complement": mm.group(1) == "true", "values": values,
                   "gt": int(operands[gt.group(1)]["values"][0]) if gt else None,
                   "lt": int(operands[lt.group(1)]["values"][0]) if lt else None}
        inter.append({"a": a, "b": b, "expect": exp})
    assert len(inter) == 196, len(inter)
    has = []
    for m in re.finditer(r'Expect\((\w+)\.Has\("([^"]*)"\)\)\.To\(Be(True|False)\(\)\)', src):
        has.append({"a": m.group(1), "value": m.group(2), "expect": m.group(3) == "True"})
    assert len(has) == 70, len(has)
    oper = []
    for m in re.finditer(r'Expect\((\w+)\.Operator\(\)\)\.To\(Equal\(v1\.(\w+)\)\)', src):
        oper.append({"a": m.group(1), "expect": OPMAP[m.group(2)]})
    assert len(oper) == 14
    lens = []
    for m in re.finditer(r'Expect\((\w+)\.Len\(\)\)\.To\(Equal\(([^)]*)\)\)', src):
        e = m.group(2).replace("math.MaxInt64", str(2**63 - 1))
        lens.append({"a": m.group(1), "expect": eval(e)})
    assert len(lens) == 14

    src2 = open(os.path.join(REF, "requirements_test.go")).read()
    compat = []
    for m in re.finditer(r'Expect\((\w+)\.Compatible\((\w+)\)\)\.(To|ToNot)\(Succeed\(\)\)', src2):
        compat.append({"a": m.group(1), "b": m.group(2), "expect": m.group(3) == "To"})
    assert len(compat) == 225, len(compat)
    # custom-label rule is exercised by the scheduling suites only (suite_test.go:400-553); the table
    # above uses the well-known zone key.
    out = {
        "source": "aws/karpenter-core pkg/scheduling/requirement_test.go:29-463, requirements_test.go:36-290",
        "operands": operands,
        "intersection": inter, "has": has, "operator": oper, "len": lens,
        "compatible_wellknown_key": "topology.kubernetes.io/zone",
        "compatible": compat,
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(inter)} intersection, {len(has)} has, {len(oper)} operator, {len(lens)} len, {len(compat)} compatible")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures are already committed")
    main()
