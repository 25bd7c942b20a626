"""The bitmask form of the Requirement algebra (karpenter_core_amd/csrc/ks_algebra.h -- the functions the
HIP kernels call) against the reference's own truth tables: host build here, device build on the GPU."""
import ctypes
import json
import os

import pytest

import __graft_entry__ as ge
from karpenter_core_amd import scheduler as S

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "requirement_tables.json")))
UNIVERSE = ["1", "2", "9", "A", "B"]            # ascending byte-wise order
VINT = [1, 2, 9, -2**31, -2**31]
NOGT, NOLT = -2**31, 2**31 - 1


class Req1(ctypes.Structure):
    _fields_ = [("mask", ctypes.c_uint64), ("gt", ctypes.c_int32), ("lt", ctypes.c_int32), ("present", ctypes.c_uint8), ("complement", ctypes.c_uint8)]


def enc(op, values):
    r = Req1(0, NOGT, NOLT, 1, 0 if op in ("In", "DoesNotExist") else 1)
    if op in ("In", "NotIn"):
        for v in values:
            r.mask |= 1 << UNIVERSE.index(v)
    if op == "Gt":
        r.gt = int(values[0])
    if op == "Lt":
        r.lt = int(values[0])
    return r


def dec(r):
    return {"complement": bool(r.complement), "values": sorted(UNIVERSE[i] for i in range(len(UNIVERSE)) if (r.mask >> i) & 1),
            "gt": None if r.gt == NOGT else r.gt, "lt": None if r.lt == NOLT else r.lt}


OPS = {k: (v["op"], v["values"]) for k, v in G["operands"].items()}


def run_tables(on_device):
    ge.build()
    ks, _ = S.libs()
    vint = (ctypes.c_int32 * 64)(*(VINT + [-2**31] * (64 - len(VINT))))
    for c in G["intersection"]:
        out = Req1()
        rc = ks.ks_probe_intersection(ctypes.byref(enc(*OPS[c["a"]])), ctypes.byref(enc(*OPS[c["b"]])), vint, len(UNIVERSE), on_device, ctypes.byref(out))
        assert rc == 0, ks.ks_last_error()
        assert dec(out) == c["expect"], c
    for c in G["compatible"]:
        absent = Req1(0, NOGT, NOLT, 0, 0)
        a = absent if c["a"] == "unconstrained" else enc(*OPS[c["a"]])
        b = absent if c["b"] == "unconstrained" else enc(*OPS[c["b"]])
        ok = ctypes.c_int()
        rc = ks.ks_probe_compatible(ctypes.byref(a), ctypes.byref(b), 1, vint, len(UNIVERSE), on_device, ctypes.byref(ok))
        assert rc == 0, ks.ks_last_error()
        assert bool(ok.value) == c["expect"], c
    # custom-label rule (requirements.go:125-130)
    for b, want in ((("In", ["A"]), False), (("Exists", []), False), (("NotIn", ["A"]), True), (("DoesNotExist", []), True)):
        ok = ctypes.c_int()
        ks.ks_probe_compatible(ctypes.byref(Req1(0, NOGT, NOLT, 0, 0)), ctypes.byref(enc(*b)), 0, vint, len(UNIVERSE), on_device, ctypes.byref(ok))
        assert bool(ok.value) == want, b


class Facts(ctypes.Structure):
    _fields_ = [("has_mask", ctypes.c_uint64), ("len", ctypes.c_int64), ("op", ctypes.c_int32), ("nidne", ctypes.c_uint8), ("len0", ctypes.c_uint8)]


def run_has_operator_len(on_device):
    """Requirement.Has 14 x {A, B, 1, 2, 9} (requirement_test.go:294-371), Operator (:372-389), Len (:390-407) on the functions the kernels
    use (kreq_has_mask and the two predicates kreq_nidne / kreq_len0 derived from Operator / Len)."""
    ge.build()
    ks, _ = S.libs()
    ks.ks_probe_has.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    vint = (ctypes.c_int32 * 64)(*(VINT + [-2**31] * (64 - len(VINT))))
    facts = {}
    for name, (op, values) in OPS.items():
        f = Facts()
        rc = ks.ks_probe_has(ctypes.byref(enc(op, values)), vint, len(UNIVERSE), on_device, ctypes.byref(f))
        assert rc == 0, ks.ks_last_error()
        facts[name] = f
    for c in G["has"]:
        assert bool((facts[c["a"]].has_mask >> UNIVERSE.index(c["value"])) & 1) == c["expect"], c
    names = ["In", "NotIn", "Exists", "DoesNotExist"]
    for c in G["operator"]:
        assert names[facts[c["a"]].op] == c["expect"], c
        assert bool(facts[c["a"]].nidne) == (c["expect"] in ("NotIn", "DoesNotExist")), c
    for c in G["len"]:
        assert facts[c["a"]].len == c["expect"], c
        assert bool(facts[c["a"]].len0) == (c["expect"] == 0), c
    assert len(G["has"]) == 70 and len(G["operator"]) == 14 and len(G["len"]) == 14


def test_has_operator_len_host_build():
    run_has_operator_len(0)


@pytest.mark.gpu
def test_has_operator_len_on_device():
    run_has_operator_len(1)


def test_tables_host_build():
    run_tables(0)


@pytest.mark.gpu
def test_tables_on_device():
    run_tables(1)
