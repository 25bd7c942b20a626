"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY -- never imported by the product)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    hdr = os.path.join(_HERE, "..", "karpenter_core_amd", "host", "ksp.hpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        l = ctypes.CDLL(build())
        l.ko_solve.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        l.ko_solve.restype = ctypes.c_int
        l.ko_free.argtypes = [ctypes.c_void_p]
        l.ko_solve_spec.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_void_p)]
        l.ko_solve_spec2.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_void_p)]
        l.ko_req_intersection.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        l.ko_req_has.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        l.ko_req_operator.argtypes = [ctypes.c_char_p]
        l.ko_req_len.argtypes = [ctypes.c_char_p]
        l.ko_req_len.restype = ctypes.c_longlong
        l.ko_reqs_compatible.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
        l.ko_parse_quantity_milli.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        l.ko_parse_quantity_milli.restype = ctypes.c_longlong
        _LIB = l
    return _LIB


def solve_text(ksp_text: str, inert_topology: bool = False, gosort: bool = False) -> str:
    """gosort: order the open nodes with the restated Go sort.Slice (pdqsort_func) instead of the canonical stable sort."""
    data = ksp_text.encode()
    out = ctypes.c_void_p()
    rc = lib().ko_solve(data, len(data), (1 if inert_topology else 0) | (2 if gosort else 0), ctypes.byref(out))
    text = ctypes.string_at(out).decode()
    lib().ko_free(out)
    if rc != 0:
        raise RuntimeError("oracle: " + text)
    return text


def solve_spec(problem, width: int = 8, flags: int = 0, max_classes: int = 0):
    """Model check of the kernel's round speculation: (result, counters dict).  flags bit 0: hostname-keyed anti-affinity /
    initially-present hostname spread groups do not cut rounds; max_classes: distinct evaluation classes per round (0 = any)."""
    from karpenter_core_amd.model import parse_result
    data = problem.to_ksp().encode()
    out = ctypes.c_void_p()
    ctr = (ctypes.c_longlong * 8)()
    rc = lib().ko_solve_spec2(data, len(data), width, flags, max_classes, ctr, ctypes.byref(out))
    text = ctypes.string_at(out).decode()
    lib().ko_free(out)
    if rc != 0:
        raise RuntimeError("oracle: " + text)
    names = ["predicted", "rounds", "sequential", "violations", "cut_topology", "cut_order", "pods_in_big_rounds", "largest_round"]
    return parse_result(text), {k: int(ctr[i]) for i, k in enumerate(names)}


def watermark_check(problem, mutate: bool = False, pre_topology_only: bool = False):
    """Model check of the kernel's watermark over the existing nodes (oracle.cpp solve_watermark_check): (result, counters).  mutate: also treat classes
    with spread / affinity items as watermark classes -- the claim is false for them.  pre_topology_only: every class, but only refusals that happen before
    the topology steps go on record (a rule the kernel does not use yet: DESIGN.md §8)."""
    from karpenter_core_amd.model import parse_result
    data = problem.to_ksp().encode()
    out = ctypes.c_void_p()
    ctr = (ctypes.c_longlong * 8)()
    rc = lib().ko_solve_spec2(data, len(data), 0, 8 | (16 if mutate else 0) | (32 if pre_topology_only else 0), 0, ctr, ctypes.byref(out))
    text = ctypes.string_at(out).decode()
    lib().ko_free(out)
    if rc != 0:
        raise RuntimeError("oracle: " + text)
    return parse_result(text), {"violations": int(ctr[0]), "dry_runs": int(ctr[1]), "watermark_pods": int(ctr[2]), "with_anti_affinity": int(ctr[3]),
                                "refusals_recorded": int(ctr[4]), "recorded_pairs_rechecked": int(ctr[5])}


def solve(problem, inert_topology: bool = False, gosort: bool = False):
    from karpenter_core_amd.model import parse_result
    return parse_result(solve_text(problem.to_ksp(), inert_topology, gosort))


def gosort_order(keys):
    """The restated Go sort.Slice (pdqsort_func) on integer keys: the permutation it produces."""
    n = len(keys)
    arr = (ctypes.c_int * max(1, n))(*keys)
    out = (ctypes.c_int * max(1, n))()
    lib().ko_gosort_order(arr, n, out)
    return [int(out[i]) for i in range(n)]


def _spec(op, values):
    return (op + " " + str(len(values)) + "".join(" " + v for v in values)).encode()


def req_intersection(a, b):
    buf = ctypes.create_string_buffer(4096)
    lib().ko_req_intersection(_spec(*a), _spec(*b), buf, 4096)
    return buf.value.decode()


def req_has(a, value):
    return bool(lib().ko_req_has(_spec(*a), value.encode()))


def req_operator(a):
    return ["In", "NotIn", "Exists", "DoesNotExist", "Gt", "Lt"][lib().ko_req_operator(_spec(*a))]


def req_len(a):
    return int(lib().ko_req_len(_spec(*a)))


def reqs_compatible(key, well_known, a, b):
    sa = b"-" if a is None else _spec(*a)
    sb = b"-" if b is None else _spec(*b)
    return bool(lib().ko_reqs_compatible(key.encode(), 1 if well_known else 0, sa, sb))


def parse_quantity_milli(s):
    err = ctypes.c_int()
    v = lib().ko_parse_quantity_milli(s.encode(), ctypes.byref(err))
    if err.value:
        raise ValueError(s)
    return int(v)
