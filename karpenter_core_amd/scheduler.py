"""Host-side mirror of the reference scheduler interface for the hot path, over the C ABI.

Reference call sites this stands in for (paths relative to aws/karpenter-core pkg/):

    scheduler, err := p.NewScheduler(ctx, pods, stateNodes, opts)     controllers/provisioning/provisioner.go:301
    nodes, existing, err := scheduler.Solve(ctx, pods)                 provisioner.go:307, deprovisioning/helpers.go:93

`NewScheduler(...)` assembles the semantic problem (provisioners -> machine templates, instance types,
state nodes, cluster pods for topology counting, daemonset pods) and hands it to libkshost.so, which
runs the host half (NewTopology / flattening, host/encode.cpp) and calls libksolve.so's HIP kernels
through the C ABI (include/ksolve.h).  `Scheduler.Solve(pods)` returns `(new_nodes, existing_nodes,
None)` -- the error is always None, exactly like scheduler.go:132.

There is NO CPU scheduling path: if the HIP library is missing, or no gfx950 device is visible, every
call raises.  (The CPU oracle lives in oracle/ and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from .model import (ClusterPod, InstanceType, NewNodeOut, Pod, Problem, Provisioner, SolveResult, StateNode,
                    parse_result)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = None

KS_OK, KS_ERR_INVALID, KS_ERR_UNSUPPORTED, KS_ERR_DEVICE, KS_ERR_CAPACITY = 0, -1, -2, -3, -4
KS_FLAG_SIMULATION, KS_FLAG_STATS, KS_FLAG_NO_RR, KS_FLAG_ONE_WAVE, KS_FLAG_NO_LEAN = 1, 2, 4, 8, 16


class KSolveError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ksolve error {code}: {msg}")
        self.code = code


def libs():
    """Load libksolve.so (HIP kernels + C ABI) and libkshost.so (host side).  Fails loudly."""
    global _LIBS
    if _LIBS is None:
        ks_path = os.path.join(_HERE, "libksolve.so")
        kh_path = os.path.join(_HERE, "libkshost.so")
        for p in (ks_path, kh_path):
            if not os.path.exists(p):
                raise KSolveError(KS_ERR_DEVICE, f"{p} is missing -- build it with __graft_entry__.build(); "
                                                 "there is no Python/CPU fallback for the scheduling path")
        ks = ctypes.CDLL(ks_path, mode=ctypes.RTLD_GLOBAL)
        kh = ctypes.CDLL(kh_path)
        ks.ks_device_count.restype = ctypes.c_int
        ks.ks_last_error.restype = ctypes.c_char_p
        ks.ks_version.restype = ctypes.c_char_p
        kh.ksh_last_error.restype = ctypes.c_char_p
        kh.ksh_open.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        kh.ksh_close.argtypes = [ctypes.c_void_p]
        kh.ksh_upload.argtypes = [ctypes.c_void_p, ctypes.c_int]
        kh.ksh_upload_batch.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32]
        kh.ksh_result_summaries.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
        kh.ksh_solve.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_solve_batch.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_grid.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        kh.ksh_grid_rows.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        kh.ksh_grid_install.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        kh.ksh_price_filter.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint32),
                                        ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        kh.ksh_dims.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        kh.ksh_rr_status.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        kh.ksh_solve_whatifs_sharded.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        ks.ks_deal_lpt.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        ks.ks_deal_lpt.restype = None
        kh.ksh_result_arrays_get.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        kh.ksh_name.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
        kh.ksh_name.restype = ctypes.c_char_p
        kh.ksh_open_whatifs.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                        ctypes.POINTER(ctypes.c_int32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        kh.ksh_parse.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        kh.ksh_parsed_free.argtypes = [ctypes.c_void_p]
        kh.ksh_solve_from_pods.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_result_text.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        kh.ksh_result_summary.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        kh.ksh_open_whatifs_parsed.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                               ctypes.POINTER(ctypes.c_int32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        kh.ksh_launch_pick.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32),
                                       ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_key_value.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int32]
        kh.ksh_key_value.restype = ctypes.c_char_p
        kh.ksh_types_subset.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32,
                                        ctypes.POINTER(ctypes.c_uint32)]
        kh.ksh_solve_batch_resident.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_result_records_dev.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32, ctypes.c_void_p]
        kh.ksh_fingerprint.argtypes = [ctypes.c_void_p]
        kh.ksh_fingerprint.restype = ctypes.c_uint64
        kh.ksh_free.argtypes = [ctypes.c_void_p]
        kh.ksh_open_whatifs_derived.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                                ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        kh.ksh_pods_ingest.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_env_ingest.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_pods_free.argtypes = [ctypes.c_void_p]
        kh.ksh_pods_count.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        kh.ksh_solve_from_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_double)]
        kh.ksh_open_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        _LIBS = (ks, kh)
    return _LIBS


def device_count() -> int:
    return int(libs()[0].ks_device_count())


class FlatProblem:
    """A Solve() problem flattened to the C-ABI `ks_problem` (host side only until `upload`)."""

    def __init__(self, problem: Optional[Problem], stats: bool = False, _handle=None, flags: int = 0):
        ks, kh = libs()
        if _handle is not None:
            self._h = _handle
        else:
            text = problem.to_ksp().encode()
            self._h = ctypes.c_void_p()
            flags = (KS_FLAG_STATS if stats else 0) | flags      # (KS_FLAG_NO_RR / _ONE_WAVE / _NO_LEAN: the kernel choice travels with the problem)
            rc = kh.ksh_open(text, len(text), flags, ctypes.byref(self._h))
            if rc != KS_OK:
                raise KSolveError(rc, kh.ksh_last_error().decode())
        d = (ctypes.c_uint32 * 10)()
        kh.ksh_dims(self._h, d)
        self.dims = dict(zip(["P", "C", "T", "M", "E", "K", "R", "G", "GH", "S"], [int(x) for x in d]))
        self.kernel_ms = None
        self.wall_ms = None

    def fingerprint(self) -> int:
        """Hash of every array of the flat problem (equal iff two construction routes flattened to the same ks_problem)."""
        return int(libs()[1].ksh_fingerprint(self._h))

    def close(self):
        if self._h:
            libs()[1].ksh_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, device: int = 0):
        rc = libs()[1].ksh_upload(self._h, device)
        if rc != KS_OK:
            raise KSolveError(rc, libs()[1].ksh_last_error().decode())

    def solve(self, decode: bool = True) -> Optional[SolveResult]:
        ks, kh = libs()
        out = ctypes.c_void_p()
        kms, wms = ctypes.c_float(), ctypes.c_double()
        rc = kh.ksh_solve(self._h, ctypes.byref(out) if decode else None, ctypes.byref(kms), ctypes.byref(wms))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        self.kernel_ms, self.wall_ms = float(kms.value), float(wms.value)
        if not decode:
            return None
        text = ctypes.string_at(out).decode()
        kh.ksh_free(out)
        return parse_result(text)

    def rr_status(self):
        """(ks_pack_rr was launched for the last solve, why it declined -- 0: it took the Solve; codes in csrc/ks_pack_rr.inc).  include/ksolve.h ks_problem_rr_status."""
        out = (ctypes.c_int * 2)()
        rc = libs()[1].ksh_rr_status(self._h, out)
        if rc != KS_OK:
            raise KSolveError(rc, "ksh_rr_status")
        return bool(out[0]), int(out[1])

    def result_arrays(self) -> dict:
        """The result through the binary door (include/kshost.h ksh_result_arrays_get): numpy copies of the arrays plus the key / resource names."""
        import numpy as np

        class RA(ctypes.Structure):
            _fields_ = [(n, ctypes.c_uint32) for n in ("n_pods", "n_existing", "n_new", "n_unscheduled", "types_words", "n_resources", "n_keys", "pad")] + \
                       [(n, ctypes.c_void_p) for n in ("pod_node", "pod_stage", "pod_reason", "unscheduled", "node_pods_off", "node_pods", "node_tmpl", "node_types", "node_requests",
                                                       "node_requests_present", "node_present", "node_complement", "node_mask", "node_gt", "node_lt", "node_it_state")]
        kh = libs()[1]
        ra = RA()
        rc = kh.ksh_result_arrays_get(self._h, ctypes.byref(ra))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())

        def arr(ptr, n, dt):
            if not n or not ptr:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()
        P, E, N, K, R, TW = ra.n_pods, ra.n_existing, ra.n_new, ra.n_keys, ra.n_resources, ra.types_words
        off = arr(ra.node_pods_off, E + N + 1, np.uint32)
        return {"n_existing": E, "n_new": N, "pod_node": arr(ra.pod_node, P, np.int32), "pod_stage": arr(ra.pod_stage, P, np.int32), "pod_reason": arr(ra.pod_reason, P, np.uint32),
                "unscheduled": arr(ra.unscheduled, ra.n_unscheduled, np.int32), "node_pods_off": off, "node_pods": arr(ra.node_pods, int(off[-1]) if len(off) else 0, np.int32),
                "node_tmpl": arr(ra.node_tmpl, N, np.int32), "node_types": arr(ra.node_types, N * TW, np.uint64).reshape(N, TW), "node_requests": arr(ra.node_requests, N * R, np.int64).reshape(N, R),
                "node_requests_present": arr(ra.node_requests_present, N, np.uint32), "node_present": arr(ra.node_present, N, np.uint32), "node_complement": arr(ra.node_complement, N, np.uint32),
                "node_mask": arr(ra.node_mask, N * K, np.uint64).reshape(N, K),
                "key_names": [(kh.ksh_name(self._h, 0, k, 0) or b"").decode() for k in range(K)], "resource_names": [(kh.ksh_name(self._h, 2, r, 0) or b"").decode() for r in range(R)],
                "key_value": lambda k, v: (kh.ksh_name(self._h, 1, k, v) or b"").decode()}

    def result(self) -> SolveResult:
        """Decode the result the handle holds (after `solve(decode=False)` or `solve_from_pods`)."""
        kh = libs()[1]
        out = ctypes.c_void_p()
        rc = kh.ksh_result_text(self._h, ctypes.byref(out))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        text = ctypes.string_at(out).decode()
        kh.ksh_free(out)
        return parse_result(text)

    def grid(self, want_bits: bool = True):
        """ks_feasibility_grid: returns (numpy uint64 [M, C, TW] or None, kernel milliseconds)."""
        import numpy as np
        kh = libs()[1]
        tw = (self.dims["T"] + 63) // 64
        arr = np.zeros((self.dims["M"], self.dims["C"], tw), dtype=np.uint64) if want_bits else None
        ms = ctypes.c_float()
        rc = kh.ksh_grid(self._h, arr.ctypes.data if arr is not None else None, ctypes.byref(ms))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        return arr, float(ms.value)


    def grid_rows(self, lo: int, hi: int, dev_ptr: int = 0):
        """ksh_grid_rows: rows [lo, hi) of the M * C grid rows computed on this handle's device; returns (numpy uint64 [hi - lo, TW], kernel ms).  dev_ptr: also
        copied device to device to that address (a slice of an all-gather's buffer)."""
        import numpy as np
        kh = libs()[1]
        tw = (self.dims["T"] + 63) // 64
        arr = np.zeros((hi - lo, tw), dtype=np.uint64)
        ms = ctypes.c_float()
        rc = kh.ksh_grid_rows(self._h, ctypes.c_uint32(lo), ctypes.c_uint32(hi), ctypes.c_void_p(arr.ctypes.data if hi > lo else None), ctypes.c_void_p(dev_ptr or None), ctypes.byref(ms))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        return arr, float(ms.value)

    def grid_install(self, lo: int, hi: int, rows=None, dev_ptr: int = 0, complete: bool = False):
        """ksh_grid_install: rows computed elsewhere put in place (numpy uint64 [hi - lo, TW] or a device address); complete: every row is in."""
        import numpy as np
        kh = libs()[1]
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.uint64)
        rc = kh.ksh_grid_install(self._h, ctypes.c_uint32(lo), ctypes.c_uint32(hi), ctypes.c_void_p(rows.ctypes.data if rows is not None and hi > lo else None), ctypes.c_void_p(dev_ptr or None),
                                 ctypes.c_int(1 if complete else 0))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())


def sharded_grid(fp: "FlatProblem", rank: int, world: int, all_gather):
    """the static feasibility grid of a Solve built by `world` GPUs -- rank r computes rows [r * MC / world, (r + 1) * MC / world) of the M * C grid rows on ITS
    device from its copy of the (small) class and catalogue tables, ONE all-gather of the bit-rows (`all_gather(numpy rows of this rank) -> list of every rank's rows, in rank
    order`: torch.distributed over RCCL / gloo in bench.py and the tests), every rank installs the others' rows.  Returns the kernel milliseconds of this rank's share."""
    mc = fp.dims["M"] * fp.dims["C"]
    cut = [mc * r // world for r in range(world + 1)]
    rows, ms = fp.grid_rows(cut[rank], cut[rank + 1])
    parts = all_gather(rows)
    for r, part in enumerate(parts):
        if r != rank:
            fp.grid_install(cut[r], cut[r + 1], rows=part)
    fp.grid_install(0, 0, complete=True)
    return ms


class ParsedProblem:
    """The problem as C++ objects in host memory (ksh_parse) -- the analogue of the []*v1.Pod, []*cloudprovider.InstanceType and
    []*state.Node a Go caller holds when it calls NewScheduler / Solve.  `solve_from_pods` starts from here."""

    def __init__(self, problem: Optional[Problem], _text: Optional[bytes] = None):
        kh = libs()[1]
        text = _text if _text is not None else problem.to_ksp().encode()
        self._p = ctypes.c_void_p()
        rc = kh.ksh_parse(text, len(text), ctypes.byref(self._p))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())

    @classmethod
    def from_text(cls, ksp_text: bytes) -> "ParsedProblem":
        return cls(None, _text=ksp_text)

    @classmethod
    def from_env_block(cls, block: dict) -> "ParsedProblem":
        """The environment through the binary door (`ksh_env_ingest`; block from `model.env_to_block`): no KSP1 text on the way.  `ingest_ms`: the library's time."""
        kh = libs()[1]
        eb = _EnvBlock(block["n_strings"], block["n_words"], block["str_off"].ctypes.data, block["str_bytes"].ctypes.data, block["words"].ctypes.data,
                       int(block.get("str_bytes_len", block["str_bytes"].size)))
        self = cls.__new__(cls)
        self._p = ctypes.c_void_p()
        ms = ctypes.c_double()
        rc = kh.ksh_env_ingest(ctypes.byref(eb), ctypes.byref(self._p), ctypes.byref(ms))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        self.ingest_ms = float(ms.value)
        return self

    def apply(self, events: Sequence[tuple], pod_node: Optional[Sequence[int]] = None) -> dict:
        """Keep the snapshot current by events instead of ingesting it again (kshost.h `ksh_env_apply`; state.Cluster's UpdateNode / DeleteNode / UpdatePod /
        DeletePod, cluster.go): `events` as `model.delta_to_ksd` takes them.  The first call needs the snapshot's bindings (`pod_node`); from then on the library
        holds them (`bindings()`), and `open_whatifs(..., pod_node=None)` means those.  Returns {"applied", "nodes", "pods", "continued", "ms"}: `continued` says the
        snapshot's flattening took the short road (same universes), `ms` is the library's time for events + flattening."""
        import numpy as np, time
        from .model import delta_to_ksd
        kh = libs()[1]
        text = delta_to_ksd(events).encode()
        pn = None if pod_node is None else np.ascontiguousarray(np.asarray(pod_node, dtype=np.int32))
        info = (ctypes.c_uint32 * 4)()
        kh.ksh_env_apply.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
        t0 = time.perf_counter()
        rc = kh.ksh_env_apply(self._p, None if pn is None or pn.size == 0 else pn.ctypes.data, text, len(text), info)
        ms = (time.perf_counter() - t0) * 1e3
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        return {"applied": int(info[0]), "nodes": int(info[1]), "pods": int(info[2]), "continued": bool(info[3]), "ms": ms}

    def bindings(self):
        """(pod -> node index, -1 for a pod that was unbound; number of node slots) as the library holds them after `apply`."""
        import numpy as np
        kh = libs()[1]
        kh.ksh_snapshot_bindings.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        n_pods, n_nodes = ctypes.c_uint32(), ctypes.c_uint32()
        if kh.ksh_snapshot_bindings(self._p, None, 0, ctypes.byref(n_pods), ctypes.byref(n_nodes)) != KS_OK:
            raise KSolveError(KS_ERR_INVALID, kh.ksh_last_error().decode())
        out = np.full(max(1, n_pods.value), -1, dtype=np.int32)
        rc = kh.ksh_snapshot_bindings(self._p, out.ctypes.data, n_pods.value, None, None)
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        return out[:n_pods.value], int(n_nodes.value)

    def snapshot_fingerprint(self, pod_node: Optional[Sequence[int]] = None, cold: bool = False) -> int:
        """Hash of the snapshot's flattening (flat problem + the tables the device derivation reads); `cold`: of one made from scratch (tests: a flattening
        continued after `apply` must equal it)."""
        import numpy as np
        kh = libs()[1]
        kh.ksh_snapshot_fingerprint.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        pn = None if pod_node is None else np.ascontiguousarray(np.asarray(pod_node, dtype=np.int32))
        out = ctypes.c_uint64()
        rc = kh.ksh_snapshot_fingerprint(self._p, None if pn is None or pn.size == 0 else pn.ctypes.data, 0, 1 if cold else 0, ctypes.byref(out))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        return int(out.value)

    def close(self):
        if self._p:
            libs()[1].ksh_parsed_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


TIMING_KEYS = ("flatten_ms", "upload_ms", "tables_grid_ms", "pack_kernel_ms", "solve_readback_ms", "total_ms")


def solve_from_pods(parsed: ParsedProblem, device: int = 0, stats: bool = False, keep: bool = True):
    """Everything the reference does inside NewScheduler + Solve for a pod list it already holds: flatten (incl. NewQueue's
    sort, per-pod requests / requirements / classes / relaxation chains), upload, static tables + feasibility grid, the pack
    kernel, read-back.  Returns (FlatProblem holding the result or None, timings dict in milliseconds)."""
    kh = libs()[1]
    h = ctypes.c_void_p()
    ms = (ctypes.c_double * 6)()
    rc = kh.ksh_solve_from_pods(parsed._p, device, KS_FLAG_STATS if stats else 0, ctypes.byref(h) if keep else None, ms)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    fp = FlatProblem(None, _handle=h) if keep else None
    if fp is not None:
        fp.kernel_ms = float(ms[3])
    return fp, dict(zip(TIMING_KEYS, [float(x) for x in ms]))


class _PodBlock(ctypes.Structure):      # include/kshost.h ksh_pod_block
    _fields_ = [("n_pods", ctypes.c_uint32), ("n_strings", ctypes.c_uint32), ("str_off", ctypes.c_void_p), ("str_bytes", ctypes.c_void_p),
                ("spec_off", ctypes.c_void_p), ("spec_words", ctypes.c_void_p), ("uid", ctypes.c_void_p), ("creation_ts", ctypes.c_void_p),
                ("str_bytes_len", ctypes.c_uint64), ("spec_words_len", ctypes.c_uint64)]


class _EnvBlock(ctypes.Structure):
    _fields_ = [("n_strings", ctypes.c_uint32), ("n_words", ctypes.c_uint32), ("str_off", ctypes.c_void_p), ("str_bytes", ctypes.c_void_p), ("words", ctypes.c_void_p), ("str_bytes_len", ctypes.c_uint64)]


class PodBatch:
    """The pending pods handed over as flat arrays (`ksh_pods_ingest`; blocks from `model.pods_to_blocks`) -- the binary door a cgo shim would
    use instead of KSP1 text.  `ingest_ms` is the library's time to take the blocks in (hash, partition, decode the distinct specs)."""

    def __init__(self, blocks: Sequence[dict]):
        kh = libs()[1]
        arr = (_PodBlock * max(1, len(blocks)))()
        for i, b in enumerate(blocks):
            arr[i] = _PodBlock(b["n_pods"], b["n_strings"], b["str_off"].ctypes.data, b["str_bytes"].ctypes.data, b["spec_off"].ctypes.data,
                               b["spec_words"].ctypes.data, b["uid"].ctypes.data, b["creation_ts"].ctypes.data,
                               int(b.get("str_bytes_len", b["str_bytes"].size)), int(b.get("spec_words_len", b["spec_words"].size)))
        self._b = ctypes.c_void_p()
        ms = ctypes.c_double()
        rc = kh.ksh_pods_ingest(ctypes.cast(arr, ctypes.c_void_p), len(blocks), ctypes.byref(self._b), ctypes.byref(ms))
        if rc != KS_OK:
            raise KSolveError(rc, kh.ksh_last_error().decode())
        self.ingest_ms = float(ms.value)
        n, s = ctypes.c_uint32(), ctypes.c_uint32()
        kh.ksh_pods_count(self._b, ctypes.byref(n), ctypes.byref(s))
        self.n_pods, self.n_specs = int(n.value), int(s.value)

    def close(self):
        if self._b:
            libs()[1].ksh_pods_free(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_batch(env: ParsedProblem, batch: PodBatch, stats: bool = False) -> FlatProblem:
    """Flatten `batch` against the environment `env` (a ParsedProblem of a Problem WITHOUT pods): host side only, like FlatProblem(problem)."""
    kh = libs()[1]
    h = ctypes.c_void_p()
    rc = kh.ksh_open_batch(env._p, batch._b, KS_FLAG_STATS if stats else 0, ctypes.byref(h))
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    return FlatProblem(None, _handle=h)


def solve_from_batch(env: ParsedProblem, batch: PodBatch, device: int = 0, stats: bool = False, keep: bool = True):
    """`solve_from_pods` for a batch that came in through the binary door."""
    kh = libs()[1]
    h = ctypes.c_void_p()
    ms = (ctypes.c_double * 6)()
    rc = kh.ksh_solve_from_batch(env._p, batch._b, device, KS_FLAG_STATS if stats else 0, ctypes.byref(h) if keep else None, ms)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    fp = FlatProblem(None, _handle=h) if keep else None
    if fp is not None:
        fp.kernel_ms = float(ms[3])
    return fp, dict(zip(TIMING_KEYS, [float(x) for x in ms]))


def open_whatifs(snapshot, pod_node: Sequence[int], candidate_sets: Sequence[Sequence[int]], threads: int = 0, stats: bool = False, derive=None, device: int = 0) -> List[FlatProblem]:
    """Flatten N consolidation what-ifs over one cluster snapshot natively (simulateScheduling, deprovisioning/helpers.go:42-115):
    `snapshot` (a `Problem`, or a `ParsedProblem` already held as objects) lists every state node and, as its pod batch, every bound pod
    (full spec); pod_node[i] = node index of pod i.  What-if w removes candidate_sets[w] from the state nodes and makes their pods
    (candidate order, then pod order) the pending batch.  The snapshot is flattened ONCE; the per-what-if part (its pods' classes, queue and
    topology groups, remainingResources) runs on `threads` host threads (0 = all usable cores)."""
    kh = libs()[1]
    n = len(candidate_sets)
    import numpy as np
    lens = np.fromiter((len(cs) for cs in candidate_sets), dtype=np.int64, count=n)
    off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(lens, out=off[1:])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(cs, dtype=np.uint32) for cs in candidate_sets]) if n else np.zeros(1, dtype=np.uint32))
    if flat.size == 0:
        flat = np.zeros(1, dtype=np.uint32)
    # pod_node None: the bindings the library holds itself since `ParsedProblem.apply`
    pn = None if pod_node is None else (np.ascontiguousarray(np.asarray(pod_node, dtype=np.int32)) if len(pod_node) else np.zeros(1, dtype=np.int32))
    c_off = off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    c_cand = flat.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    c_pn = None if pn is None else pn.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    hs = (ctypes.c_void_p * max(1, n))()
    # derive: None = derive the what-ifs on the device when the snapshot allows it (a ParsedProblem without topology terms / volume limits), else flatten
    # them one by one on the host; True = derive or raise; False = always flatten on the host.  Derived what-ifs are resident on `device` at once.
    if derive is not False and not stats and n:
        if not isinstance(snapshot, ParsedProblem):
            snapshot = ParsedProblem(snapshot)       # (the handles keep what they need of it alive)
        rc = kh.ksh_open_whatifs_derived(snapshot._p, 0, n, c_off, c_cand, c_pn, device, hs)
        if rc == KS_OK:
            return [FlatProblem(None, _handle=ctypes.c_void_p(hs[i])) for i in range(n)]
        if derive is True or rc not in (KS_ERR_UNSUPPORTED, KS_ERR_DEVICE):
            raise KSolveError(rc, kh.ksh_last_error().decode())
    elif derive is True:
        raise KSolveError(KS_ERR_UNSUPPORTED, "derived what-ifs carry no reference-algorithm statistics")
    if isinstance(snapshot, ParsedProblem):
        rc = kh.ksh_open_whatifs_parsed(snapshot._p, KS_FLAG_STATS if stats else 0, n, c_off, c_cand, c_pn, threads, hs)
    else:
        text = snapshot.to_ksp().encode()
        rc = kh.ksh_open_whatifs(text, len(text), KS_FLAG_STATS if stats else 0, n, c_off, c_cand, c_pn, threads, hs)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    return [FlatProblem(None, _handle=ctypes.c_void_p(hs[i])) for i in range(n)]


def check_whatif_derivation(snapshot: "ParsedProblem", pod_node: Sequence[int], candidates: Sequence[int]) -> None:
    """Diagnostic, no GPU needed (kshost.h `ksh_check_whatif_derivation`): what the device would derive for this candidate set -- group activity, domain
    counts, hostname rows -- restated on the host and compared with the what-if flattened by itself.  Raises KSolveError with the first difference."""
    import numpy as np
    kh = libs()[1]
    cand = np.ascontiguousarray(np.asarray(list(candidates) or [0], dtype=np.uint32))
    pn = None if pod_node is None else (np.ascontiguousarray(np.asarray(pod_node, dtype=np.int32)) if len(pod_node) else np.zeros(1, dtype=np.int32))
    kh.ksh_check_whatif_derivation.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_int32)]
    rc = kh.ksh_check_whatif_derivation(snapshot._p, 0, cand.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(candidates), None if pn is None else pn.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())


def solve_batch(flats: Sequence[FlatProblem], decode: bool = True):
    """N independent Solve() calls in one launch (consolidation what-ifs)."""
    kh = libs()[1]
    n = len(flats)
    hs = (ctypes.c_void_p * n)(*[f._h for f in flats])
    outs = (ctypes.c_void_p * n)()
    kms, wms = ctypes.c_float(), ctypes.c_double()
    rc = kh.ksh_solve_batch(hs, n, outs if decode else None, ctypes.byref(kms), ctypes.byref(wms))
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    res = None
    if decode:
        res = []
        for i in range(n):
            res.append(parse_result(ctypes.string_at(outs[i]).decode()))
            kh.ksh_free(outs[i])
    return res, float(kms.value), float(wms.value)


def deal_lpt(weights: Sequence[int], nshards: int) -> List[int]:
    """Which shard (GPU / rank) each what-if goes to: longest predicted work first, each to the least loaded shard (include/ksolve.h ks_deal_lpt) -- instead of i mod N, which
    leaves a batch as long as its longest what-if plus whatever happened to be dealt beside it."""
    n = len(weights)
    w = (ctypes.c_uint64 * max(1, n))(*[int(x) for x in weights])
    out = (ctypes.c_uint32 * max(1, n))()
    libs()[0].ks_deal_lpt(w, n, nshards, out)
    return [int(out[i]) for i in range(n)]


def solve_whatifs_sharded(shards: Sequence[Sequence[FlatProblem]], ids: Sequence[Sequence[int]], words: int):
    """The what-if fan-out in one C call (include/kshost.h ksh_solve_whatifs_sharded): every shard's what-ifs (resident on that shard's device) in one batched launch, all
    shards concurrently, the decision records gathered into one table ordered by id.  Returns ([n, 3 + words] uint64 numpy table, slowest shard's kernel milliseconds)."""
    import numpy as np
    kh = libs()[1]
    flat = [f for sh in shards for f in sh]
    n = len(flat)
    hs = (ctypes.c_void_p * max(1, n))(*[f._h for f in flat])
    off = (ctypes.c_uint32 * (len(shards) + 1))(*([0] + list(np.cumsum([len(sh) for sh in shards]))))
    idv = (ctypes.c_uint64 * max(1, n))(*[int(i) for sh in ids for i in sh])
    rows = np.zeros((n, 3 + words), dtype=np.uint64)
    kms = ctypes.c_float()
    rc = kh.ksh_solve_whatifs_sharded(hs, off, len(shards), idv, words, rows.ctypes.data_as(ctypes.c_void_p), ctypes.byref(kms))
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    return rows, float(kms.value)


def result_records(flats: Sequence[FlatProblem], ids: Sequence[int], words: int):
    """[len(flats), 3 + words] int64 numpy table of fixed-size result records `[id, n_new, n_unscheduled, first new node's
    InstanceTypeOptions]` straight from the binary results (no text round trip): what the ranks exchange after a what-if batch."""
    import numpy as np
    kh = libs()[1]
    n = len(flats)
    rows = np.zeros((n, 2 + words), dtype=np.uint64)
    hs = (ctypes.c_void_p * max(1, n))(*[f._h for f in flats])
    rc = kh.ksh_result_summaries(hs, n, rows.ctypes.data_as(ctypes.c_void_p), words)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    out = np.empty((n, 3 + words), dtype=np.uint64)
    out[:, 0] = np.asarray(ids, dtype=np.uint64)
    out[:, 1:] = rows
    return out.view(np.int64)


def solve_batch_resident(flats: Sequence[FlatProblem]):
    """The batched launch with the results left on the device (nothing but the error words is read back).  Returns (kernel ms, wall ms)."""
    kh = libs()[1]
    n = len(flats)
    hs = (ctypes.c_void_p * max(1, n))(*[f._h for f in flats])
    kms, wms = ctypes.c_float(), ctypes.c_double()
    rc = kh.ksh_solve_batch_resident(hs, n, ctypes.byref(kms), ctypes.byref(wms))
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    return float(kms.value), float(wms.value)


def result_records_dev(flats: Sequence[FlatProblem], ids: Sequence[int], words: int, out_tensor):
    """The records of `result_records`, built ON the device from the results a `solve_batch_resident` left there, into `out_tensor`: a contiguous
    int64 / uint64 device tensor [len(flats), 3 + words] on the problems' device (anything with .data_ptr()).  Complete on return."""
    kh = libs()[1]
    n = len(flats)
    hs = (ctypes.c_void_p * max(1, n))(*[f._h for f in flats])
    c_ids = (ctypes.c_uint64 * max(1, n))(*[int(x) for x in ids])
    rc = kh.ksh_result_records_dev(hs, n, c_ids, words, ctypes.c_void_p(int(out_tensor.data_ptr())))
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    return out_tensor


def upload_batch(flats: Sequence[FlatProblem], device: int = 0, threads: int = 0):
    """Upload a batch of problems (typically the what-ifs of one snapshot) on host threads."""
    kh = libs()[1]
    n = len(flats)
    hs = (ctypes.c_void_p * max(1, n))(*[f._h for f in flats])
    rc = kh.ksh_upload_batch(hs, n, device, threads)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())


def price_filter(flats: Sequence[FlatProblem], nodes: Sequence[int], max_prices: Sequence[float], spot_only: Optional[Sequence[bool]] = None) -> List[List[int]]:
    """filterByPrice (deprovisioning/helpers.go:148-157) on the device, over the results the last solve / solve_batch of
    `flats` left there: for flats[i], the instance-type indices of new node nodes[i]'s InstanceTypeOptions whose worst
    launch price is < max_prices[i] (ascending type index; the caller restores the option order).  spot_only[i]: price the
    node as if its capacity-type requirement were already In [spot] (consolidation.go:262-265)."""
    kh = libs()[1]
    n = len(flats)
    if n == 0:
        return []
    stride = max((f.dims["T"] + 63) // 64 for f in flats)
    hs = (ctypes.c_void_p * n)(*[f._h for f in flats])
    node = (ctypes.c_uint32 * n)(*[int(x) for x in nodes])
    mp = (ctypes.c_double * n)(*[float(x) for x in max_prices])
    masks = (ctypes.c_uint64 * (n * stride))()
    counts = (ctypes.c_uint32 * n)()
    so = (ctypes.c_uint32 * n)(*[1 if x else 0 for x in spot_only]) if spot_only is not None else None
    rc = kh.ksh_price_filter(hs, n, node, mp, so, masks, stride, counts)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    out = []
    for i in range(n):
        row = [w * 64 + b for w in range(stride) for b in range(64) if (masks[i * stride + w] >> b) & 1]
        assert len(row) == counts[i]
        out.append(row)
    return out


def launch_pick(flats: Sequence[FlatProblem], nodes: Sequence[int]):
    """The launch-time instance-type pick of the reference's in-memory provider (cloudprovider/fake/cloudprovider.go:79-84) on the device, over
    the results the last solve of `flats` left there: for flats[i]'s new node nodes[i], (instance-type index, zone, capacity type, price) of the
    option whose cheapest available offering under the node's zone / capacity-type requirements is cheapest (ties: lowest type index), or None.
    The (zone, capacity type) returned are THAT cheapest offering's.  The in-memory provider labels the launched node with the first
    available offering, in the type's own Offerings order, that is compatible with the requirements (fake/cloudprovider.go:92-102) -- not
    necessarily the cheapest one; the flat problem does not keep the offering order, so callers that need the provider's label choice take
    (type, price) from here and walk the type's Offerings themselves (tests/helpers.py's cluster simulator does)."""
    kh = libs()[1]
    n = len(flats)
    if n == 0:
        return []
    hs = (ctypes.c_void_p * n)(*[f._h for f in flats])
    node = (ctypes.c_uint32 * n)(*[int(x) for x in nodes])
    ty, zo, ct, pr = (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)(), (ctypes.c_double * n)()
    rc = kh.ksh_launch_pick(hs, n, node, ty, zo, ct, pr)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    out = []
    for i in range(n):
        if ty[i] < 0:
            out.append(None)
        else:
            out.append((int(ty[i]), kh.ksh_key_value(flats[i]._h, 0, zo[i]).decode(), kh.ksh_key_value(flats[i]._h, 1, ct[i]).decode(), float(pr[i])))
    return out


def types_subset(flats: Sequence[FlatProblem], nodes: Sequence[int], type_sets: Sequence[Sequence[int]]) -> List[bool]:
    """instanceTypesAreSubset (deprovisioning/helpers.go:118-122) on the device: is type_sets[i] (instance-type indices) a subset of the
    InstanceTypeOptions of flats[i]'s new node nodes[i]?"""
    kh = libs()[1]
    n = len(flats)
    if n == 0:
        return []
    stride = max((f.dims["T"] + 63) // 64 for f in flats)
    hs = (ctypes.c_void_p * n)(*[f._h for f in flats])
    node = (ctypes.c_uint32 * n)(*[int(x) for x in nodes])
    lhs = (ctypes.c_uint64 * (n * stride))()
    for i, ts in enumerate(type_sets):
        for t in ts:
            lhs[i * stride + t // 64] |= 1 << (t % 64)
    out = (ctypes.c_uint32 * n)()
    rc = kh.ksh_types_subset(hs, n, node, lhs, stride, out)
    if rc != KS_OK:
        raise KSolveError(rc, kh.ksh_last_error().decode())
    return [bool(x) for x in out]


def solve_problem(problem: Problem, stats: bool = False) -> SolveResult:
    fp = FlatProblem(problem, stats=stats)
    try:
        return fp.solve()
    finally:
        fp.close()


# ---------------------------------------------------------------------------------------------
# Reference-shaped interface
# ---------------------------------------------------------------------------------------------
@dataclass
class SchedulerOptions:
    """scheduling.SchedulerOptions, scheduler.go:36-40."""
    SimulationMode: bool = False


@dataclass
class Node:
    """scheduling.Node as its callers read it: embedded MachineTemplate fields + Pods."""
    ProvisionerName: str
    Pods: List[Pod]
    InstanceTypeOptions: List[InstanceType]
    Requirements: Dict[str, object]
    Requests: Dict[str, int]

    def ToMachine(self, provisioner: Optional[Provisioner] = None) -> dict:
        """MachineTemplate.ToMachine (machinetemplate.go:77-100) as far as Solve's result determines it: the requirements
        gain `instance-type In [names of InstanceTypeOptions]` (Requirements.Add = intersection with what is there) and
        are emitted through NodeSelectorRequirements() (requirements.go:80-84, one entry per key); requests are the
        node's accumulated requests.  Labels / taints / kubelet / provider ref come from the provisioner unchanged."""
        from .model import LABEL_INSTANCE_TYPE, RequirementOut
        reqs = dict(self.Requirements)
        names = [it.name for it in self.InstanceTypeOptions]
        cur = reqs.get(LABEL_INSTANCE_TYPE)
        if cur is None:
            merged = RequirementOut(LABEL_INSTANCE_TYPE, False, tuple(names), None, None)
        else:                                   # Intersection of `In names` with the existing requirement (requirement.go:117-150)
            keep = [n for n in names if (n not in cur.values) == cur.complement]
            merged = RequirementOut(LABEL_INSTANCE_TYPE, False, tuple(keep), None, None)
        reqs[LABEL_INSTANCE_TYPE] = merged
        out = {"generateName": self.ProvisionerName,
               "requirements": sorted(r.node_selector_requirement() for r in reqs.values()),
               "resources": {"requests": dict(sorted(self.Requests.items()))}}
        if provisioner is not None:
            out["labels"] = dict(provisioner.labels)
            out["taints"] = [(t.key, t.value, t.effect) for t in provisioner.taints]
        return out


WELL_KNOWN_LABELS = ("karpenter.sh/provisioner-name", "topology.kubernetes.io/zone", "topology.kubernetes.io/region", "node.kubernetes.io/instance-type",
                     "kubernetes.io/arch", "kubernetes.io/os", "karpenter.sh/capacity-type")
RESTRICTED_LABEL_DOMAINS = ("kubernetes.io", "k8s.io", "karpenter.sh")
LABEL_DOMAIN_EXCEPTIONS = ("kops.k8s.io", "node.kubernetes.io", "testing.karpenter.sh")
RESTRICTED_LABELS = ("karpenter.sh/emptiness-timestamp", "kubernetes.io/hostname")


def is_restricted_node_label(key: str, extra_well_known: Sequence[str] = ()) -> bool:
    """v1alpha5.IsRestrictedNodeLabel (labels.go:123-140): labels Karpenter must not put on a node itself."""
    if key in WELL_KNOWN_LABELS or key in extra_well_known:
        return True
    domain = key.split("/", 1)[0] if "/" in key else ""
    if domain in LABEL_DOMAIN_EXCEPTIONS:
        return False
    if any(domain.endswith(d) for d in RESTRICTED_LABEL_DOMAINS):
        return True
    return key in RESTRICTED_LABELS


def requirements_labels(requirements: Dict[str, object], extra_well_known: Sequence[str] = ()) -> Dict[str, str]:
    """Requirements.Labels() (requirements.go:208-218) -- the labels MachineTemplate.ToNode puts on the node object besides the provisioner's own
    (machinetemplate.go:62-74): for every key that is not restricted, Requirement.Any() (requirement.go:152-168).  Any() draws at random where the
    reference leaves a choice (one of the In values; an integer in (gt, lt) for NotIn / Exists); this mirror takes the smallest admissible one."""
    out: Dict[str, str] = {}
    for key, r in requirements.items():
        if is_restricted_node_label(key, extra_well_known):
            continue
        if not r.complement:
            if r.values:                                        # In
                out[key] = sorted(r.values)[0]
            continue                                            # DoesNotExist: ""
        lo_ = 0 if r.greater_than is None else r.greater_than + 1      # NotIn / Exists
        out[key] = str(lo_)
    return out


@dataclass
class ExistingNode:
    """scheduling.ExistingNode: the state node plus the pods Solve placed on it."""
    Node: StateNode
    Pods: List[Pod]


class Scheduler:
    """scheduling.Scheduler (scheduler.go:81-94)."""

    def __init__(self, provisioners, instance_types, state_nodes, daemonset_pods, cluster_pods, extra_well_known, opts):
        self.provisioners, self.instance_types, self.state_nodes = provisioners, instance_types, state_nodes
        self.daemonset_pods, self.cluster_pods, self.extra_well_known, self.opts = daemonset_pods, cluster_pods, extra_well_known, opts

    def Solve(self, pods: Sequence[Pod]) -> Tuple[List[Node], List[ExistingNode], None]:
        problem = Problem(instance_types=self.instance_types, provisioners=self.provisioners, pods=list(pods),
                          daemonset_pods=self.daemonset_pods, nodes=self.state_nodes, cluster_pods=self.cluster_pods,
                          extra_well_known=self.extra_well_known, simulation_mode=self.opts.SimulationMode)
        res = solve_problem(problem)
        self.last_result = res
        by_name = {it.name: it for it in self.instance_types}
        nodes = [Node(n.provisioner, [pods[i] for i in n.pods], [by_name[x] for x in n.instance_types], n.requirements, n.requests)
                 for n in res.new_nodes]
        state = {n.name: n for n in self.state_nodes}
        existing = [ExistingNode(state[name], [pods[i] for i in idxs]) for name, idxs in res.existing.items()]
        return nodes, existing, None


def NewScheduler(provisioners: Sequence[Provisioner], instance_types: Sequence[InstanceType],
                 state_nodes: Sequence[StateNode] = (), daemonset_pods: Sequence[Pod] = (),
                 cluster_pods: Sequence[ClusterPod] = (), extra_well_known: Sequence[str] = (),
                 opts: Optional[SchedulerOptions] = None) -> Scheduler:
    """provisioning.(*Provisioner).NewScheduler (provisioner.go:237-296): every provisioner offers the
    instance types listed in `Provisioner.instance_types` (indices into `instance_types`)."""
    return Scheduler(list(provisioners), list(instance_types), list(state_nodes), list(daemonset_pods), list(cluster_pods),
                     list(extra_well_known), opts or SchedulerOptions())
