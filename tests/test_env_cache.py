"""The environment cache (host/encode.hpp `EnvCache`): the flattening of everything but the pods -- 2 000 instance types, templates, state-node rows,
the instance-type lattice; provisioner.go:237-296 rebuilds all of it per Solve -- is kept with the caller's objects and adopted by the next batch
that names the same label keys / values / bounds / resources.  A hit and a miss must produce the same flat problem (`ksh_fingerprint`), and a
batch with another universe must not be served from a stale one."""
import ctypes
import dataclasses
import threading

from karpenter_core_amd import scheduler as S, workloads as W
from karpenter_core_amd.model import Expr, LABEL_ARCH, pods_to_blocks
from test_fuzz import fuzz_problem
from test_fuzz_mid import mid_problem


def _open_parsed(pp):
    kh = S.libs()[1]
    kh.ksh_open_parsed.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
    h = ctypes.c_void_p()
    rc = kh.ksh_open_parsed(pp._p, 0, ctypes.byref(h))
    assert rc == 0, kh.ksh_last_error()
    return S.FlatProblem(None, _handle=h)


def test_hit_and_miss_flatten_to_the_same_problem():
    for name, pr in [("config3", W.config3(pods=4000, sizes=10, seed=3)), ("config2", W.config2(pods=1500)), ("config5", W.config5(pods=2500, sizes=10)),
                     ("fuzz3", fuzz_problem(3)), ("fuzz7", fuzz_problem(7)), ("fuzz19", fuzz_problem(19)), ("mid14", mid_problem(14))]:
        ref = S.FlatProblem(pr).fingerprint()               # text route: no cache behind it
        pp = S.ParsedProblem(pr)
        assert _open_parsed(pp).fingerprint() == ref, name   # miss: builds the cached flattening, then takes the cached road itself
        assert _open_parsed(pp).fingerprint() == ref, name   # hit


def test_batches_over_one_environment():
    pr = W.config3(pods=6000, sizes=10, seed=3)
    env = S.ParsedProblem(dataclasses.replace(pr, pods=[]))
    halves = [pr.pods[:3000], pr.pods[3000:]]
    for pods in halves + halves[:1]:
        got = S.open_batch(env, S.PodBatch(pods_to_blocks(pods, 2)))
        assert got.fingerprint() == S.FlatProblem(dataclasses.replace(pr, pods=pods)).fingerprint()
    # a batch that names a key the cached universe does not know: served by a fresh flattening, not by the stale one
    odd = [dataclasses.replace(p) for p in pr.pods[:500]]
    odd[7] = dataclasses.replace(odd[7], required_affinity=[[Expr("example.com/rack", "NotIn", ["r1"])]])
    odd[9] = dataclasses.replace(odd[9], node_selector={LABEL_ARCH: "arm64"})
    got = S.open_batch(env, S.PodBatch(pods_to_blocks(odd, 1)))
    assert got.fingerprint() == S.FlatProblem(dataclasses.replace(pr, pods=odd)).fingerprint()
    got = S.open_batch(env, S.PodBatch(pods_to_blocks(halves[1], 1)))          # and back
    assert got.fingerprint() == S.FlatProblem(dataclasses.replace(pr, pods=halves[1])).fingerprint()


def test_concurrent_batches_share_the_cache():
    pr = W.config3(pods=8000, sizes=10, seed=5)
    pp = S.ParsedProblem(pr)
    want = S.FlatProblem(pr).fingerprint()
    got, errs = [], []

    def run():
        try:
            for _ in range(3):
                got.append(_open_parsed(pp).fingerprint())
        except BaseException as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=run) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and got == [want] * 12


def test_a_failed_spec_confirmation_starts_the_flattening_over(monkeypatch):
    """Round 6: a large batch's spec merges are confirmed field by field on a thread of their own, beside the flattening; if that ever finds a merge that does not
    hold (two specs with one 128-bit hash) the flattening starts over with the confirmation in line.  KSH_TEST_CONFIRM_FAILS makes the first attempt report one."""
    from karpenter_core_amd import scheduler as S, workloads as W
    pp = S.ParsedProblem(W.config3(pods=9000, seed=3))
    kh = S.libs()[1]

    def fingerprint():
        h = ctypes.c_void_p()
        assert kh.ksh_open_parsed(pp._p, 0, ctypes.byref(h)) == 0, kh.ksh_last_error()
        f = int(kh.ksh_fingerprint(h)); kh.ksh_close(h)
        return f
    kh.ksh_open_parsed.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
    want = fingerprint()
    monkeypatch.setenv("KSH_SYNC_CONFIRM", "1"); monkeypatch.setenv("KSH_SYNC_QUEUE_SORT", "1")
    assert fingerprint() == want          # nothing beside the flattening: the same flat problem
    monkeypatch.delenv("KSH_SYNC_CONFIRM"); monkeypatch.delenv("KSH_SYNC_QUEUE_SORT")
    monkeypatch.setenv("KSH_TEST_CONFIRM_FAILS", "1")
    assert fingerprint() == want          # the first attempt was thrown away
    pp.close()
