"""(GPU) The register-resident pack kernel (ks_pack_rr: it takes the single LEAN Solves it covers since round 4; KS_NO_RR=1 leaves everything to ks_pack) against
the oracle: the same C ABI, the same canonical result.  The kernel declines what it does not cover and ks_pack takes over -- `ran_rr` says which one ran."""
import hashlib
import json
import os

import pytest

from karpenter_core_amd import scheduler as S, workloads as W
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def _fp(res):
    return hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest()


def ran_rr(res) -> bool:
    """ks_pack_rr leaves its round count in stats slot 8 and the pods it placed in RUN rounds in slot 22; ks_pack (not a KS_PROBES build) leaves 0 there."""
    return res.stats.get("p22", 0) > 0 or res.stats.get("eq_pods", 0) > 0


@pytest.mark.parametrize("maker", [lambda: W.config1(pods=1000, types=50, seed=42), lambda: W.config3(pods=700, sizes=10, seed=7),
                                   lambda: W.config3(pods=3500, sizes=20, seed=44), lambda: W.config3(pods=20000, sizes=50, seed=45)])
def test_rr_matches_oracle(maker, monkeypatch):
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = maker()
    got = S.solve_problem(p)
    assert got.canonical() == O.solve(p).canonical()
    assert ran_rr(got)


@pytest.mark.parametrize("seed", [0, 1, 3, 5, 8, 13])
def test_rr_mid_scale_family(seed, monkeypatch):
    """The mid-scale family of the timed kernel (tests/test_fuzz_mid.py) through ks_pack_rr: in-flight nodes, two provisioners, relaxations, exact-filter winners."""
    import test_fuzz_mid as T
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = T.mid_problem(seed)
    gold = T._gold()[str(seed)]
    got = S.solve_problem(p)
    assert T.fingerprints(got) == {"sha256": gold["sha256"], "reasons_sha256": gold["reasons_sha256"]}


def test_rr_full_size_config3_fingerprint(monkeypatch):
    """BASELINE configs[2] at its full size (100 000 pods / 2 000 types) through ks_pack_rr: the oracle's offline fingerprint."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_hashes.json")))["config3_100k_2k"]
    res = S.solve_problem(W.config3())
    assert len(res.new_nodes) == gold["new_nodes"] and _fp(res) == gold["sha256"]
    assert ran_rr(res)


def test_rr_twice_on_one_resident_problem(monkeypatch):
    """A resident problem solved twice: the kernel re-initialises what it caches per class (rr_memo, rr_mcnrc)."""
    monkeypatch.delenv("KS_NO_RR", raising=False)
    p = W.config3(pods=3500, sizes=20, seed=44)
    fp = S.FlatProblem(p)
    try:
        a = fp.solve(); b = fp.solve()
        assert a.canonical() == b.canonical() == O.solve(p).canonical()
    finally:
        fp.close()


def test_both_pack_kernels_agree(monkeypatch):
    """The same problem through ks_pack_rr and, with KS_NO_RR=1, through ks_pack: one canonical result."""
    if os.environ.get("KS_TEST_SIM"):
        pytest.skip("the emulator build has no ks_pack to switch to (its dispatch always takes ks_pack_rr)")
    p = W.config3(pods=20000, sizes=50, seed=46)
    monkeypatch.delenv("KS_NO_RR", raising=False)
    a = S.solve_problem(p)
    monkeypatch.setenv("KS_NO_RR", "1")
    b = S.solve_problem(p)
    assert ran_rr(a) and not ran_rr(b)
    assert a.canonical() == b.canonical() == O.solve(p).canonical()
