"""Fuzz campaign beyond the committed seeds (debug tool: imports the oracle): mid-scale problems of the three families of tests/test_fuzz_mid.py and
the small family of tests/test_fuzz.py, the ks_pack_rr-covered family of tests/test_fuzz_rr.py (round 6), GPU == oracle on each, in worker processes (the oracle and the generators are CPU work; the GPU solves are
short).   python tools/debug_fuzz_campaign.py FIRST_SEED COUNT [PROCS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(job):
    fam, seed = job
    from oracle import oracle_py as O
    from karpenter_core_amd import scheduler as S
    import test_fuzz_mid as M
    import test_fuzz as F
    import test_fuzz_rr as R
    t0 = time.time()
    pr = {"base": lambda: M.mid_problem(seed, "base"), "wide": lambda: M.mid_problem_wide(seed), "general": lambda: M.mid_problem_general(seed), "small": lambda: F.fuzz_problem(seed), "rr": lambda: R.rr_problem(seed)}[fam]()
    try:
        want = O.solve(pr)
    except Exception as e:      # noqa: BLE001
        return fam, seed, "oracle-error " + repr(e)[:120], 0, 0.0, "-"
    kern = "?"
    try:
        fp = S.FlatProblem(pr)
        try:
            got = fp.solve()
            started, why = fp.rr_status()      # which pack kernel took it: ks_pack_rr (code 0), ks_pack after a decline (why), ks_pack alone
            kern = "rr" if started and why == 0 else (f"rr-declined-{why}" if started else "ks_pack")
        finally:
            fp.close()
    except S.KSolveError as e:
        return fam, seed, ("unsupported" if e.code == S.KS_ERR_UNSUPPORTED else "gpu-error " + str(e)[:120]), len(pr.pods), time.time() - t0, kern
    ok = got.canonical() == want.canonical() and got.reasons == want.reasons
    return fam, seed, "ok" if ok else "MISMATCH", len(pr.pods), time.time() - t0, kern


if __name__ == "__main__":
    import multiprocessing as mp
    first, count = int(sys.argv[1]), int(sys.argv[2]); procs = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    jobs = [(fam, first + i) for i in range(count) for fam in ("base", "wide", "general", "small", "rr")]
    t0 = time.time(); bad = []; n = {}; kerns = {}
    with mp.get_context("spawn").Pool(procs) as pool:
        for fam, seed, verdict, npods, dt, kern in pool.imap_unordered(one, jobs):
            n[verdict.split()[0]] = n.get(verdict.split()[0], 0) + 1
            kerns[kern] = kerns.get(kern, 0) + 1
            if verdict != "ok":
                bad.append((fam, seed, verdict)); print(fam, seed, verdict, npods, f"{dt:.1f}s", flush=True)
    print("campaign", first, count, "verdicts", n, "pack kernel", kerns, f"{time.time() - t0:.0f}s", "bad", bad[:20])
