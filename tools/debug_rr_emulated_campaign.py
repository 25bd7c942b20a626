#!/usr/bin/env python3
"""(debug tool: imports the oracle; CPU only) the rr-covered fuzz family through ks_pack_rr's SOURCE on the lane-fibre emulator against the oracle, in worker processes:
   tools/debug_rr_emulated_campaign.py FIRST_SEED COUNT [PROCS] [FAMILY]      FAMILY: rr (default) | small | base | wide | general | config3 (the benchmark's shape at 500-4 500 pods, 5-24 sizes) | herd (hostname groups crowded: the census of zero counters) -- the other fuzz families (what ks_pack_rr
   declines, and everything small, goes through ks_pack's single-wave variants there: the kernel of every what-if)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(job):
    seed, fam = job
    import simlib
    S = simlib.use_sim()
    from oracle import oracle_py as O
    import test_fuzz_rr as R, test_fuzz as F, test_fuzz_mid as M
    p = {"rr": lambda: R.rr_problem(seed), "small": lambda: F.fuzz_problem(seed), "base": lambda: M.mid_problem(seed, "base"), "wide": lambda: M.mid_problem_wide(seed), "general": lambda: M.mid_problem_general(seed),
         "config3": lambda: __import__("karpenter_core_amd.workloads", fromlist=["config3"]).config3(pods=500 + (seed * 37) % 4000, sizes=5 + seed % 20, seed=seed),
         "herd": lambda: __import__("karpenter_core_amd.workloads", fromlist=["hostname_herd"]).hostname_herd(pods=200 + (seed * 53) % 1500, labels=2 + seed % 6, seed=seed)}[fam](); t0 = time.time()
    want = O.solve(p)
    f = S.FlatProblem(p)
    try:
        got = f.solve(); st = f.rr_status()
    finally:
        f.close()
    ok = got.canonical() == want.canonical() and got.reasons == want.reasons
    return seed, "ok" if ok else "MISMATCH", len(p.pods), tuple(st), round(time.time() - t0, 1)


if __name__ == "__main__":
    import multiprocessing as mp
    first, count = int(sys.argv[1]), int(sys.argv[2]); procs = int(sys.argv[3]) if len(sys.argv) > 3 else 6; fam = sys.argv[4] if len(sys.argv) > 4 else "rr"
    os.environ.setdefault("KS_SIM_ALARM", "900")
    import simlib; simlib.use_sim()      # (build once, before the workers)
    with mp.get_context("spawn").Pool(procs) as pool:
        bad, took = [], 0
        for seed, verdict, pods, st, secs in pool.imap_unordered(one, [(sd, fam) for sd in range(first, first + count)]):
            took += st == (1, 0)
            if verdict != "ok": bad.append(seed)
            print(seed, verdict, pods, st, secs, flush=True)
    print("emulated campaign", fam, first, count, "taken by rr", took, "bad", bad)
