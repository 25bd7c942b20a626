#!/usr/bin/env python3
"""(debug tool: imports the oracle) fuzz seeds through ks_pack's single-wave variants on the GPU (KS_FLAG_ONE_WAVE) against the oracle: tools/debug_one_wave_fuzz.py SEED..."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from karpenter_core_amd import scheduler as S  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
import test_fuzz as F  # noqa: E402
for seed in [int(x) for x in sys.argv[1:]]:
    p = F.fuzz_problem(seed)
    f = S.FlatProblem(p, flags=S.KS_FLAG_ONE_WAVE); r = f.solve(); f.close()
    w = O.solve(p)
    print(seed, "OK" if r.canonical() == w.canonical() and r.reasons == w.reasons else "MISMATCH", len(p.pods), flush=True)
