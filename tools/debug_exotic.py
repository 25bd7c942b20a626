"""Debug helper: derived vs flattened what-ifs over the exotic clusters of tests/test_whatif_derived.py (`_exotic_snapshot`), solved on the GPU."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                               # noqa: E402
from test_whatif_derived import _exotic_snapshot                 # noqa: E402
from karpenter_core_amd import scheduler as S                    # noqa: E402
first, count = int(sys.argv[1]), int(sys.argv[2])
ok = bad = uns = 0
for seed in range(first, first + count):
    rs = np.random.RandomState(seed + 7)
    nodes, snap, pod_node = _exotic_snapshot(seed)
    sets = [[int(x) for x in rs.choice(len(nodes), size=int(rs.choice([1, 1, 2, 3, 6])), replace=False)] for _ in range(16)]
    try:
        parsed = S.ParsedProblem(snap)
        derived = S.open_whatifs(parsed, pod_node, sets, derive=True)
        flat = S.open_whatifs(parsed, pod_node, sets, derive=False)
    except S.KSolveError as e:
        uns += 1
        continue
    got, _, _ = S.solve_batch(derived)
    want, _, _ = S.solve_batch(flat)
    for i, (a, b) in enumerate(zip(got, want)):
        if a.canonical() == b.canonical() and a.reasons == b.reasons:
            ok += 1
        else:
            bad += 1
            print("MISMATCH seed", seed, "what-if", i, sets[i])
    for f in derived + flat:
        f.close()
print("exotic: equal", ok, "different", bad, "snapshots refused", uns)
