#!/usr/bin/env python3
"""Diagnostics: evaluation-class ids of a config #3 problem (how many distinct classes the round planner sees)."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from karpenter_core_amd import scheduler as S, workloads as W
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
fp = S.FlatProblem(W.config3(pods=pods)); fp.upload(0)
kh = S.libs()[1]
C = fp.dims["C"]
briefs = np.zeros((C, 16), dtype=np.uint64)
plans = np.zeros((C, 4096), dtype=np.uint8)
kh.ksh_debug_classes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
psz = kh.ksh_debug_classes(fp._h, briefs.ctypes.data, None)
plans = np.zeros((C, psz), dtype=np.uint8)
kh.ksh_debug_classes(fp._h, briefs.ctypes.data, plans.ctypes.data)
ev = (briefs[:, 3] & 0xFFFFFFFF).astype(np.int64)
print("classes", C, "distinct evaluation classes", len(set(ev.tolist())), "plan bytes", psz)
groups = collections.defaultdict(list)
for c in range(C): groups[int(ev[c])].append(c)
print("largest groups", sorted((len(v) for v in groups.values()), reverse=True)[:10])
w = plans.view(np.uint32)
a, b = 0, 1
diff = [i for i in range(w.shape[1]) if w[a, i] != w[b, i]]
print("classes 0 and 1 differ in plan words", diff[:40], "ev", ev[a], ev[b])
