#!/usr/bin/env python3
"""Host flattening alone (no GPU needed): median / min milliseconds of ksh_open_parsed on config #3.   usage: tools/time_flatten.py [pods] [reps]
   env KSH_TIMING=1 prints the phases, KSH_THREADS=n sets the worker threads."""
import ctypes, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
if os.environ.get("KS_VARIANT"):      # an experiment build of the libraries (karpenter_core_amd/_variants/<name>)
    S._HERE = os.path.join(os.path.dirname(S.__file__), "_variants", os.environ["KS_VARIANT"]); S._LIBS = None; S.libs()
pp = S.ParsedProblem(W.config3(pods=pods))
kh = S.libs()[1]
kh.ksh_open_parsed.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
ms = []
for _ in range(reps):
    h = ctypes.c_void_p(); t = time.perf_counter()
    assert kh.ksh_open_parsed(pp._p, 0, ctypes.byref(h)) == 0
    ms.append((time.perf_counter() - t) * 1e3); kh.ksh_close(h)
print("flatten ms: median %.2f  min %.2f  all %s" % (statistics.median(ms), min(ms), " ".join("%.1f" % m for m in ms)))
