#!/usr/bin/env python3
"""Round statistics of the register-resident pack kernel (ks_pack_rr) on config #3 and, with a KS_PROBES build, where a round's cycles go.
usage: tools/phase_profile_rr.py [pods]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W

pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
if os.environ.get("KS_VARIANT"):      # an experiment build under karpenter_core_amd/_variants/<name> (e.g. a -DKS_PROBES one) instead of the library in the tree
    S._HERE = os.path.join(os.path.dirname(S.__file__), "_variants", os.environ["KS_VARIANT"]); S._LIBS = None; S.libs()
pr = W.config3(pods=pods)
fp = S.FlatProblem(pr)
fp.upload(0)
fp.grid(want_bits=False)
fp.solve(decode=False)
r = fp.solve()
st = r.stats
tot = st["kernel_cycles"]
rounds = max(st["eq_pods"], 1)
print("kernel_ms", fp.kernel_ms, "nodes", len(r.new_nodes), "cycles", tot, "per pod", tot / pods)
print("rounds", st["eq_pods"], "evaluating", st["reuse_exhausted"], "bubbles", st["reuse_seeds"], "phase A runs", st["reuse_hits"], "nrcs", st["cyc_pop"], "exact checks", st["cyc_stage"])
names = [("leader: form the batch (rest)", "cyc_evalout"), ("leader: wait at B1", "cyc_full"), ("leader: prepare entries", "cyc_commit"), ("leader: picks / run steps", "cyc_order"),
         ("leader: resolve", "cyc_new"), ("leader: run records", "p25"), ("leader: after records", "p26"),
         ("worker 1: evaluate (new classes)", "n_kind2"), ("worker 1: round tail (incl. runs)", "scan_chunks"),
         ("worker 1 run: extraction", "cyc_kind0"), ("worker 1 run: barrier", "cyc_kind1"), ("worker 1 run: merge", "cyc_kind2"), ("worker 1 run: commit + retry", "n_kind1")]
print("dyn1 answers", st.get("full_fails"), "| runs", st.get("p24"), "pods in runs", st.get("p22"), "steps", st.get("p23"))
for nm, k in names:
    print(f"{nm:32s} {st.get(k, 0) / rounds:9.0f} cycles / round")
print("raw:", {k: v for k, v in st.items() if v})
pr, pn = st.get("relaxations", 0) >> 32, st.get("relaxations", 0) & 0xFFFFFFFF
print("run rounds: %d pods, %.0f cycles/pod | normal rounds: %d rounds, %d pods, %.0f cycles/pod, %.0f cycles/round" % (pr, st.get("attempts", 0) / max(pr, 1), st.get("n_kind2", 0), pn, st.get("types_scanned", 0) / max(pn, 1), st.get("types_scanned", 0) / max(st.get("n_kind2", 0), 1)))
steps = max(st.get("p23", 0), 1)
for nm, k in (("extraction", "cyc_kind0"), ("barrier", "cyc_kind1"), ("merge", "cyc_kind2"), ("commit + retry", "n_kind1")):
    print(f"  run step: {nm:20s} {st.get(k, 0) / steps:9.0f} cycles / step")
