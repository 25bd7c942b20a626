#!/usr/bin/env python3
"""Round statistics of the register-resident pack kernel (ks_pack_rr) on config #3 and, with a KS_PROBES build, where a round's cycles go.
usage: tools/phase_profile_rr.py [pods]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W

pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
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
names = [("leader: form the batch (rest)", "cyc_evalout"), ("leader: wait at B1", "cyc_full"), ("leader: prepare entries", "cyc_commit"), ("leader: picks + records", "cyc_order"),
         ("leader: resolve", "cyc_new"), ("worker 1: wait at B1", "p22"), ("worker 1: leader's order", "p23"),
         ("worker 1: evaluate (new classes)", "p24"), ("worker 1: pick: candidate + atomic", "p25"), ("worker 1: pick: barrier", "p26"), ("worker 1: pick: decide", "n_kind2"), ("worker 1: commit + loop", "scan_chunks"),
         ("  form: entry lookup", "cyc_kind0"), ("  form: flags / masks", "cyc_kind1"), ("  form: cached answers", "cyc_kind2"), ("  form: dyn1 answers", "n_kind1")]
print("dyn1 answers", st.get("full_fails"), "| worker 1: commits", st.get("cyc_pop"), "touched evaluations", st.get("cyc_stage"), "| picks", st.get("queue_pops"))
for nm, k in names:
    print(f"{nm:32s} {st.get(k, 0) / rounds:9.0f} cycles / round")
