#!/usr/bin/env python3
"""Kernel time of one config #3 Solve and, with a KS_PROBES build (KS_PROBES=1 python -c 'import __graft_entry__ as g;
g.build(True)'), the s_memtime phase counters of the pack kernel.

usage: tools/phase_profile.py [pods]        env KS_ONE_WAVE=1 -> single-wave kernel, KS_NO_LEAN=1 -> general variant

The probe slots (ks_result.stats[8..31]) mean different things in the two kernels:
  single wave : 12..19 pop / stage / scan / publish / filter / commit / order / new-node phases of every pod,
                27..29 cycles per pod kind, 30..31 pods of kind 1 / 2, 8..11 fit-bitmap reuse counters
  multi wave  : 12..19 the same phases but of SEQUENTIAL pods only, 28/30 cycles and count of sequential pods,
                8 pods offered to rounds, 9 rounds, 10 committed, 11 assigned, 27 round cycles,
                22..25 round phases on the leader's clock (evaluate / resolve / publish+filter | order moves / commit)
"""
import os
import sys

os.environ.setdefault("KS_NO_RR", "1")      # (this tool reads ks_pack's probe slots; tools/phase_profile_rr.py is the one for ks_pack_rr)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W

pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pr = W.config3(pods=pods)
fp = S.FlatProblem(pr)
fp.upload(0)
fp.grid(want_bits=False)
fp.solve(decode=False)
r = fp.solve()
print("kernel_ms", fp.kernel_ms, "nodes", len(r.new_nodes))
st = r.stats
tot = st["kernel_cycles"]
print("total cycles", tot, "per pod", tot / pods, "new-node filters", st["full_checks"], "failed", st["full_fails"])
if not (st.get("cyc_pop") or st.get("cyc_kind0")):
    sys.exit(0)                                   # not a KS_PROBES build
multi = not os.environ.get("KS_ONE_WAVE")
scope = "sequential pods" if multi else "all pods"
for k in ("cyc_pop", "cyc_stage", "cyc_scan", "cyc_evalout", "cyc_full", "cyc_commit", "cyc_order", "cyc_new"):
    print(f"{k:12s} {st[k]:>14d}  {100 * st[k] / tot:5.1f}%  ({scope})")
if multi:
    nr = max(st.get("reuse_exhausted", 0), 1)
    print("rounds %d, pods offered %d, assigned %d, committed %d (%.2f per round), %.0f cycles per round" % (
        st.get("reuse_exhausted", 0), st.get("eq_pods", 0), st.get("reuse_hits", 0), st.get("reuse_seeds", 0), st.get("reuse_seeds", 0) / nr,
        st.get("cyc_kind0", 0) / nr))
    print("round phases on the leader's clock (cycles/round): evaluate %.0f  resolve %.0f  publish+filter | order moves %.0f  commit %.0f" % (
        st.get("p22", 0) / nr, st.get("p23", 0) / nr, st.get("p24", 0) / nr, st.get("p25", 0) / nr))
    print("sequential pods %d at %.0f cycles each" % (st.get("n_kind1", 0), st.get("cyc_kind1", 0) / max(st.get("n_kind1", 0), 1)))
else:
    n1, n2 = st.get("n_kind1", 0), st.get("n_kind2", 0)
    n0 = st["queue_pops"] - n1 - n2
    for nm, cy, n in (("no topology in eval", st.get("cyc_kind0", 0), n0), ("narrow-key topology", st.get("cyc_kind1", 0), n1),
                      ("hostname topology", st.get("cyc_kind2", 0), n2)):
        print("%-22s pops %7d  cycles/pop %8.0f  share %4.1f%%" % (nm, n, cy / max(n, 1), 100 * cy / tot))
    print("fit-bitmap reuse: eligible pods %d, windows seeded %d, pods placed without evaluating %d, windows exhausted %d" % (
        st.get("eq_pods", 0), st.get("reuse_seeds", 0), st.get("reuse_hits", 0), st.get("reuse_exhausted", 0)))
