import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pr = W.config3(pods=pods)
fp = S.FlatProblem(pr); fp.upload(0); fp.grid(want_bits=False)
fp.solve(decode=False)
r = fp.solve()
print("kernel_ms", fp.kernel_ms, "nodes", len(r.new_nodes))
st = r.stats; tot = st["kernel_cycles"]
for k in ("cyc_pop","cyc_stage","cyc_scan","cyc_evalout","cyc_full","cyc_commit","cyc_order","cyc_new"):
    print(f"{k:12s} {st[k]:>14d}  {100*st[k]/tot:5.1f}%  per pod {st[k]/pods:9.0f}")
for k,n in (("p22","to eval start"),("p23","gather loads"),("p24","taints..host topo"),("p25","touch loop"),("p26","after eval")): print(f"{k} {n:20s} {st.get(k,0)/pods:9.0f}")
print("total cycles", tot, "per pod", tot/pods, "chunks/pod", st["scan_chunks"]/pods, "full_checks", st["full_checks"], "full_fails", st["full_fails"])
n1, n2 = st.get("n_kind1",0), st.get("n_kind2",0); n0 = st["queue_pops"] - n1 - n2
for nm, cy, n in (("no topology in eval", st.get("cyc_kind0",0), n0), ("narrow-key topology", st.get("cyc_kind1",0), n1), ("hostname topology", st.get("cyc_kind2",0), n2)): print("%-22s pops %7d  cycles/pop %8.0f  share %4.1f%%" % (nm, n, cy/max(n,1), 100*cy/tot))
print("eq-eligible pods %d, window seeds %d, reuse hits %d, exhausted %d" % (st.get("eq_pods",0), st.get("reuse_seeds",0), st.get("reuse_hits",0), st.get("reuse_exhausted",0)))
if not os.environ.get("KS_ONE_WAVE"):
    print("multi-wave: rounds %d, pods offered %d, assigned %d, committed %d (%.2f per round); round cycles %.0f each; sequential pods %d at %.0f cycles" % (
        st.get("reuse_exhausted",0), st.get("eq_pods",0), st.get("reuse_hits",0), st.get("reuse_seeds",0), st.get("reuse_seeds",0)/max(st.get("reuse_exhausted",0),1),
        st.get("cyc_kind0",0)/max(st.get("reuse_exhausted",0),1), st.get("n_kind1",0), st.get("cyc_kind1",0)/max(st.get("n_kind1",0),1)))
    nr = max(st.get("reuse_exhausted",0),1)
    print("round phases (leader clock, cycles/round): evaluate %.0f  resolve %.0f  publish+filter %.0f  commit+moves %.0f" % (st.get("p22",0)/nr, st.get("p23",0)/nr, st.get("p24",0)/nr, st.get("p25",0)/nr))
