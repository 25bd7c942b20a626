#!/usr/bin/env python3
"""Where the head window's cycles go (a -DKS_PROBES -DKS_PROBES_WIN build under karpenter_core_amd/_variants/<KS_VARIANT>): tools/win_profile.py [pods]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from karpenter_core_amd import scheduler as S, workloads as W

pods = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
if os.environ.get("KS_VARIANT"):
    S._HERE = os.path.join(os.path.dirname(S.__file__), "_variants", os.environ["KS_VARIANT"]); S._LIBS = None; S.libs()
pr = W.config3(pods=pods)
fp = S.FlatProblem(pr); fp.upload(0); fp.grid(want_bits=False); fp.solve(decode=False)
r = fp.solve(); st = r.stats
tot = st["kernel_cycles"]
wp, wph = max(st.get("reuse_seeds", 0), 1), max(st.get("reuse_hits", 0), 1)
sw = st.get("reuse_exhausted", 0)
print("kernel_ms %.2f cycles %d nodes %d | rounds %d | runs %d pods %d steps %d | window pods %d phases %d | phases ended: the leader's business %d, nothing in the window accepts %d, exact filter %d" % (fp.kernel_ms, tot, len(r.new_nodes), st.get("eq_pods", 0), st.get("p24", 0), st.get("p22", 0), st.get("p23", 0), wp, wph, sw & 0x1FFFFF, (sw >> 21) & 0x1FFFFF, sw >> 42))
M = 1e6
rows = [("leader: form the batch (rest)", "cyc_evalout"), ("leader: wait at B1", "cyc_full"), ("leader: prepare entries", "cyc_commit"), ("leader: picks / run steps", "cyc_order"), ("leader: resolve", "cyc_new"),
        ("leader: entry lookup (form_one)", "p20"), ("leader: flags (form_one)", "full_checks"), ("leader: cached answers (form_one)", "full_fails"), ("leader: dyn1 answers (form_one)", "queue_pops"),
        ("leader: run records", "p25"), ("leader: after records", "p26"),
        ("window: wait for the lend (B0, B0b)", "cyc_kind0"), ("window: set-up", "cyc_kind1"), ("window: rr_window in all", "cyc_kind2"),
        ("  rr_window: parameters + control word", "n_kind1"), ("  rr_window: dyn1", "n_kind2"), ("  rr_window: evaluate + minimum", "scan_chunks"), ("  rr_window: commit + records", "cyc_pop"), ("  rr_window: preparing records", "cyc_stage")]
acc = 0
for nm, k in rows:
    v = st.get(k, 0)
    if not nm.startswith("  "): acc += v
    print(f"{nm:40s} {v / M:9.2f} M cycles  {100.0 * v / tot:5.1f} %   {v / wp:8.0f} / window pod  {v / wph:8.0f} / phase")
print(f"{'sum':40s} {acc / M:9.2f} M of {tot / M:.2f} M")
if not os.environ.get("KS_WQ"):      # (a -DKS_PROBES_WIN build of round 6: the run statistics' slots hold the loop's waits by cause; part of "parameters + control word")
    print("  the loop waits for: the static part %.2f M | the dyn1 answers %.2f M | answers behind their group's version %.2f M cycles" % (st.get("p22", 0) / M, st.get("p23", 0) / M, st.get("p24", 0) / M))
if os.environ.get("KS_WQ"):      # a -DKS_PROBES_WQ build: the same slots hold what happens AROUND the window's loop
    print("around the loop (the formation rows above are NOT formation in this build):")
    for nm, k in (("rr_window_fast, entry to exit", "cyc_evalout"), ("waiting for the workers' answer", "cyc_full"), ("the hand-over of the answer's node", "cyc_commit"),
                  ("a machine opened + joined", "cyc_order"), ("rr_window_gen", "cyc_new"), ("the rest of the wrapper", "p20")):
        print(f"  {nm:38s} {st.get(k, 0) / M:9.2f} M cycles")
print("run rounds cycles %.1f M, normal rounds (incl. window phases) %.1f M" % (st.get("attempts", 0) / M, st.get("types_scanned", 0) / M))
print("raw:", {k: v for k, v in st.items() if v})
