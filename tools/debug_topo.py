"""Debug helper: derived vs flattened what-ifs over a topology snapshot (tests/test_whatif_derived.py), first differing what-if in detail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_whatif_derived import _topology_snapshot, _whatif_problem
from karpenter_core_amd import scheduler as S
from oracle import oracle_py as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kinds = set(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[2] != "all" else None
noextras = len(sys.argv) > 3 and sys.argv[3] == "x"
anti = len(sys.argv) > 4
rs = np.random.RandomState(seed)
its, prov, nodes, bound, snap, pod_node = _topology_snapshot(int(rs.randint(24, 120)), int(rs.randint(4, 8)), 50 + seed, spare=int(rs.choice([-1, 0, 3])), extras=(seed != 0 and not noextras), kinds=kinds, anti=anti)
sets = [[int(x) for x in rs.choice(len(nodes), size=int(rs.choice([1, 1, 2, 4, 9])), replace=False)] for _ in range(20)] + [[len(nodes) - 1], [0, len(nodes) - 1]]
parsed = S.ParsedProblem(snap)
derived = S.open_whatifs(parsed, pod_node, sets, derive=True)
flat = S.open_whatifs(parsed, pod_node, sets, derive=False)
got, _, _ = S.solve_batch(derived)
want, _, _ = S.solve_batch(flat)
for i, (a, b) in enumerate(zip(got, want)):
    if a.canonical() == b.canonical() and a.reasons == b.reasons:
        continue
    pr = _whatif_problem(snap, pod_node, sets[i]); ref = O.solve(pr)
    print("what-if", i, sets[i], "flat==oracle", b.canonical() == ref.canonical(), "derived==oracle", a.canonical() == ref.canonical(), "dims", flat[i].dims)
    ca, cb = a.canonical(), b.canonical()
    where_a = {p: n for n, ps in ca["existing"].items() for p in ps}; where_b = {p: n for n, ps in cb["existing"].items() for p in ps}
    for j, nn in enumerate(ca["new_nodes"]):
        for p in nn["pods"]: where_a[p] = f"new{j}"
    for j, nn in enumerate(cb["new_nodes"]):
        for p in nn["pods"]: where_b[p] = f"new{j}"
    for p in sorted(set(where_a) | set(where_b)):
        if where_a.get(p) != where_b.get(p):
            pod = pr.pods[p]
            print("  pod", p, pod.uid, "derived->", where_a.get(p), "flat->", where_b.get(p), "stage", a.final_stage[p], b.final_stage[p], "labels", pod.labels,
                  "spread", [(c.max_skew, c.topology_key, c.when_unsatisfiable) for c in pod.spread], "aff", [(t.topology_key, t.label_selector.match_labels) for t in pod.affinity_required],
                  "affp", [(t.term.topology_key) for t in pod.affinity_preferred], "antip", [(t.term.topology_key) for t in pod.anti_preferred], "ns", pod.node_selector, "ra", len(pod.required_affinity))
    print("  unscheduled", ca["unscheduled"], cb["unscheduled"])
    cs = set(sets[i]); batch = {p.uid for p in pr.pods}
    for nm in sorted(set(list(where_a.values()) + list(where_b.values())))[:4]:
        cnt = {}
        for cp in snap.cluster_pods:
            if cp.node_name == nm and cp.uid not in batch:
                cnt[cp.labels.get("my-label")] = cnt.get(cp.labels.get("my-label"), 0) + 1
        idx = [k for k, n in enumerate(snap.nodes) if n.name == nm]
        print("  node", nm, "index", idx, "zone", snap.nodes[idx[0]].labels.get("topology.kubernetes.io/zone") if idx else None, "staying pods by label", cnt,
              "derived puts", [(q, pr.pods[q].labels["my-label"]) for q in ca["existing"].get(nm, [])], "flat puts", [(q, pr.pods[q].labels["my-label"]) for q in cb["existing"].get(nm, [])])
    break
else:
    print("all equal")
