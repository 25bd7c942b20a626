#!/usr/bin/env python3
"""Golden fingerprints of full-size Solve() results, produced by the CPU oracle (slow: ~2 min for
config #3 at 100k pods).  The fingerprint is sha256 over the canonical result JSON
(model.SolveResult.canonical(): new nodes in creation order with pod lists in commit order, instance
type option lists, request vectors, requirement sets, existing-node pods, unscheduled queue, stages).

    python tests/golden/make_config_hashes.py            # rewrites tests/golden/config_hashes.json
    python tests/golden/make_config_hashes.py NAME...    # refreshes the named entries only
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from karpenter_core_amd import workloads as W  # noqa: E402
from karpenter_core_amd.model import parse_result  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def fingerprint(result) -> str:
    return hashlib.sha256(json.dumps(result.canonical(), sort_keys=True).encode()).hexdigest()


CASES = {
    "config1_1k_50": lambda: W.config1(),
    "config2_10k_500": lambda: W.config2(),
    "config3_100k_2k": lambda: W.config3(),
    "config5_5k_types": lambda: W.config5(pods=CONFIG5_PODS, sizes=50, seed=46),     # 5 000 instance types, full constraint set
    "config5_250k_5k_types": lambda: W.config5(pods=250_000, sizes=50, seed=46),     # 2 h 23 min of the oracle in the build container (1 M needs most of a day)
}
CONFIG5_PODS = 100_000

def config4_entry():
    """BASELINE configs[3] at its stated size: 512 what-ifs over the 2048-node snapshot, one fingerprint per what-if
    (a checksum of checksums pins the whole set)."""
    t = time.time()
    probs = W.config4(512)
    fps, pods, new_nodes, unsched = [], 0, 0, 0
    for pr in probs:
        r = parse_result(O.solve_text(pr.to_ksp()))
        fps.append(fingerprint(r)); pods += len(pr.pods); new_nodes += len(r.new_nodes); unsched += len(r.unscheduled)
    return {"sha256_of_sha256s": hashlib.sha256("".join(fps).encode()).hexdigest(), "whatif_sha256": fps, "whatifs": len(probs), "existing_nodes": 2048,
            "instance_types": len(probs[0].instance_types), "pods": pods, "new_nodes": new_nodes, "unscheduled": unsched, "oracle_seconds": round(time.time() - t, 1)}


def config4b_entry():
    """The replace variant of configs[3] (workloads.config4b_snapshot: a cluster full by pod count): per what-if the fingerprint of the simulation
    AND of the consolidation command the reference's computeConsolidation derives from it (price filter, spot rules; oracle/consolidation_ref.py),
    plus the launch-time pick for the what-ifs that open exactly one node."""
    from karpenter_core_amd import consolidation as C
    from oracle import consolidation_ref as R
    t = time.time()
    its, prov, nodes, bound = W.config4b_snapshot()
    snap = C.Snapshot(its, prov, nodes, bound)
    sets = W.config4_sets(512, 2048, 47)
    fps, cmds, picks, actions, new_nodes, unsched, pods = [], [], [], {}, 0, 0, 0
    for cs in sets:
        sink = []
        cmd = R.canonical(R.compute_consolidation(snap, cs, sink))
        r = sink[0]
        fps.append(fingerprint(r)); cmds.append(hashlib.sha256(json.dumps(cmd, sort_keys=True, default=str).encode()).hexdigest())
        actions[cmd[0]] = actions.get(cmd[0], 0) + 1
        new_nodes += len(r.new_nodes); unsched += len(r.unscheduled); pods += sum(len(bound[i]) for i in cs)
        pk = R.launch_pick(its, r.new_nodes[0]) if len(r.new_nodes) == 1 else None
        picks.append([pk[0], pk[1]] if pk else None)
    return {"whatif_sha256": fps, "command_sha256": cmds, "launch_pick": picks, "actions": actions, "whatifs": len(sets), "existing_nodes": 2048, "instance_types": len(its),
            "pods": pods, "new_nodes": new_nodes, "unscheduled": unsched, "one_new_node": sum(1 for p in picks if p is not None), "oracle_seconds": round(time.time() - t, 1)}


def _config4t_one(args):
    spare, i = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_whatif_derived import _whatif_problem
    snap, pod_node, sets = _config4t_inputs(spare)
    r = parse_result(O.solve_text(_whatif_problem(snap, pod_node, sets[i]).to_ksp()))
    return i, fingerprint(r), len(r.new_nodes), len(r.unscheduled), sum(1 for st in r.final_stage if st)


_C4T = {}


def _config4t_inputs(spare):
    if spare not in _C4T:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_whatif_derived import _topology_snapshot
        its, prov, nodes, bound, snap, pod_node = _topology_snapshot(2048, 50, 45, spare=spare, extras=True, anti=True, zone_anti=False)
        _C4T[spare] = (snap, pod_node, W.config4_sets(512, 2048, 45))
    return _C4T[spare]


def config4t_entry(spare):
    """BASELINE configs[3]'s shape over a cluster whose bound pods carry topology terms (tests/test_whatif_derived.py `_topology_snapshot`: zonal / hostname
    spreads, affinities, preferred terms, required anti-affinity per hostname; cluster-pod records, daemon-like pods, a node no provisioner owns): one
    fingerprint per what-if.  spare = -1: a roomy cluster (the pods move to other nodes); 3: full by pod count (they open nodes)."""
    import multiprocessing as mp
    t = time.time()
    snap, pod_node, sets = _config4t_inputs(spare)
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = sorted(pool.map(_config4t_one, [(spare, i) for i in range(len(sets))], chunksize=1))
    fps = [r[1] for r in rows]
    return {"sha256_of_sha256s": hashlib.sha256("".join(fps).encode()).hexdigest(), "whatif_sha256": fps, "whatifs": len(sets), "nodes": len(snap.nodes), "bound_pods": len(snap.pods),
            "cluster_pod_records": len(snap.cluster_pods), "spare_pod_slots": spare,
            "new_nodes": sum(r[2] for r in rows), "unscheduled": sum(r[3] for r in rows), "relaxed_pods": sum(r[4] for r in rows), "oracle_seconds_8_procs": round(time.time() - t, 1)}


if __name__ == "__main__":
    only = sys.argv[1:]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_hashes.json")
    out = json.load(open(path)) if only and os.path.exists(path) else {}
    if not only or "config4_512x2048" in only:
        out["config4_512x2048"] = config4_entry()
        print("config4_512x2048", {k: v for k, v in out["config4_512x2048"].items() if k != "whatif_sha256"}, flush=True)
    if not only or "config4b_512x2048_replace" in only:
        out["config4b_512x2048_replace"] = config4b_entry()
        print("config4b_512x2048_replace", {k: v for k, v in out["config4b_512x2048_replace"].items() if k not in ("whatif_sha256", "command_sha256", "launch_pick")}, flush=True)
    for name, spare in (("config4t_512x2048_topology", -1), ("config4t_512x2048_topology_replace", 3)):
        if not only or name in only:
            out[name] = config4t_entry(spare)
            print(name, {k: v for k, v in out[name].items() if k != "whatif_sha256"}, flush=True)
    CASES = {k: v for k, v in CASES.items() if not only or k in only}
    for name, mk in CASES.items():
        pr = mk()
        t = time.time()
        r = parse_result(O.solve_text(pr.to_ksp()))
        out[name] = {"sha256": fingerprint(r), "pods": len(pr.pods), "instance_types": len(pr.instance_types),
                     "new_nodes": len(r.new_nodes), "unscheduled": len(r.unscheduled), "attempts": r.stats["attempts"],
                     "types_scanned": r.stats["types_scanned"], "oracle_seconds": round(time.time() - t, 1)}
        print(name, out[name], flush=True)
    if only and os.path.exists(path):      # (another refresh may have finished meanwhile: merge into the file as it is NOW)
        cur = json.load(open(path)); cur.update({k: out[k] for k in only if k in out}); out = cur
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
