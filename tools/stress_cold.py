#!/usr/bin/env python3
"""Stress run for the pack kernels (run on the GPU box): N cold processes, each doing the FIRST Solve of its life on config #3 (the 8-wave LEAN
kernel, straight from the pod list), then M what-if batches in one process (single-wave batch kernel over the shared snapshot).  Every result is
hashed and compared with the first one: a GPU memory fault shows up as a child that dies, an uninitialised read as a hash that differs.

usage: tools/stress_cold.py [--cold N] [--batches M] [--pods P] [--poison BYTE]     (KS_POISON is passed to the children when --poison is given)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import ctypes, hashlib, sys
sys.path.insert(0, %r)
from karpenter_core_amd import scheduler as S
ks, kh = S.libs()
text = open(sys.argv[1], 'rb').read()
p = ctypes.c_void_p()
assert kh.ksh_parse(text, len(text), ctypes.byref(p)) == 0
class _P: pass
pp = _P(); pp._p = p
fp, ms = S.solve_from_pods(pp, 0)
out = ctypes.c_void_p(); assert kh.ksh_result_text(fp._h, ctypes.byref(out)) == 0
t = ctypes.string_at(out)
body = b"\n".join(l for l in t.split(b"\n") if not l.startswith(b"STATS") and not l.startswith(b"STAT "))
print(hashlib.sha256(body).hexdigest(), "%%.1f" %% ms["pack_kernel_ms"])
""" % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", type=int, default=100)
    ap.add_argument("--batches", type=int, default=50)
    ap.add_argument("--pods", type=int, default=100_000)
    ap.add_argument("--poison", default=None)
    args = ap.parse_args()
    from karpenter_core_amd import scheduler as S, workloads as W
    env = dict(os.environ)
    if args.poison is not None:
        env["KS_POISON"] = args.poison
    out = {"cold_processes": args.cold, "pods": args.pods, "poison": args.poison, "cold_faults": 0, "cold_hash_mismatches": 0, "batches": args.batches,
           "batch_mismatches": 0}
    if args.cold:
        path = "/tmp/ks_stress_cfg3.ksp"
        with open(path, "w") as f:
            f.write(W.config3(pods=args.pods).to_ksp())
        child = "/tmp/ks_stress_child.py"
        with open(child, "w") as f:
            f.write(CHILD)
        want = None
        t0 = time.time()
        fails = []
        for i in range(args.cold):
            p = subprocess.run([sys.executable, child, path], env=env, capture_output=True, text=True)
            if p.returncode != 0:
                out["cold_faults"] += 1
                fails.append({"run": i, "rc": p.returncode, "stderr": p.stderr[-600:]})
                continue
            h = p.stdout.split()[0]
            if want is None:
                want = h
            elif h != want:
                out["cold_hash_mismatches"] += 1
        out["cold_seconds"] = time.time() - t0
        out["cold_result_sha256"] = want
        out["cold_failures"] = fails[:5]
    if args.batches:
        if args.poison is not None:
            os.environ["KS_POISON"] = args.poison
        its, prov, nodes, bound = W.cluster_snapshot(2048, 50, 45)
        snap, pod_node = W.snapshot_problem(its, prov, nodes, bound, False)
        parsed = S.ParsedProblem(snap)
        sets = W.config4_sets(512, 2048, 45)
        want = None
        t0 = time.time()
        for b in range(args.batches):
            flats = S.open_whatifs(parsed, pod_node, sets)
            S.upload_batch(flats, 0)
            S.solve_batch(flats, decode=False)
            rec = S.result_records(flats, list(range(len(sets))), (len(its) + 63) // 64)
            h = hashlib.sha256(rec.tobytes()).hexdigest()
            for f in flats:
                f.close()
            if want is None:
                want = h
            elif h != want:
                out["batch_mismatches"] += 1
        out["batch_seconds"] = time.time() - t0
        out["batch_records_sha256"] = want
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
