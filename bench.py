#!/usr/bin/env python3
"""bench.py -- pod-placement decisions/sec through Solve() on MI355X (BASELINE.json metric).

N=1 workload: BASELINE configs[2] -- 100k pending pods, 2k instance types, zonal + hostname topology spread and hostname
pod anti-affinity (the configuration the metric "@100k pods" is quoted on; it fits one GPU).

What one timed "step" is (N=1): the whole of the reference's Solve() for a pod list the caller already holds in memory
("queue sort -> last commit"), i.e. `scheduler.solve_from_pods`:
    per-pod RequestsForPods / NewPodRequirements / classing / relaxation chains, NewQueue's sort (queue.go:35),
    the rest of the flattening, upload, static tables + feasibility grid, the pack kernel, result read-back.
`value` = pods / that time.  The pack loop alone on inputs already resident in HBM (round 1's number) is reported next to it
as `resident` -- it is NOT `value`.  decisions = len(pods) per Solve, the reference's own definition
(scheduling_benchmark_test.go:170).

N>1: a single Solve() is a serial dependency chain and does not shard ("replicas only").  What shards is
consolidation's what-if fan-out (BASELINE configs[3]): the 512 what-ifs over one 2048-node snapshot are dealt to the ranks by predicted work (LPT),
every rank solves its shard in ONE batched launch, and ONE RCCL all-gather of fixed-size result records `[id, n_new,
n_unscheduled, first node's InstanceTypeOptions]` closes the step.  Fixed total work: "strong" scaling; `value` = decisions of
all 512 what-ifs / max-over-ranks time; its N=1 reference is the `whatif_batch` object of the N=1 line.

Extra objects on the JSON line:
  roofline     dominant kernel ks_pack: algorithmic bytes of the REFERENCE algorithm for this workload (SURVEY 8d formula with
               the documented R=4, K=8 record sizes; attempts / scanned-type counts from an untimed KS_FLAG_STATS launch) divided
               by the mean launch duration measured with HIP events on the solve stream; `traffic` only from a PMC file recorded
               for the same kernel source; `issue` = the roofline that actually binds (instructions per cycle of one CU).
  cpu_baseline the CPU oracle (a restatement of the Go path; no Go toolchain) on 1 host core on a bounded sample of the same
               generator, plus the recorded full-size oracle time.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
ROOFLINE_R, ROOFLINE_K = 4, 8  # record sizes the SURVEY 8d byte formula is quoted with (fixed: the fraction must not move with the encoding)


def algorithmic_bytes(T, stats, placed):
    """bytes(p) = B_pod + sum_attempted(B_node + |alive| B_it) + B_node(write-back), R=4, K=8."""
    R, K, TW = ROOFLINE_R, ROOFLINE_K, (T + 63) // 64
    b_pod = 8 * R + 16 * K + 16
    b_it = 8 * R + 16 * K + 8
    b_node = 8 * R + 16 * K + 8 + 8 * TW
    return stats["queue_pops"] * b_pod + stats["attempts"] * b_node + stats["types_scanned"] * b_it + placed * b_node


def pmc_leg(leg, kernel_ms):
    """Counter passes of another leg's dominant kernel (tools/profile_bench.sh: `whatifs` = the single-wave batch kernel of the 512 what-ifs, `config5` = the general
    4-wave kernel at 250 000 pods), from the latest profiles/*_bench_pmc.json IF it was recorded with this kernel source: (traffic bytes per launch, issue object, source) or (None, None, None)."""
    try:
        pmc_name = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_pmc.json"))[-1]
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
        if pmc.get("kernel_source_sha16") != kernel_source_sha16():
            return None, None, None
        L = pmc["legs"][leg]
        issue = None
        if L.get("instructions") and kernel_ms:
            clk = pmc.get("ks_pack_per_launch", {}).get("shader_clock_ghz", 2.4); cus = L.get("workgroups", 1)
            issue = {"instructions_per_launch": L["instructions"], "instruction_mix": L.get("instruction_mix"), "workgroups": cus, "waves_per_workgroup": L.get("waves_per_workgroup"),
                     "achieved_ipc_per_workgroup": L["instructions"] / cus / (kernel_ms / 1e3 * clk * 1e9),
                     "wait_fraction": L.get("wait_fraction"),
                     "note": "one workgroup per Solve on its own CU: a CU issues up to 4 instructions per cycle (one per SIMD), a single wave at most 1"}
        return L.get("hbm_bytes_fetch_x2_plus_write"), issue, f"profiles/{pmc_name} legs.{leg} ({L.get('kernel')}; FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, same kernel source)"
    except Exception:
        return None, None, None


def kernel_source_sha16():
    h = hashlib.sha256()
    for f in ("karpenter_core_amd/csrc/ksolve.hip", "karpenter_core_amd/csrc/ks_pack_rr.inc", "karpenter_core_amd/csrc/ks_algebra.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pods", type=int, default=100_000)
    ap.add_argument("--sizes", type=int, default=50, help="ladder sizes; instance types = sizes*40")
    ap.add_argument("--cpu-sample-pods", type=int, default=20_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-only", action="store_true", help="time the CPU oracle on the bounded sample only (skip the full-size run, about two minutes of one host core)")
    ap.add_argument("--whatifs", type=int, default=512, help="consolidation what-ifs (BASELINE configs[3]); 0 skips the N=1 what-if leg")
    ap.add_argument("--config5", type=int, default=0, metavar="PODS", help="BASELINE configs[4] instead of configs[2]: PODS pods (1000000 = the stated size), 5 000 instance types, full "
                    "constraint set, one Solve on one GPU through the general 4-wave kernel; prints the contract line for THAT workload (generation takes minutes)")
    ap.add_argument("--config5-sample", type=int, default=250_000, metavar="PODS", help="also solve BASELINE configs[4]'s generator at PODS pods (250000: the size with an oracle "
                    "fingerprint) and put the object `config5` on the line; 0 skips it (about a minute of generation and a 3 s Solve)")
    ap.add_argument("--whatifs-only", action="store_true", help="diagnostic: only the N=1 what-if leg (prints its object alone, not the contract line)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)      # `python bench.py --gpus N` launches its own N ranks (one per GPU); under torchrun the ranks arrive here with WORLD_SIZE set
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    if os.environ.get("KS_BENCH_DRY"):
        return dry_fanout(args, rank, world)
    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # KS_BENCH_BACKEND=gloo is a rehearsal hook for boxes with fewer GPUs than ranks (the ranks then share devices and the records are
        # gathered through host memory); the driver's runs use the default: one GPU per rank, RCCL.
        backend = os.environ.get("KS_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        elif torch.cuda.device_count() < world:
            sys.exit(f"bench.py: --gpus {world} needs {world} GPUs on this node, {torch.cuda.device_count()} visible (rank {rank} would have no device of its own)")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)       # "nccl" is RCCL on ROCm

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from karpenter_core_amd import scheduler as S, workloads as W

    if world > 1:
        return whatif_fanout(args, rank, world, local_rank, torch, dist, S, W)
    if args.whatifs_only:
        print(json.dumps(whatif_leg(args, 0, 1, local_rank, torch, None, S, W), indent=1))
        return
    if args.config5:
        print(json.dumps(config5_leg(args, local_rank, torch, S, W)))
        return

    # ---- the pod list and the cluster objects in host memory (untimed: the caller holds them) ----
    t0 = time.time()
    problem = W.config3(pods=args.pods, sizes=args.sizes, seed=44)
    parsed = S.ParsedProblem(problem)
    prep_s = time.time() - t0

    def step():
        fp, ms = S.solve_from_pods(parsed, local_rank)
        fp.close()
        return ms

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    rows, lat_ms = [], []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        rows.append(step())
        lat_ms.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    value = args.pods * args.steps / elapsed
    phase = {k: statistics.mean(r[k] for r in rows) for k in S.TIMING_KEYS}

    # ---- inputs resident in HBM: the pack loop alone (round 1's measurement), and the statistics launch for the roofline ----
    fp, _ = S.solve_from_pods(parsed, local_rank)
    dims = fp.dims
    res = fp.result()
    resident = []
    for _ in range(3):
        t1 = time.perf_counter()
        fp.solve(decode=False)
        resident.append(((time.perf_counter() - t1) * 1e3, fp.kernel_ms))
    _, grid_ms = fp.grid(want_bits=False)
    # which pack kernel took the Solve (ks_pack_rr, round 4, takes single LEAN Solves it covers; ks_pack everything else), and the OTHER one on the same
    # resident problem beside it (KS_NO_RR=1): not `value`
    took_rr = bool(res.stats.get("eq_pods"))
    alt = None
    try:
        os.environ["KS_NO_RR"] = "1"
        fp.solve(decode=False)
        ks = []
        for _ in range(3):
            fp.solve(decode=False)
            ks.append(fp.kernel_ms)
        ra = fp.solve()
        alt = {"kernel": "ks_pack<FAST, LEAN, 8 waves> (KS_NO_RR=1): speculation rounds over a 64-candidate window -- the kernel of rounds 1-3",
               "kernel_ms": statistics.median(ks), "decisions_per_s_kernel": dims["P"] / (statistics.median(ks) / 1e3),
               "same_result_as_the_default_kernel": ra.canonical() == res.canonical()}
    finally:
        os.environ.pop("KS_NO_RR", None)
    pack_name = "ks_pack_rr" if took_rr else "ks_pack"
    ws = res.stats.get("cyc_kind2", 0)
    rr_stats = {"rounds": res.stats.get("eq_pods"), "pods_placed_in_runs": res.stats.get("p22"), "run_steps": res.stats.get("p23"), "runs": res.stats.get("p24"),
                "pods_placed_by_the_head_window": res.stats.get("cyc_kind0"), "window_phases": res.stats.get("cyc_kind1"),
                "window_phases_ended_because": {"the_leaders_business": ws & 0x1FFFFF, "nothing_in_the_window_accepts": (ws >> 21) & 0x1FFFFF, "exact_filter": ws >> 42}} if took_rr else None
    # what a caller pays to read the result as KSR1 text (Node.Pods, InstanceTypeOptions, Requirements, Requests for every node): outside `value`, reported
    t1 = time.perf_counter(); fp.result(); egress_text_ms = (time.perf_counter() - t1) * 1e3
    # ... and as arrays (ksh_result_arrays_get: pod -> node, Node.Pods as a CSR in commit order, type masks, requests, requirement records; numpy copies included)
    fp.result_arrays(); t1 = time.perf_counter(); fp.result_arrays(); egress_arrays_ms = (time.perf_counter() - t1) * 1e3
    fp.close()
    fps, _ = S.solve_from_pods(parsed, local_rank, stats=True)
    st = fps.result().stats
    fps.close()
    TW = (dims["T"] + 63) // 64
    abytes = algorithmic_bytes(dims["T"], st, dims["P"])
    mean_kernel_s = phase["pack_kernel_ms"] / 1e3
    achieved = abytes / mean_kernel_s / 1e9
    grid_bytes = dims["C"] * (8 * dims["R"] + 16 * dims["K"] + 16) + dims["T"] * (8 * dims["R"] + 16 * dims["K"] + 8) + dims["M"] * dims["C"] * TW * 8
    traffic, traffic_src, issue = None, None, None
    try:
        pmc_name = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_pmc.json"))[-1]      # the latest round's counter passes
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
        if pmc.get("kernel_source_sha16") == kernel_source_sha16() and pmc.get("pods") == dims["P"] and pmc.get("instance_types") == dims["T"]:
            traffic = pmc["ks_pack_per_launch"]["hbm_bytes_fetch_x2_plus_write"]
            traffic_src = f"profiles/{pmc_name} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, same kernel source)"
            ipl = pmc["ks_pack_per_launch"].get("instructions")
            if ipl:
                clk = pmc["ks_pack_per_launch"].get("shader_clock_ghz", 2.4)
                issue = {"instructions_per_launch": ipl, "instructions_per_pod": ipl / dims["P"], "cu_issue_capacity_per_cycle": 4,
                         "achieved_ipc_one_cu": ipl / (mean_kernel_s * clk * 1e9), "frac": ipl / (mean_kernel_s * clk * 1e9) / 4.0,
                         "note": "ks_pack runs ONE workgroup on ONE CU: issue capacity is 4 instructions/cycle/CU (one per SIMD)"}
    except Exception:
        pass
    out = {
        "metric": "pod-placement decisions/sec (Solve())", "value": value, "unit": "decisions/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "n/a", "scaling_note": "N=1: one Solve on one GPU (a single Solve is replicas-only, SURVEY 8e): nothing scales; the N>1 line (the what-if fan-out) says strong / weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: {dims['P']} pods, {dims['T']} instance types, zonal+hostname topology spread and "
                               f"hostname pod anti-affinity (workloads.config3 seed 44)", "pods": dims["P"], "instance_types": dims["T"],
                   "pod_classes": dims["C"], "topology_groups": dims["G"], "new_nodes": len(res.new_nodes),
                   "unschedulable": len(res.unscheduled), "parallelism": "1 Solve on 1 GPU",
                   "timed_window": "Solve() from the pod list in host memory: per-pod requests/requirements/classes, relaxation chains, topology groups, NewQueue sort, "
                                   "flattening of the batch, upload, static tables + grid, pack kernel, read-back (scheduler.solve_from_pods).  The environment's own "
                                   "flattening (instance types, templates, state nodes) is cached with the caller's objects per universe signature and adopted by the timed "
                                   "steps (the warm-up step builds it; KSH_NO_ENV_CACHE=1 re-does it per Solve: +11 ms); the reference's benchmark leaves NewScheduler's "
                                   "whole assembly out of its timer (scheduling_benchmark_test.go:130)"},
        "p50_solve_latency_ms": statistics.median(lat_ms),
        "phases_ms_mean": phase, "host_threads": os.cpu_count(), "prep_seconds_untimed": prep_s,
        "resident": {"what": "pack loop only, flattened problem already resident in HBM (ks_solve_dev incl. read-back) -- round 1's window",
                     "decisions_per_s": dims["P"] / (statistics.median(r[0] for r in resident) / 1e3),
                     "wall_ms": statistics.median(r[0] for r in resident), "kernel_ms": statistics.median(r[1] for r in resident)},
        "roofline": {"kernel": pack_name, "kernel_stats": rr_stats, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": abytes, "formula": f"SURVEY 8d with R={ROOFLINE_R}, K={ROOFLINE_K}",
                     "kernel_ms_mean": phase["pack_kernel_ms"], "ref_attempts": st["attempts"], "ref_types_scanned": st["types_scanned"],
                     "issue": issue,
                     "note": "one Solve() is a serial dependency chain executed by ONE 8-wave workgroup (1 of 256 CUs): instruction issue and "
                             "dependent-access latency bind it, the HBM fraction is reported because the contract asks for it (DESIGN.md)"},
        "alt_kernel": alt,
        "egress": {"what": "ksh_result_text + its parse in the Python mirror for the whole result (every pod, every node): the boundary's output side, NOT inside `value` (the timed "
                           "window ends with the binary result on the host)", "ms": egress_text_ms,
                   "arrays_ms": egress_arrays_ms, "arrays_what": "ksh_result_arrays_get + numpy copies of every array: the binary door for the same result (no text)"},
        "grid": {"kernel": "ks_grid_mc+ks_grid_types", "ms": grid_ms, "algorithmic_bytes": grid_bytes,
                 "achieved_GBs": grid_bytes / (grid_ms / 1e3) / 1e9 if grid_ms else None},
    }
    out["ingress"] = ingress_leg(args, problem, local_rank, S)
    out["value_with_ingress"] = out["ingress"]["decisions_per_s_with_ingress"]
    # door to door, no text: pods in as binary blocks, Solve, the result out as arrays (the environment resident, as it is between catalogue changes)
    out["value_end_to_end"] = len(problem.pods) / ((out["ingress"]["end_to_end_ms"] + egress_arrays_ms) / 1e3)
    if args.whatifs:
        out["whatif_batch"] = whatif_leg(args, 0, 1, local_rank, torch, None, S, W)
    if args.config5_sample:
        try:
            sub = argparse.Namespace(**{**vars(args), "config5": args.config5_sample, "steps": 1, "warmup": 1})
            c5 = config5_leg(sub, local_rank, torch, S, W)
            out["config5"] = {"what": "BASELINE configs[4]'s generator at the size the oracle's offline fingerprint exists for (the stated 1 M pods: `bench.py --config5 1000000`, "
                                      "profiles/); one Solve from the pod list, not `value`", "pods": c5["config"]["pods"], "instance_types": c5["config"]["instance_types"],
                              "decisions_per_s": c5["value"], "ms_per_solve": c5["ms_per_step"], "new_nodes": c5["config"]["new_nodes"], "unschedulable": c5["config"]["unschedulable"],
                              "oracle_fingerprint": c5["oracle_fingerprint"], "phases_ms_mean": c5["phases_ms_mean"], "roofline": c5["roofline"], "prep_seconds_untimed": c5["prep_seconds_untimed"]}
        except Exception as e:      # (the object is a diagnostic: the contract line must not die with it)
            out["config5"] = {"error": str(e)[:300]}
    if not args.no_cpu_baseline:
        from oracle import oracle_py
        sample = W.config3(pods=args.cpu_sample_pods, sizes=args.sizes, seed=44)
        from karpenter_core_amd.model import parse_result
        r = parse_result(oracle_py.solve_text(sample.to_ksp()))
        secs = r.stats["solve_ns"] / 1e9
        full = None
        try:
            full = json.load(open(os.path.join(ROOT, "tests", "golden", "config_hashes.json"))).get("config3_100k_2k", {}).get("oracle_seconds")
        except Exception:
            pass
        sample_obj = {"pods": args.cpu_sample_pods, "seconds": secs, "decisions_per_s": args.cpu_sample_pods / secs,
                      "what": f"same generator at {args.cpu_sample_pods} pods / {dims['T']} instance types; the oracle's cost grows super-linearly with pods"}
        here = None
        if not args.cpu_sample_only and args.pods == 100_000 and args.sizes == 50:
            # the headline configuration itself, on this box's host cores beside the GPU (one core, like the Go path's single goroutine): ~2 minutes
            t1 = time.perf_counter()
            rf = parse_result(oracle_py.solve_text(problem.to_ksp()))
            wall = time.perf_counter() - t1
            here = {"pods": args.pods, "oracle_seconds": rf.stats["solve_ns"] / 1e9, "wall_seconds_incl_parse": wall,
                    "same_result_as_the_gpu": rf.canonical() == res.canonical()}
        if here:
            out["cpu_baseline"] = {"value": args.pods / here["oracle_seconds"], "unit": "decisions/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
                                   "seconds": here["oracle_seconds"], "same_result_as_the_gpu": here["same_result_as_the_gpu"],
                                   "sample": f"the headline workload itself ({args.pods} pods / {dims['T']} instance types), Solve() incl. the queue sort, one thread, timed on this box",
                                   "bounded_sample": sample_obj}
        else:
            out["cpu_baseline"] = {"value": args.cpu_sample_pods / secs, "unit": "decisions/s", "cores": 1, "kind": "port",
                                   "host_cores": os.cpu_count(), "seconds": secs,
                                   "sample": f"same generator at {args.cpu_sample_pods} pods / {dims['T']} instance types (Solve() incl. the queue sort, single "
                                             "thread like the Go path); the oracle's cost grows super-linearly with pods",
                                   "full_size_recorded": {"pods": 100000, "oracle_seconds": full, "decisions_per_s": (100000 / full) if full else None,
                                                          "source": "tests/golden/config_hashes.json (tests/golden/make_config_hashes.py, build container)"}}
    print(json.dumps(out))


def config5_leg(args, device, torch, S, W):
    """BASELINE configs[4]: 1 M pods / 5 000 instance types / the full constraint set (taints, Gt selectors on an integer label, zonal + hostname +
    capacity-type spread, pod affinity and anti-affinity, host ports, two weighted provisioners, one with a cpu limit).  A single Solve does not
    shard, so this is one GPU; host ports and limits route it through the general (non-LEAN, BOUNDS) 4-wave kernel."""
    t0 = time.time()
    problem = W.config5(pods=args.config5, sizes=50, seed=46)
    parsed = S.ParsedProblem(problem)
    prep_s = time.time() - t0
    steps = max(1, min(args.steps, 3))
    for _ in range(min(args.warmup, 1)):
        fp, _ = S.solve_from_pods(parsed, device); fp.close()
    torch.cuda.synchronize()
    rows, lat = [], []
    t_start = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        fp, ms = S.solve_from_pods(parsed, device)
        lat.append((time.perf_counter() - t1) * 1e3); rows.append(ms)
        if len(rows) < steps:
            fp.close()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    dims, res = fp.dims, fp.result()
    fp.close()
    fps, _ = S.solve_from_pods(parsed, device, stats=True)
    st = fps.result().stats
    fps.close()
    abytes = algorithmic_bytes(dims["T"], st, dims["P"])
    kms = statistics.mean(r["pack_kernel_ms"] for r in rows)
    gold = None
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "config_hashes.json")))
        for name in ("config5_250k_5k_types", "config5_5k_types"):
            if g.get(name, {}).get("pods") == dims["P"]:
                gold = {"entry": name, "matches_oracle_fingerprint": hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest() == g[name]["sha256"]}
    except Exception:
        pass
    return {"metric": "pod-placement decisions/sec (Solve())", "value": dims["P"] * steps / elapsed, "unit": "decisions/s", "n_gpus": 1, "steps": steps, "warmup": min(args.warmup, 1),
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: {dims['P']} pods, {dims['T']} instance types, full constraint set (workloads.config5 seed 46)", "pods": dims["P"],
                       "instance_types": dims["T"], "pod_classes": dims["C"], "topology_groups": dims["G"], "new_nodes": len(res.new_nodes), "unschedulable": len(res.unscheduled),
                       "parallelism": "1 Solve on 1 GPU (a single Solve is replicas-only)", "timed_window": "Solve() from the pod list in host memory (scheduler.solve_from_pods)"},
            "p50_solve_latency_ms": statistics.median(lat), "phases_ms_mean": {k: statistics.mean(r[k] for r in rows) for k in S.TIMING_KEYS}, "prep_seconds_untimed": prep_s,
            "oracle_fingerprint": gold,
            "roofline": {"kernel": "ks_pack<general, 4 waves>", "bound": "hbm", "achieved": abytes / (kms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": abytes / (kms / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": (pmc_leg("config5", kms)[0] if dims["P"] == 250_000 else None), "traffic_source": (pmc_leg("config5", kms)[2] if dims["P"] == 250_000 else None),
                         "issue": (pmc_leg("config5", kms)[1] if dims["P"] == 250_000 else None), "algorithmic_bytes_per_launch": abytes, "kernel_ms_mean": kms,
                         "ref_attempts": st["attempts"], "ref_types_scanned": st["types_scanned"], "formula": f"SURVEY 8d with R={ROOFLINE_R}, K={ROOFLINE_K}"}}


def ingress_leg(args, problem, device, S):
    """How the caller's pods get INTO the library, and what Solve() costs when that is counted: the binary door (`ksh_pods_ingest`: flat u32
    records + string tables, what a cgo shim fills from its []*v1.Pod) against the KSP1 text door (`ksh_parse`).  Filling the blocks / printing the
    text is the caller's side and is not timed here (done in Python)."""
    import dataclasses
    from karpenter_core_amd.model import pods_to_blocks
    blocks = pods_to_blocks(problem.pods, 4)
    env = S.ParsedProblem(dataclasses.replace(problem, pods=[]))
    S.PodBatch(blocks).close()                                  # warm-up (worker threads, allocator)
    ing, tot, rows = [], [], []
    for _ in range(max(3, args.steps)):
        t1 = time.perf_counter()
        b = S.PodBatch(blocks)
        fp, ms = S.solve_from_batch(env, b, device)
        tot.append((time.perf_counter() - t1) * 1e3)
        ing.append(b.ingest_ms)
        rows.append(ms)
        fp.close()
        b.close()
    text = problem.to_ksp().encode()
    t1 = time.perf_counter()
    pp = S.ParsedProblem.from_text(text)
    parse_ms = (time.perf_counter() - t1) * 1e3
    pp.close()
    med = statistics.median
    # the ENVIRONMENT's own doors (instance types + offerings, provisioners, state nodes, cluster pods, daemonsets): binary (ksh_env_ingest) against its KSP1 text; and the whole
    # call with NOTHING held by the library beforehand -- environment in, pods in, Solve -- which is what a first call after a catalogue change costs
    from karpenter_core_amd.model import env_to_block
    eblk = env_to_block(problem)
    etext = dataclasses.replace(problem, pods=[]).to_ksp().encode()
    env_bin_ms, env_text_ms, cold = [], [], []
    for _ in range(3):
        e2 = S.ParsedProblem.from_env_block(eblk); env_bin_ms.append(e2.ingest_ms); e2.close()
        t1 = time.perf_counter(); e3 = S.ParsedProblem.from_text(etext); env_text_ms.append((time.perf_counter() - t1) * 1e3); e3.close()
        t1 = time.perf_counter()
        e4 = S.ParsedProblem.from_env_block(eblk); b = S.PodBatch(blocks); fp, _ = S.solve_from_batch(e4, b, device)
        cold.append((time.perf_counter() - t1) * 1e3)
        fp.close(); b.close(); e4.close()
    return {"what": "ksh_pods_ingest (binary pod blocks, 4 blocks) + ksh_solve_from_batch: Solve() counted from the moment the caller hands its pods over",
            "ingress_ms": med(ing), "solve_from_batch_ms": med(r["total_ms"] for r in rows), "flatten_ms": med(r["flatten_ms"] for r in rows),
            "end_to_end_ms": med(tot), "decisions_per_s_with_ingress": len(problem.pods) / (med(tot) / 1e3),
            "block_bytes": int(sum(sum(v.nbytes for v in b.values() if hasattr(v, "nbytes")) for b in blocks)),
            "ksp1_text_door": {"bytes": len(text), "ksh_parse_ms": parse_ms, "note": "whole problem incl. the catalogue; the door round 2 had"},
            "environment": {"what": "everything but the pods (2000 instance types with their offerings, the provisioner): ksh_env_ingest (one stream of u32 words + a string table) "
                                    "against ksh_parse of its KSP1 text; normally paid once per catalogue change -- the environment's flattening is cached across batches",
                            "env_ingest_ms": med(env_bin_ms), "env_block_bytes": int(sum(v.nbytes for v in eblk.values() if hasattr(v, "nbytes"))),
                            "env_text_parse_ms": med(env_text_ms), "env_text_bytes": len(etext),
                            "end_to_end_nothing_cached_ms": med(cold), "decisions_per_s_nothing_cached": len(problem.pods) / (med(cold) / 1e3)}}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1."""
    import socket
    import subprocess
    if not os.environ.get("KS_BENCH_DRY") and os.environ.get("KS_BENCH_BACKEND", "nccl") == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs on this node, {have} visible")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def dry_fanout(args, rank, world):
    """KS_BENCH_DRY=1 (CPU rehearsal of the launch + exchange plumbing, tests/test_distributed.py): the ranks deal the what-if ids out exactly as the
    real leg does and all-gather id-only records over gloo; nothing is solved and the line says so (`dry`: it is not a measurement)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    from karpenter_core_amd import consolidation as C
    total = args.whatifs or 512
    from karpenter_core_amd import workloads as W
    dealt = C.deal([len(cs) for cs in W.config4_sets(total, 2048, 45)], world)      # (the real leg weighs a what-if by its pods; the rehearsal by its candidate nodes)
    mine = dealt[rank]
    per = max(len(d) for d in dealt)
    rec = torch.zeros((len(mine), 3), dtype=torch.int64)
    rec[:, 0] = torch.tensor(mine, dtype=torch.int64)
    table = C.all_gather_records(rec, per)
    if rank == 0:
        print(json.dumps({"metric": "pod-placement decisions/sec (Solve())", "value": None, "dry": True, "n_gpus": world,
                          "config": {"whatifs": total, "records_gathered": int(table.shape[0]), "ids_in_order": bool((table[:, 0] == torch.arange(total)).all())}}))
    dist.destroy_process_group()


def whatif_snapshot(args, S, W):
    """BASELINE configs[3]'s cluster: 2048 existing nodes with their bound pods, held as objects in host memory (untimed, like the pod list)."""
    c_its, c_prov, c_nodes, c_bound = W.cluster_snapshot(2048, args.sizes, 45)
    c_snap, c_pn = W.snapshot_problem(c_its, c_prov, c_nodes, c_bound, False)
    whatif_snapshot.its = c_its
    return S.ParsedProblem(c_snap), c_pn, len(c_its), W.config4_sets(args.whatifs or 512, 2048, 45)


def whatif_leg(args, rank, world, local_rank, torch, dist, S, W):
    """BASELINE configs[3] on one GPU, end to end: flatten the what-ifs over the shared snapshot, upload, ONE batched launch, binary result records."""
    parsed, pod_node, T, sets = whatif_snapshot(args, S, W)
    mine = list(range(rank, len(sets), world))
    words = (T + 63) // 64

    rec_e2e = torch.full((len(mine), 3 + words), -1, dtype=torch.int64, device=f"cuda:{local_rank}")

    def end_to_end(pod_node=pod_node):
        """What a consolidation pass pays per batch of candidate sets over a snapshot it already holds (multinodeconsolidation.go:74-114): open the
        what-ifs (derived on the device from the resident snapshot: candidate masks up, batches built there), one batched launch, the fixed-size
        decision records built on the device and brought to the host."""
        t0 = time.perf_counter()
        flats = S.open_whatifs(parsed, pod_node, [sets[i] for i in mine], device=local_rank)
        t1 = time.perf_counter()
        S.upload_batch(flats, local_rank)                      # (derived what-ifs are resident already: a no-op; the host-flattened fallback uploads here)
        t2 = time.perf_counter()
        kms, _ = S.solve_batch_resident(flats)
        S.result_records_dev(flats, mine, words, rec_e2e)
        rec = rec_e2e.cpu().numpy()
        t3 = time.perf_counter()
        return flats, rec, {"open_ms": (t1 - t0) * 1e3, "upload_ms": (t2 - t1) * 1e3, "solve_records_ms": (t3 - t2) * 1e3, "kernel_ms": kms, "total_ms": (t3 - t0) * 1e3}

    flats, rec, first = end_to_end()       # first batch over this snapshot: also flattens the snapshot itself (once per snapshot, cached in the
                                           # parsed object), puts that flattening on the device (shared catalogue + derived tables), warms the pools
    for f in flats:
        f.close()
    runs = []
    for _ in range(3):
        flats, rec, ms = end_to_end()
        runs.append(ms)
        if len(runs) < 3:
            for f in flats:
                f.close()
    ms = sorted(runs, key=lambda r: r["total_ms"])[1]
    # resident: the batch already in HBM -- exactly the step of the N>1 fan-out at world size 1: one batched launch with the results left on the
    # device, the fixed-size records built there (no host hop), nothing else
    rec_dev = torch.full((len(flats), 3 + words), -1, dtype=torch.int64, device=f"cuda:{local_rank}")
    res = []
    for _ in range(5):
        t1 = time.perf_counter()
        kms, _ = S.solve_batch_resident(flats)
        S.result_records_dev(flats, mine, words, rec_dev)
        res.append(((time.perf_counter() - t1) * 1e3, kms))
    wms, kms = sorted(res)[2]
    S.solve_batch(flats, decode=False)                       # the same batch read back in full (every pod's node): the host-built records must agree
    assert (S.result_records(flats, mine, words) == rec).all() and (rec_dev.cpu().numpy() == rec).all(), "device-built records differ from the host-built ones"
    pods_mine = sum(f.dims["P"] for f in flats)
    # roofline of the batch kernel: the REFERENCE algorithm's bytes for these what-ifs (untimed KS_FLAG_STATS batch) over the launch's HIP-event time
    sflats = S.open_whatifs(parsed, pod_node, [sets[i] for i in mine], stats=True)
    S.upload_batch(sflats, local_rank)
    sres, _, _ = S.solve_batch(sflats)
    abytes = sum(algorithmic_bytes(T, r.stats, f.dims["P"]) for r, f in zip(sres, sflats))
    for f in sflats:
        f.close()
    out = {"workload": f"{len(flats)} consolidation what-ifs over 2048 existing nodes / {T} instance types (BASELINE configs[3])",
           "whatifs": len(flats), "decisions": pods_mine, "records": int(rec.shape[0]),
           "first_batch_over_the_snapshot": dict(first, what="cold: + the snapshot's own flattening and its upload (once per snapshot), buffer pools, code objects"),
           "end_to_end": dict(ms, what="a batch of candidate sets over a snapshot already seen (a consolidation pass probes many): what-ifs derived on the device from the resident snapshot + one batched launch + decision records to the host",
                              decisions_per_s=pods_mine / (ms["total_ms"] / 1e3), whatifs_per_s=len(flats) / (ms["total_ms"] / 1e3)),
           "resident": {"what": "the N>1 fan-out's step at world size 1: batched launch, results left on the device, records built there",
                        "kernel_ms": kms, "wall_ms": wms, "decisions_per_s_kernel": pods_mine / (kms / 1e3), "decisions_per_s_wall": pods_mine / (wms / 1e3),
                        "whatifs_per_s_wall": len(flats) / (wms / 1e3)},
           "roofline": {"kernel": "ks_pack<single wave> x what-ifs in one launch", "bound": "hbm", "achieved": abytes / (kms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": abytes / (kms / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_leg("whatifs", kms)[0], "traffic_source": pmc_leg("whatifs", kms)[2], "issue": pmc_leg("whatifs", kms)[1],
                        "algorithmic_bytes_per_launch": abytes, "kernel_ms": kms,
                        "formula": f"SURVEY 8d with R={ROOFLINE_R}, K={ROOFLINE_K}, summed over the what-ifs",
                        "bound_as_it_is": "the batch is its LONGEST what-if: one wave placing its pods one after the other (a chain of dependent instructions at about one per ten cycles) while the "
                                          "other 511 workgroups have long finished; `achieved` / `frac` price the REFERENCE algorithm's bytes -- the attempts the watermark and the run commit skip are in "
                                          "them -- against the kernel's time, `traffic` is what the counters saw move",
                        "note": "a batch takes as long as its longest what-if (one wave each); the watermark / run commit skip most of the attempts the reference makes"}}
    for f in flats:
        f.close()
    # ---- the cluster changed by ONE node (a machine joined, 20 pods were bound to it) -- the events go to the snapshot the library holds
    # (ksh_env_apply) and the next batch of what-ifs is opened over it, against ingesting + flattening the cluster again (`first_batch_over_the_snapshot`) ----
    import numpy as np
    rs = np.random.RandomState(7)
    incr = []
    for r in range(3):
        name = f"joined-{r}"
        events = [("node+", W.fresh_node(whatif_snapshot.its, name, rs))] + [("bind", name, W.generic_pod(rs, f"joined-{r}-{k}")) for k in range(20)]
        info = parsed.apply(events, pod_node if r == 0 else None)
        flats2, rec2, ms2 = end_to_end(None)
        for f in flats2:
            f.close()
        incr.append({"apply_ms": info["ms"], "continued": info["continued"], **ms2, "total_with_apply_ms": info["ms"] + ms2["total_ms"]})
    out["after_a_one_node_change"] = dict(sorted(incr, key=lambda r: r["total_with_apply_ms"])[1],
                                          what="SURVEY 8f-1: one node joined and 20 pods were bound to it -- the events patched the resident snapshot (ksh_env_apply: objects in place, the "
                                               "flattening continued from the one before), then the same 512 candidate sets were opened, solved and their records read; "
                                               "`first_batch_over_the_snapshot` is what ingesting the changed cluster again would cost on top of the parse",
                                          events_per_change=21, runs=incr)
    return out


def whatif_fanout(args, rank, world, local_rank, torch, dist, S, W):
    """N>1: the 512 what-ifs dealt to the ranks by predicted work (strong scaling), one batched launch per rank, ONE all-gather of result records; the same leg under weak
    scaling (every rank all 512) beside it."""
    parsed, pod_node, T, sets = whatif_snapshot(args, S, W)
    total_whatifs = len(sets)
    from karpenter_core_amd import consolidation as C
    # dealt by predicted work -- the pods of a what-if's candidate nodes --, longest first to the least loaded rank (round 5; i mod N before): the step is no longer its
    # longest what-if plus whatever happened to sit beside it
    pods_on = {}
    for nd in pod_node:
        pods_on[nd] = pods_on.get(nd, 0) + 1
    weights = [sum(pods_on.get(c, 0) for c in cs) for cs in sets]
    dealt = C.deal(weights, world)
    mine = dealt[rank]
    flats = S.open_whatifs(parsed, pod_node, [sets[i] for i in mine], device=local_rank)
    S.upload_batch(flats, local_rank)
    words = (T + 63) // 64
    per = max(len(d) for d in dealt)
    pods_mine = sum(f.dims["P"] for f in flats)

    on_device = dist.get_backend() == "nccl"
    dev = f"cuda:{local_rank}"
    width = 3 + words
    local = torch.full((per, width), -1, dtype=torch.int64, device=dev)               # rows [0, len(mine)) are rewritten by every step; padding rows keep id -1
    gathered = torch.empty((world * per, width), dtype=torch.int64, device=dev)

    def step():
        S.solve_batch_resident(flats)                                                  # one batched launch; the results stay on the device
        S.result_records_dev(flats, mine, words, local)                               # [id, n_new, n_unscheduled, options] built on the device, complete on return
        if on_device:
            dist.all_gather_into_tensor(gathered, local)                              # the single collective of the path: RCCL all-gather of the device buffer as is
            return gathered
        parts = [torch.empty_like(local) for _ in range(world)]                       # (gloo rehearsal on a box with fewer GPUs than ranks)
        dist.all_gather(parts, local)
        return torch.cat(parts, 0)

    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table = step()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    red_dev = "cuda" if on_device else "cpu"
    tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    tp = torch.tensor([pods_mine], device=red_dev, dtype=torch.int64)
    dist.all_reduce(tp)
    total_pods = int(tp.item())
    table = table[table[:, 0] >= 0]
    table = table[torch.argsort(table[:, 0])]
    # kernel time of the batched launch (HIP events on the solve stream inside ks_solve_batch_dev), mean over a few launches
    kms = []
    for _ in range(3):
        k, _ = S.solve_batch_resident(flats)
        kms.append(k)
    # the same total work on ONE GPU of this box (rank 0 alone, the others wait): the strong-scaling reference measured beside the N-rank number
    n1 = None
    if rank == 0:
        allf = S.open_whatifs(parsed, pod_node, sets, device=local_rank)
        S.upload_batch(allf, local_rank)
        one = torch.full((total_whatifs, width), -1, dtype=torch.int64, device=dev)
        ids = list(range(total_whatifs))
        for _ in range(max(1, args.warmup)):
            S.solve_batch_resident(allf); S.result_records_dev(allf, ids, words, one)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            S.solve_batch_resident(allf); S.result_records_dev(allf, ids, words, one)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        n1 = {"what": "all what-ifs on rank 0's GPU alone, same step without the collective", "ms_per_step": e1 / args.steps * 1e3,
              "decisions_per_s": sum(f.dims["P"] for f in allf) * args.steps / e1, "records_equal_gathered": bool((one == table).all().item())}
        for f in allf:
            f.close()
    # ---- the same leg under WEAK scaling: every rank solves ALL the what-ifs over its own copy of the snapshot (N times the work on N GPUs), one all-gather of all records ----
    weak = None
    try:
        allw = S.open_whatifs(parsed, pod_node, sets, device=local_rank)
        S.upload_batch(allw, local_rank)
        idsw = [rank * total_whatifs + i for i in range(total_whatifs)]
        localw = torch.full((total_whatifs, width), -1, dtype=torch.int64, device=dev)
        gatheredw = torch.empty((world * total_whatifs, width), dtype=torch.int64, device=dev)

        def stepw():
            S.solve_batch_resident(allw); S.result_records_dev(allw, idsw, words, localw)
            if on_device:
                dist.all_gather_into_tensor(gatheredw, localw)
            else:
                parts = [torch.empty_like(localw) for _ in range(world)]
                dist.all_gather(parts, localw)
        for _ in range(max(1, args.warmup)):
            stepw()
        dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            stepw()
        torch.cuda.synchronize(); dist.barrier()
        tw = torch.tensor([time.perf_counter() - t1], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        pods_all = sum(f.dims["P"] for f in allw)
        weak = {"what": f"every rank solves all {total_whatifs} what-ifs over its own copy of the snapshot ({world}x the work), one all-gather of {world * total_whatifs} records",
                "scaling": "weak", "whatifs_per_gpu": total_whatifs, "ms_per_step": float(tw.item()) / args.steps * 1e3,
                "decisions_per_s": world * pods_all * args.steps / float(tw.item())}
        for f in allw:
            f.close()
    except Exception as e:      # (diagnostic object: the contract line must not die with it)
        weak = {"error": str(e)[:300]}
    # ---- SURVEY 8e row 2 beside it: the static feasibility grid of ONE Solve (BASELINE configs[1]: 10 000 pods / 500 types) built by all ranks -- each its share of the rows on
    # its GPU, ONE all-gather of the bit-rows, every rank installs the others' -- and the Solve over the gathered grid against the same Solve with the grid built on one GPU ----
    grid_sh = None
    try:
        import hashlib
        p2 = W.config2()
        fg = S.FlatProblem(p2); fg.upload(local_rank)
        gdev = "cuda" if on_device else "cpu"
        t1 = time.perf_counter()
        gms = S.sharded_grid(fg, rank, world, lambda rows: C.all_gather_grid_rows(rows, gdev))
        torch.cuda.synchronize(); t_sh = (time.perf_counter() - t1) * 1e3
        h_sh = int(hashlib.sha256(json.dumps(fg.solve().canonical(), sort_keys=True).encode()).hexdigest()[:15], 16)
        fg.close()
        one_g = S.FlatProblem(p2); one_g.upload(local_rank)
        _, gms1 = one_g.grid(want_bits=False)
        h_one = int(hashlib.sha256(json.dumps(one_g.solve().canonical(), sort_keys=True).encode()).hexdigest()[:15], 16)
        one_g.close()
        hv = torch.tensor([h_sh, h_one], device=red_dev, dtype=torch.int64)
        hs = [torch.empty_like(hv) for _ in range(world)]
        dist.all_gather(hs, hv)
        grid_sh = {"what": "SURVEY 8e row 2: the feasibility grid's rows (template x class pairs) of one Solve split over the ranks, one all-gather of bit-rows (scheduler.sharded_grid)",
                   "workload": "BASELINE configs[1]: 10000 pods / 500 instance types", "rows": fg.dims["M"] * fg.dims["C"], "row_words": (fg.dims["T"] + 63) // 64,
                   "this_ranks_rows_kernel_ms": gms, "sharded_build_wall_ms": t_sh, "whole_grid_on_one_gpu_kernel_ms": gms1,
                   "solve_equal_on_every_rank_and_to_the_single_gpu_grid": bool(all(int(x[0]) == h_one and int(x[1]) == h_one for x in hs))}
    except Exception as e:      # (diagnostic object: the contract line must not die with it)
        grid_sh = {"error": str(e)[:300]}
    dist.barrier()
    if rank != 0:
        return
    # roofline of the dominant kernel of this leg (the single-wave batch kernel): algorithmic bytes of the REFERENCE algorithm for rank 0's
    # what-ifs (attempts / scanned types from an untimed KS_FLAG_STATS batch), SURVEY 8d formula with the fixed R = 4, K = 8
    sflats = S.open_whatifs(parsed, pod_node, [sets[i] for i in mine], stats=True)
    S.upload_batch(sflats, local_rank)
    sres, _, _ = S.solve_batch(sflats)
    abytes = sum(algorithmic_bytes(T, r.stats, f.dims["P"]) for r, f in zip(sres, sflats))
    for f in sflats:
        f.close()
    k_s = statistics.mean(kms) / 1e3
    got = int(table.shape[0])
    out = {"metric": "pod-placement decisions/sec (Solve())", "value": total_pods * args.steps / elapsed, "unit": "decisions/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "int64", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[3]: {total_whatifs} consolidation what-ifs over 2048 existing nodes / {T} instance types, dealt to the ranks by predicted work (pods; longest first to the least loaded rank); "
                                  "one batched launch per rank + ONE RCCL all-gather of result records", "whatifs": total_whatifs, "decisions_per_step": total_pods,
                      "records_gathered": got, "parallelism": f"{world} ranks x <= {per} what-ifs", "pods_per_rank": None, "single_gpu_same_workload": n1, "weak": weak, "grid_sharded": grid_sh,
                      "n1_reference": "the `whatif_batch` object of the --gpus 1 line (same workload on one GPU); a single Solve() does not shard (replicas only)"},
           "roofline": {"kernel": "ks_pack<single wave> x what-ifs of rank 0 in one launch", "bound": "hbm", "achieved": abytes / k_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": abytes / k_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": abytes, "kernel_ms_mean": k_s * 1e3,
                        "formula": f"SURVEY 8d with R={ROOFLINE_R}, K={ROOFLINE_K}, summed over rank 0's {len(mine)} what-ifs",
                        "note": "a batch takes as long as its longest what-if (one wave each); the watermark / run commit skip most of the attempts the reference makes"}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
