#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>) into the files kept under profiles/:
  <tag>_bench.json                 the bench.py JSON line (plain run, default arguments)
  <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command: per-kernel calls / total / average ns
  <tag>_bench_pmc.json             per-kernel counter sums of the separate --pmc passes, and per ks_pack launch: HBM bytes (FETCH_SIZE x2 +
                                   WRITE_SIZE, MI355X_MICROARCH.md gfx950 note), instructions issued, wave-cycle split; keyed by the sha of the
                                   kernel source so bench.py only uses it for the kernel it was recorded with
usage (build container, after the gpurun call): tools/summarize_profile.py gpurun_out/prof_r02 r02"""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

src, tag = sys.argv[1], sys.argv[2]
line = json.loads([l for l in open(f"{src}/bench.json") if l.startswith("{")][-1])
json.dump(line, open(f"profiles/{tag}_bench.json", "w"), indent=1)
prof_line = None
try:
    prof_line = json.loads([l for l in open(f"{src}/bench_under_rocprof.json") if l.startswith("{")][-1])
except Exception:
    pass
for f in glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"profiles/{tag}_bench_kernel_stats.csv")


def sums(kind):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{src}/{kind}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            a = agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return agg


out = {"kernel_source_sha16": bench.kernel_source_sha16(), "pods": line["config"]["pods"], "instance_types": line["config"]["instance_types"],
       "command": "rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --whatifs 0   (one pass per counter group)",
       "per_kernel": {}}
allk = {}
for kind in ("fetch", "write", "insts", "cycles"):
    for k, ctrs in sums(kind).items():
        for c, (v, n) in ctrs.items():
            allk.setdefault(k, {})[c] = {"sum": v, "dispatches": n, "per_dispatch": v / n}
out["per_kernel"] = {k: allk[k] for k in sorted(allk)}
# the kernel that takes the headline Solve: ks_pack_rr since round 4 (when it ran), else the 8-wave ks_pack variant
pk = ([k for k in allk if k.startswith("ks_pack_rr")] if line.get("roofline", {}).get("kernel") == "ks_pack_rr" else []) or \
     [k for k in allk if "ks_pack" in k and "8>" in k.replace(" ", "")] or [k for k in allk if "ks_pack" in k]
if pk:
    k = pk[0]; c = allk[k]
    per = lambda name: c[name]["per_dispatch"] if name in c else None
    f, w = per("FETCH_SIZE"), per("WRITE_SIZE")
    inst_names = [n for n in c if n.startswith("SQ_INSTS_")]
    insts = sum(c[n]["per_dispatch"] for n in inst_names) if inst_names else None
    out["ks_pack_per_launch"] = {
        "kernel": k, "fetch_kb_reported": f, "write_kb_reported": w,
        "hbm_bytes_fetch_x2_plus_write": int((2 * f + w) * 1024) if f is not None and w is not None else None,
        "hbm_bytes_as_reported": int((f + w) * 1024) if f is not None and w is not None else None,
        "instructions": insts, "instruction_mix": {n: c[n]["per_dispatch"] for n in sorted(inst_names)},
        "wave_cycles": {n: per(n) for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE") if per(n) is not None},
        "shader_clock_ghz": 2.4,
        "note": "FETCH_SIZE / WRITE_SIZE are reported in KB; MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide "
                "coalesced reads, so it is doubled; this kernel's reads are scattered 8/16-B accesses for which the counter is uncalibrated -- the "
                "as-reported figure is kept beside it.  SQ_INSTS_* are summed over the workgroup's 8 waves."}
# the other two dominant kernels: per launch of the single-wave batch kernel over the 512 what-ifs / of the general 4-wave kernel on config #5 at 250 000 pods
def leg(tag, pick, workgroups, waves):
    k_all = {}
    for kind in ("fetch", "write", "insts", "cycles"):
        for k, ctrs in sums(f"{kind}_{tag}").items():
            for c, (v, n) in ctrs.items():
                k_all.setdefault(k, {})[c] = {"sum": v, "dispatches": n, "per_dispatch": v / n}
    ks = [k for k in k_all if pick(k.replace(" ", ""))]
    if not ks:
        return None
    # (the timed launches are the ones with the most pods: the kernel of that name with the largest instruction count per dispatch, where several variants ran)
    k = max(ks, key=lambda k: k_all[k].get("SQ_INSTS_VALU", {"per_dispatch": 0})["per_dispatch"]); c = k_all[k]
    per = lambda name: c[name]["per_dispatch"] if name in c else None
    f, w = per("FETCH_SIZE"), per("WRITE_SIZE"); inst_names = [n for n in c if n.startswith("SQ_INSTS_")]
    wc, wa = per("SQ_WAVE_CYCLES"), per("SQ_WAIT_ANY")
    return {"kernel": k, "dispatches_seen": {n: c[n]["dispatches"] for n in ("FETCH_SIZE", "WRITE_SIZE") if n in c}, "fetch_kb_reported": f, "write_kb_reported": w,
            "hbm_bytes_fetch_x2_plus_write": int((2 * f + w) * 1024) if f is not None and w is not None else None,
            "hbm_bytes_as_reported": int((f + w) * 1024) if f is not None and w is not None else None,
            "instructions": sum(c[n]["per_dispatch"] for n in inst_names) if inst_names else None, "instruction_mix": {n: c[n]["per_dispatch"] for n in sorted(inst_names)},
            "wave_cycles": {n: per(n) for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE") if per(n) is not None},
            "wait_fraction": (wa / wc) if wc and wa is not None else None, "workgroups": workgroups, "waves_per_workgroup": waves}
out["legs"] = {}
wi = leg("wi", lambda k: "ks_pack<" in k and k.rstrip(">").endswith(",1") and "true,false,true" in k, 512, 1)
if wi: out["legs"]["whatifs"] = dict(wi, command="bench.py --whatifs-only")
c5 = leg("c5", lambda k: "ks_pack<" in k and k.rstrip(">").endswith(",4"), 1, 4)
if c5: out["legs"]["config5"] = dict(c5, command="bench.py --config5 250000 --steps 1 --warmup 0")
if prof_line:
    out["bench_under_rocprof"] = {"value": prof_line["value"], "pack_kernel_ms_mean": prof_line["phases_ms_mean"]["pack_kernel_ms"]}
out["bench_plain"] = {"value": line["value"], "pack_kernel_ms_mean": line["phases_ms_mean"]["pack_kernel_ms"]}
json.dump(out, open(f"profiles/{tag}_bench_pmc.json", "w"), indent=1)
print(json.dumps(out.get("ks_pack_per_launch"), indent=1))
for row in csv.DictReader(open(f"profiles/{tag}_bench_kernel_stats.csv")):
    if "ks_pack" in row.get("Name", ""):
        print(row)
