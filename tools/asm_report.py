#!/usr/bin/env python3
"""What the compiler made of the pack kernel (build container, no GPU needed):   tools/asm_report.py [--uniformity]
  * registers / spills / LDS of every ks_pack variant
  * for the 8-wave LEAN kernel: scratch (VGPR-spill) accesses and FLAT accesses with the source lines they belong to, the static instruction
    mix of the resolver loop, and the loops the compiler treats as divergent (exec-masked back edges)
  * --uniformity: LLVM's own uniformity analysis on the kernel's IR -- the source lines of branches it takes for divergent.  The Solve loop's
    sequential state is wave-uniform by construction; where the analysis disagrees the state lives in VGPRs and every branch is an exec-mask
    dance (DESIGN.md 4.1 "Uniform means scalar").
Three things this report caught in round 2: arrays above 64 KiB of LDS (base registers hoisted and spilled to scratch), a pointer chosen
between LDS and global memory (FLAT accesses), and a loop counter the analysis took for divergent."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "karpenter_core_amd", "csrc", "ksolve.hip")
K8 = "_Z7ks_packILb1ELb0ELb1ELi8EEvPK7DevProbPK8DevStatej"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
tmp = tempfile.mkdtemp(prefix="ksasm")


def compile_to(ext, extra):
    out = os.path.join(tmp, "ks." + ext)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "-gline-tables-only", "--cuda-device-only", "-Wno-unused-value"] + extra + ["-o", out, SRC],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


src = open(SRC).read().split("\n")
asm = compile_to("s", [])
print("== registers / spills / LDS")
for m in re.finditer(r"\.group_segment_fixed_size: (\d+)\n(?:.*\n)*?\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_spill_count: (\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)", asm):
    if "ks_pack" in m.group(2):
        print("  %-52s lds %6s  vgpr %3s  sgpr spills %3s  vgpr spills %2s" % (m.group(2)[3:55], m.group(1), m.group(4), m.group(3), m.group(5)))
lines = asm.split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(K8 + ":")][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
cur, scratch, flat, per_line, labels = None, [], collections.Counter(), collections.Counter(), {}
for i in range(start, end):
    l = lines[i]
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s+\d+", l)
    if m:
        cur = int(m.group(1)); continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i; continue
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    per_line[cur] += 1
    if "scratch_" in t:
        scratch.append((cur, t.split(";")[0].strip()))
    if t.startswith("flat_"):
        flat[cur] += 1
print("== 8-wave kernel: %d instructions" % sum(per_line.values()))
print("== scratch accesses (VGPR spills): %d" % len(scratch))
for ln, t in scratch:
    print("  L%-5s %-60s | %s" % (ln, t, src[ln - 1].strip()[:90] if ln else ""))
w0 = [i for i, l in enumerate(src) if "while (k < rn) {" in l][0] + 1
w1 = [i for i, l in enumerate(src) if "if (n_ok == rn) CUT(16);" in l][0] + 1
print("== resolver loop (source lines %d-%d): %d static instructions" % (w0, w1, sum(v for k, v in per_line.items() if k and w0 <= k <= w1)))
print("== FLAT accesses after the one-off initialisation (source lines): ", sorted((k, v) for k, v in flat.items() if k and k > w0 - 600 and k < w1 + 450))
div = collections.Counter()
cur = None
for i in range(start, end):
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s+\d+", lines[i])
    if m:
        cur = int(m.group(1)); continue
    m = re.match(r"\s*s_cbranch_exec(?:n?)z\s+(\.LBB\d+_\d+)", lines[i])
    if m and labels.get(m.group(1), 10 ** 9) < i and cur:
        div[cur] += 1
print("== loops with exec-masked back edges (the compiler takes their trip count for divergent), by source line:")
for ln, n in sorted(div.items()):
    print("  L%-5d x%d  %s" % (ln, n, src[ln - 1].strip()[:110]))
if "--uniformity" in sys.argv:
    ll = compile_to("ll", ["-emit-llvm"])
    opt = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin", "opt")
    p = subprocess.run([opt, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-passes=print<uniformity>", "-disable-output", os.path.join(tmp, "ks.ll")], stderr=subprocess.PIPE, text=True)
    loc = {}
    for m in re.finditer(r"^!(\d+) = !DILocation\(line: (\d+), column: \d+, scope: !\d+(?:, inlinedAt: !(\d+))?\)", ll, re.M):
        loc[int(m.group(1))] = (int(m.group(2)), int(m.group(3)) if m.group(3) else None)

    def outer(n):
        last = None
        while n is not None and n in loc:
            last, n = loc[n][0], loc[n][1]
        return last
    u = p.stderr.split("\n")
    a = [i for i, l in enumerate(u) if "UniformityInfo for function '" + K8 in l][0]
    b = [i for i, l in enumerate(u) if i > a and "UniformityInfo for function" in l] + [len(u)]
    cnt = collections.Counter()
    for l in u[a:b[0]]:
        if "DIVERGENT" in l and " br i1" in l:
            m = re.search(r"!dbg !(\d+)", l)
            if m and outer(int(m.group(1))):
                cnt[outer(int(m.group(1)))] += 1
    print("== branches LLVM's uniformity analysis takes for divergent, by (outermost) source line:")
    for ln, n in sorted(cnt.items()):
        print("  L%-5d x%d  %s" % (ln, n, src[ln - 1].strip()[:110]))
