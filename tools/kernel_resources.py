#!/usr/bin/env python3
"""Register / LDS budget of every kernel in ksolve.hip, from the compiler's own remarks (runs in the build container, no GPU):
    python tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "karpenter_core_amd", "csrc", "ksolve.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-c",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/ks_dev.o", src]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: ([A-Za-z \[\]/]+): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage   (ks_pack<FAST, BOUNDS, LEAN, waves>)")
print(f"{'kernel':58s} {'VGPRs':>6s} {'SGPRs':>6s} {'SGPR spill':>10s} {'VGPR spill':>10s} {'scratch B/lane':>14s} {'LDS B/block':>12s} {'waves/SIMD':>10s}")
for r in rows:
    print(f"{r['name']:58s} {r.get('VGPRs','?'):>6s} {r.get('TotalSGPRs', r.get('SGPRs','?')):>6s} {r.get('SGPRs Spill','?'):>10s} {r.get('VGPRs Spill','?'):>10s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>14s} {r.get('LDS Size [bytes/block]','?'):>12s} {r.get('Occupancy [waves/SIMD]','?'):>10s}")
