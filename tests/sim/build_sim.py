"""TEST INFRASTRUCTURE: builds tests/sim/_build/libksolve.so -- karpenter_core_amd/csrc/ksolve.hip compiled by g++ against the lane-fibre
emulator (tests/sim/hip_sim.h) -- and a copy of libkshost.so next to it, so that the C-ABI parity tests can drive the pack kernels
(ks_pack_rr and, since round 5, ks_pack: -DKS_SIM_PACK) on the host, against the CPU oracle, in a container without a GPU.  Nothing outside tests/ uses this."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(ROOT, "karpenter_core_amd")
OUT = os.path.join(HERE, "_build")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force: bool = False, opt: str = "-O1") -> str:
    os.makedirs(OUT, exist_ok=True)
    ks_so = os.path.join(OUT, "libksolve.so")
    src = [os.path.join(PKG, "csrc", f) for f in ("ksolve.hip", "ks_pack_rr.inc", "ks_algebra.h")] + \
          [os.path.join(HERE, "hip_sim.h"), os.path.join(HERE, "hip_sim.cpp"), os.path.join(ROOT, "include", "ksolve.h")]
    if force or _newer(ks_so, src):
        subprocess.check_call(["g++", opt, "-g", "-std=c++17", "-fPIC", "-shared", "-w", "-DKS_SIM", "-I" + os.path.join(HERE, "fakeinc"),
                               "-DRR_WINDOW=1", "-DKS_SIM_PACK", "-include", os.path.join(HERE, "hip_sim.h"), "-x", "c++", src[0], os.path.join(HERE, "hip_sim.cpp"),
                               "-o", ks_so, "-pthread"])
    kh_so = os.path.join(OUT, "libkshost.so")
    host = os.path.join(PKG, "host")
    kh_src = [os.path.join(host, f) for f in ("encode.cpp", "api.cpp", "encode.hpp", "hreq.hpp", "ksp.hpp", "kspb.hpp")] + [ks_so]
    if force or _newer(kh_so, kh_src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", kh_so, os.path.join(host, "encode.cpp"),
                               os.path.join(host, "api.cpp"), "-L" + OUT, "-lksolve", "-pthread", "-Wl,-rpath,$ORIGIN"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
