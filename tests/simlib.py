"""TEST INFRASTRUCTURE: point karpenter_core_amd.scheduler at the emulator build of the two libraries (tests/sim/_build), so that the tests that
normally need a GPU can drive the register-resident pack kernel's SOURCE on the host, through the same C ABI, and compare it with the oracle.
The product never does this: scheduler.libs() loads karpenter_core_amd/libksolve.so (hipcc) and nothing else."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def use_sim():
    """Build (if stale) and load the emulator libraries in place of the HIP ones for this process.  Returns the scheduler module."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "sim")); import build_sim
    from karpenter_core_amd import scheduler
    out = build_sim.build()
    if getattr(scheduler, "_SIM_ACTIVE", False):
        return scheduler
    real_here = scheduler._HERE
    scheduler._HERE = out
    scheduler._LIBS = None
    try:
        scheduler.libs()
    finally:
        scheduler._HERE = real_here
    scheduler._SIM_ACTIVE = True
    return scheduler
