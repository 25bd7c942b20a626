import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


# KS_TEST_SIM=1: run the `-m gpu` tests against the EMULATOR build of the kernels (tests/sim: the HIP source compiled by g++, a fibre per lane) -- test
# infrastructure for containers without a GPU.  ks_pack_rr and ks_pack's single-wave variants are emulated (tests/sim/build_sim.py); what neither takes is reported as skipped.
if os.environ.get("KS_TEST_SIM"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import simlib
    simlib.use_sim()

    @pytest.hookimpl(hookwrapper=True)
    def pytest_runtest_call(item):
        outcome = yield
        exc = outcome.excinfo
        if exc is not None and "emulator build" in str(exc[1]):
            outcome.force_exception(pytest.skip.Exception("declined by ks_pack_rr (not emulated further)"))
