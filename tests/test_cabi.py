"""CPU-side checks of the boundary: the library builds, loads, exports every symbol include/ksolve.h
declares, the host flattening runs on every config shape, and -- because there is no GPU here -- the
compute entry points refuse loudly instead of falling back."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as ge
from karpenter_core_amd import fake, scheduler as S, workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    ge.build()


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ksolve.h")).read()
    names = set(re.findall(r"\b(ks_[a-z_0-9]+)\s*\(", hdr))
    assert {"ks_solve", "ks_solve_dev", "ks_solve_batch", "ks_solve_batch_dev", "ks_problem_upload", "ks_problem_free",
            "ks_feasibility_grid", "ks_probe_intersection", "ks_probe_compatible", "ks_device_count", "ks_last_error",
            "ks_version"} <= names
    ks, _ = S.libs()
    for n in names:
        assert hasattr(ks, n), f"libksolve.so does not export {n}"


def test_host_library_exports_every_declared_symbol():
    """include/kshost.h is the boundary the Python mirror and the GPU tests actually drive (ksh_*)."""
    hdr = open(os.path.join(ROOT, "include", "kshost.h")).read()
    names = set(re.findall(r"\b(ksh_[a-z_0-9]+)\s*\(", hdr))
    assert {"ksh_parse", "ksh_solve_from_pods", "ksh_open", "ksh_upload", "ksh_solve", "ksh_solve_batch", "ksh_open_whatifs", "ksh_price_filter",
            "ksh_result_text", "ksh_result_summary", "ksh_grid"} <= names
    _, kh = S.libs()
    for n in names:
        assert hasattr(kh, n), f"libkshost.so does not export {n}"


def test_solve_from_pods_refuses_without_gpu():
    if S.device_count() > 0:
        pytest.skip("GPU present")
    pp = S.ParsedProblem(W.config1(pods=20, types=5))
    with pytest.raises(S.KSolveError) as ei:
        S.solve_from_pods(pp, 0)
    assert ei.value.code == S.KS_ERR_DEVICE


def test_no_cpu_fallback_without_gpu():
    if S.device_count() > 0:
        pytest.skip("GPU present")
    fp = S.FlatProblem(W.config1(pods=20, types=5))
    with pytest.raises(S.KSolveError) as ei:
        fp.solve()
    assert ei.value.code == S.KS_ERR_DEVICE


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "karpenter_core_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in src and "oracle_py" not in src and "ko_solve" not in src, f
    for f in os.listdir(os.path.join(ROOT, "tools")):          # measurement helpers are not test infrastructure either
        if f.endswith((".py", ".sh")):
            src = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            assert f.startswith("debug_") or ("oracle_py" not in src and "liboracle" not in src and "from oracle" not in src), f
    bench = open(os.path.join(ROOT, "bench.py")).read()          # bench.py: the cpu_baseline leg only
    assert bench.count("from oracle import") == 1 and bench.index("from oracle import") > bench.index("if not args.no_cpu_baseline")


@pytest.mark.parametrize("name,problem,dims", [
    ("config1", lambda: W.config1(), dict(P=1000, T=50, M=1, E=0, G=0)),
    ("config2-small", lambda: W.config2(pods=1500), dict(P=1500, T=500, M=5)),
    ("config3-small", lambda: W.config3(pods=3500), dict(P=3500, T=2000, M=1)),
    ("config5-small", lambda: W.config5(pods=3000, sizes=10), dict(P=3000, M=2)),
])
def test_flattening_shapes(name, problem, dims):
    fp = S.FlatProblem(problem())
    for k, v in dims.items():
        assert fp.dims[k] == v, (name, fp.dims)
    assert fp.dims["K"] <= 32 and fp.dims["R"] <= 8 and fp.dims["C"] <= fp.dims["P"] * 3
    fp.close()


def test_whatif_flattening():
    its, prov, nodes, bound = W.cluster_snapshot(existing=64, sizes=5, seed=3)
    pr = W.whatif(its, prov, nodes, bound, [0, 1, 2])
    fp = S.FlatProblem(pr)
    assert fp.dims["E"] == 61 and fp.dims["P"] == sum(len(b) for b in bound[:3])
    fp.close()


def test_unsupported_is_loud():
    pr = W.reference_benchmark(50, instance_count=100)   # `integer` label with 100 distinct values ...
    S.FlatProblem(pr).close()                             # ... is fine while only instance types carry it (the key is left out)
    from karpenter_core_amd.model import Expr
    pr.pods[0].required_affinity = [[Expr(fake.LABEL_INTEGER, "Gt", ["50"])]]   # a pod bounds it: the 100 values fall into classes by that bound (tests/test_value_classes.py)
    S.FlatProblem(pr).close()
    from karpenter_core_amd.model import DO_NOT_SCHEDULE, LabelSelector, TopologySpreadConstraint
    pr.pods[1].spread = [TopologySpreadConstraint(1, fake.LABEL_INTEGER, DO_NOT_SCHEDULE, LabelSelector({"app": "x"}))]   # a topology key needs every value as a domain of its own
    with pytest.raises(S.KSolveError) as ei:
        S.FlatProblem(pr)
    assert ei.value.code == S.KS_ERR_UNSUPPORTED


def test_ambiguous_queue_order_is_refused():
    """queue.go:102-108 breaks ties by UID: two pods that tie on cpu, memory, creation timestamp AND UID leave the order undefined -- refused at
    flattening time, with or without cluster pods (two code paths: the UID table, or the check on the sorted queue)."""
    from karpenter_core_amd.model import ClusterPod
    base = W.config3(pods=70, sizes=4, seed=3)
    base.pods[5].uid = base.pods[40].uid
    base.pods[5].containers, base.pods[5].creation_ts = base.pods[40].containers, base.pods[40].creation_ts
    for cps in ([], [ClusterPod(uid="bound-1", namespace="default", node_name="nowhere")]):
        base.cluster_pods = cps
        with pytest.raises(S.KSolveError):
            S.FlatProblem(base)


@pytest.mark.gpu
def test_two_concurrent_solves():
    """The provisioner and the deprovisioner are two goroutines that may call Solve at the same time (provisioner.go:102-104,
    deprovisioning/controller.go:103-105): two threads solve different problems concurrently through the C ABI (ctypes drops the GIL)
    and both get the oracle's answer, repeatedly."""
    import threading
    from oracle import oracle_py as O
    probs = [W.config3(pods=3000, sizes=10, seed=21), W.config2(pods=3000, sizes=10, seed=22), W.config5(pods=1500, sizes=8, seed=23)]
    wants = [O.solve(p).canonical() for p in probs]
    got, errs = [None] * len(probs), []

    def run(i):
        try:
            for _ in range(4):
                r = S.solve_problem(probs[i]).canonical()
                assert r == wants[i]
                got[i] = r
        except BaseException as e:      # noqa: BLE001 -- surfaced below
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(probs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert got == wants


def test_concurrent_flattening_matches_serial():
    """The host flattening runs its parallel phases on a shared pool of worker threads; a second caller arriving while the pool is busy
    spawns threads of its own.  Four threads flattening different problems at once (provisioner and deprovisioner goroutines do) must
    produce the flat problems a serial run produces, fingerprint for fingerprint, run after run."""
    import threading
    problems = [W.config3(pods=6000, sizes=10, seed=3), W.config2(pods=5000, sizes=8, seed=4), W.config5(pods=5000, sizes=50, seed=5),
                W.config1(pods=7000, types=40, seed=6)]
    parsed = [S.ParsedProblem(p) for p in problems]

    def flatten(i):
        fp = S.FlatProblem(problems[i])
        f = fp.fingerprint()
        fp.close()
        return f

    serial = [flatten(i) for i in range(len(problems))]
    for _ in range(3):
        got = [None] * len(problems)
        ths = [threading.Thread(target=lambda i=i: got.__setitem__(i, flatten(i))) for i in range(len(problems))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert got == serial
    del parsed


def test_flattening_fingerprints_are_pinned():
    """The host flattening of a fixed problem set hashes to the committed fingerprints (tests/golden/make_flat_fingerprints.py): changes of
    the host code that are meant to be result-neutral -- threading, caching, the order work is done in -- are caught on the CPU."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_flat_fingerprints as M
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "flat_fingerprints.json")))
    got = M.compute()
    assert got == want, sorted(k for k in want if got.get(k) != want[k])


def test_headers_are_plain_c(tmp_path):
    """include/*.h must be usable from C (what cgo compiles): tests/cabi_usage.c -- parse an environment (and take the same one in through ksh_env_ingest), hand one pod over through the binary door,
    flatten, ask for a Solve -- is compiled as C99 with -Wall -Werror -pedantic, linked against both libraries and run.  Without a GPU the Solve
    must be refused with KS_ERR_DEVICE (no CPU path); with one it solves."""
    import dataclasses
    import subprocess
    pkg = os.path.join(ROOT, "karpenter_core_amd")
    exe = str(tmp_path / "cabi_usage")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_usage.c"),
                           "-o", exe, "-L", pkg, "-lkshost", "-lksolve", "-Wl,-rpath," + pkg])
    env_file = tmp_path / "env.ksp"
    env_file.write_text(dataclasses.replace(W.config1(pods=1, types=5), pods=[]).to_ksp())
    import numpy as np
    from karpenter_core_amd.model import env_to_block
    blk = env_to_block(W.config1(pods=1, types=5))
    blk_file = tmp_path / "env.block"
    blk_file.write_bytes(np.asarray([blk["n_strings"], blk["n_words"]], dtype=np.uint32).tobytes() + blk["str_off"].tobytes() + blk["words"].tobytes() + blk["str_bytes"].tobytes())
    out = subprocess.run([exe, str(env_file), str(blk_file)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pods 1 specs 1 flat P=1 C=1 T=5" in out.stdout
    assert "binary environment: the same flat problem" in out.stdout
    if S.device_count() == 0:
        assert "solve refused: -3" in out.stdout
    else:
        assert "solved in" in out.stdout


def test_pod_and_node_named_like_section_keywords():
    """KSP1's optional sections (VOL after a pod, VL / VU after a state node) are told from the next record by peeking at a bare token: the writer always emits them, so
    a pod whose uid is "VOL" (or a node called "VL" / "VU") parses as what it is."""
    from karpenter_core_amd import fake
    from karpenter_core_amd.model import Container, Pod, Problem
    from oracle import oracle_py as O
    its = fake.default_instance_types()
    pods = [Pod(uid="p0", containers=[Container(requests={"cpu": "1"})]), Pod(uid="VOL", containers=[Container(requests={"cpu": "1"})]),
            Pod(uid="VL", containers=[Container(requests={"cpu": "1"})])]
    pr = Problem(instance_types=its, provisioners=[fake.provisioner("default", len(its))], pods=pods, extra_well_known=fake.EXTRA_WELL_KNOWN)
    r = O.solve(pr)
    assert sorted(q for n in r.new_nodes for q in n.pods) == [0, 1, 2]


def test_build_refuses_a_pack_kernel_that_fell_off_its_register_budget(tmp_path):
    """ks_pack_rr keeps its nodes in registers, a few VGPRs below the limit; past it the allocator moves the arrays to scratch and the kernel runs three times slower
    without any error.  build() reads the compiler's resource remarks: a library like that is removed and the build fails (the remarks below are the two the round saw)."""
    def remarks(vgprs, spill, scratch):
        pre = "csrc/ks_pack_rr.inc:1425:1: remark:     "
        return "\n".join(["csrc/ks_pack_rr.inc:1425:1: remark: Function Name: _Z10ks_pack_rrPK7DevProbPK8DevStatej [-Rpass-analysis=kernel-resource-usage]",
                          pre + f"VGPRs: {vgprs} [-Rpass-analysis=kernel-resource-usage]", pre + f"ScratchSize [bytes/lane]: {scratch} [-Rpass-analysis=kernel-resource-usage]",
                          pre + f"VGPRs Spill: {spill} [-Rpass-analysis=kernel-resource-usage]"])
    so = tmp_path / "libksolve.so"
    so.write_bytes(b"x")
    ge._check_register_budget(remarks(248, 0, 336), str(so))
    assert so.exists()
    with pytest.raises(RuntimeError, match="register budget"):
        ge._check_register_budget(remarks(256, 1, 1356), str(so))
    assert not so.exists()


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md is the map from the C ABI to the reference's symbols: an entry point the headers declare and the document does not mention is a gap in that map."""
    import re
    names = set()
    for h in ("ksolve.h", "kshost.h"):
        names |= set(re.findall(r"\b(ksh?_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert not [n for n in sorted(names) if n not in doc]
