"""The N>1 path on CPU: world_size-2 `gloo` process group, what-ifs sharded round-robin, one all-gather of
result records.  The solver plugged in here is the CPU oracle (no GPU in this container); on the GPU box the
same function runs with the HIP path (test_whatif_records_gpu)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from karpenter_core_amd import consolidation as C, workloads as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_many(problems):
    from oracle import oracle_py
    return [oracle_py.solve(p) for p in problems]


def _problems():
    its, prov, nodes, bound = W.cluster_snapshot(existing=24, sizes=4, seed=11)
    return [W.whatif(its, prov, nodes, bound, list(range(0, i + 1))) for i in range(5)] + \
           [W.whatif(its, prov, nodes, bound, [i]) for i in (7, 9, 13)]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table = C.solve_whatifs(_problems(), _oracle_many)
        out[rank] = table.tolist()
    finally:
        dist.destroy_process_group()


def test_shard_round_robin():
    assert C.shard(7, 0, 2) == [0, 2, 4, 6] and C.shard(7, 1, 2) == [1, 3, 5]
    assert sorted(sum((C.shard(512, r, 8) for r in range(8)), [])) == list(range(512))


def test_two_rank_gloo_matches_serial():
    probs = _problems()
    serial = C.solve_whatifs(probs, _oracle_many).tolist()
    assert [r[0] for r in serial] == list(range(len(probs)))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == serial and out[1] == serial


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # what `bench.py --gpus N` hands over: binary records [id, n_new, n_unscheduled, InstanceTypeOptions words] of this rank's share, i mod N
        ids = C.shard(5, rank, world)
        rec = torch.tensor([[i, i % 2, 0, (1 << i) - 1, -(i + 1)] for i in ids], dtype=torch.int64).reshape(len(ids), 5)
        out[rank] = C.all_gather_records(rec, (5 + world - 1) // world).tolist()
    finally:
        dist.destroy_process_group()


def test_two_rank_record_gather():
    """The record builder of the sharded path: uneven shares (3 + 2 records), padding rows dropped, rows in what-if order on every rank."""
    want = [[i, i % 2, 0, (1 << i) - 1, -(i + 1)] for i in range(5)]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gather_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == want and out[1] == want
    assert C.all_gather_records(torch.tensor(want[::-1], dtype=torch.int64), 5).tolist() == want      # world of one: sorted by id


@pytest.mark.gpu
def test_whatif_records_gpu():
    probs = _problems()
    want = C.solve_whatifs(probs, _oracle_many).tolist()
    got = C.solve_whatifs(probs, C.gpu_solve_many).tolist()
    assert got == want


def _run_bench(extra_env, *argv, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher must start two ranks itself (the driver's command shape).  KS_BENCH_DRY=1: the launch and the
    exchange plumbing over gloo without a GPU -- ids dealt out i mod N, ONE all-gather, 512 records back in what-if order on rank 0."""
    out = _run_bench({"KS_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert out["n_gpus"] == 2 and out["dry"] is True
    assert out["config"]["records_gathered"] == 512 and out["config"]["ids_in_order"] is True


def test_bench_gpus_mismatch_is_loud():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", KS_BENCH_DRY="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_gpus_2_fanout_on_one_gpu():
    """The whole N>1 leg as the driver launches it, rehearsed on a 1-GPU box: two ranks share the device, records are built on the device and
    gathered over gloo.  512 records, equal to what one GPU computes alone for the same what-ifs."""
    out = _run_bench({"KS_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert out["n_gpus"] == 2 and out["config"]["records_gathered"] == 512
    assert out["config"]["single_gpu_same_workload"]["records_equal_gathered"] is True
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


def test_lpt_deal_balances_and_matches_the_c_abi():
    """Round 5: what-ifs are dealt by predicted work, longest first to the least loaded rank (consolidation.deal == include/ksolve.h ks_deal_lpt): every id once, the longest
    what-if's rank carries little else, loads within the longest item of each other."""
    from karpenter_core_amd import scheduler as S
    sets = W.config4_sets(512, 2048, 45)
    weights = [len(cs) * 20 for cs in sets]
    for world in (1, 2, 4, 8):
        dealt = C.deal(weights, world)
        assert sorted(sum(dealt, [])) == list(range(512))
        loads = [sum(weights[i] for i in d) for d in dealt]
        assert max(loads) - min(loads) <= max(weights)
        if world == 2:
            assert (max(loads) - min(loads)) / max(loads) < 0.10
        of = S.deal_lpt(weights, world)
        assert [sorted(i for i in range(512) if of[i] == r) for r in range(world)] == dealt


@pytest.mark.gpu
def test_whatifs_sharded_in_one_c_call():
    """The fan-out in the C ABI (ksh_solve_whatifs_sharded): two shards -- both on the one GPU of this box -- solved concurrently, their records gathered into one table by
    id: equal to the records of the unsharded batch."""
    from karpenter_core_amd import scheduler as S
    its, prov, nodes, bound = W.cluster_snapshot(existing=96, sizes=10, seed=13)
    snap, pod_node = W.snapshot_problem(its, prov, nodes, bound, False)
    sets = [list(range(0, i + 1)) for i in range(12)] + [[i] for i in (20, 33, 47, 60)]
    parsed = S.ParsedProblem(snap)
    words = (len(its) + 63) // 64
    weights = [sum(len(bound[c]) for c in cs) for cs in sets]
    dealt = C.deal(weights, 2)
    shards = [S.open_whatifs(parsed, pod_node, [sets[i] for i in d], device=0) for d in dealt]
    allf = S.open_whatifs(parsed, pod_node, sets, device=0)
    try:
        for sh in shards:
            S.upload_batch(sh, 0)
        S.upload_batch(allf, 0)
        rows, kms = S.solve_whatifs_sharded(shards, dealt, words)
        S.solve_batch_resident(allf)
        import torch
        one = torch.full((len(sets), 3 + words), -1, dtype=torch.int64, device="cuda:0")
        S.result_records_dev(allf, list(range(len(sets))), words, one)
        assert rows.astype("int64").tolist() == one.cpu().tolist() and kms > 0
    finally:
        for f in allf + [f for sh in shards for f in sh]:
            f.close()


# ---- the static feasibility grid's rows split over the ranks, ONE all-gather of bit-rows (scheduler.sharded_grid, consolidation.all_gather_grid_rows) ----
class _FakeGrid:
    """Stands in for a FlatProblem on the CPU: row i of the grid is a function of i; what gets installed is recorded."""
    def __init__(self, m, c, t):
        import numpy as np
        self.dims = {"M": m, "C": c, "T": t}; self.tw = (t + 63) // 64
        self.table = np.zeros((m * c, self.tw), dtype=np.uint64); self.complete = False

    @staticmethod
    def row(i, tw):
        import numpy as np
        return (np.arange(tw, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(i * 2654435761 + 1)) | (np.uint64(1) << np.uint64(63))

    def grid_rows(self, lo, hi, dev_ptr=0):
        import numpy as np
        rows = np.stack([self.row(i, self.tw) for i in range(lo, hi)]) if hi > lo else np.zeros((0, self.tw), dtype=np.uint64)
        self.table[lo:hi] = rows
        return rows, 0.0

    def grid_install(self, lo, hi, rows=None, dev_ptr=0, complete=False):
        if hi > lo:
            self.table[lo:hi] = rows
        self.complete = self.complete or complete


def _grid_worker(rank, world, port, out):
    from karpenter_core_amd import scheduler as S
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fp = _FakeGrid(3, 37, 200)      # 111 rows over 2 ranks: 55 + 56
        S.sharded_grid(fp, rank, world, C.all_gather_grid_rows)
        out[rank] = (fp.table.tolist(), fp.complete)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_grid_is_the_whole_grid():
    import numpy as np
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grid_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    want = np.stack([_FakeGrid.row(i, 4) for i in range(111)]).tolist()
    assert out[0] == (want, True) and out[1] == (want, True)


def _grid_gpu_worker(rank, world, port, out):
    import hashlib, json
    from karpenter_core_amd import scheduler as S
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = W.config2(pods=3000, sizes=10, seed=43)
        fp = S.FlatProblem(p); fp.upload(0)
        ms = S.sharded_grid(fp, rank, world, C.all_gather_grid_rows)      # this rank's share of the rows on the device, the others' installed from the ONE gather
        res = fp.solve()                                                     # (the Solve does not build the grid again: every row is in)
        out[rank] = (hashlib.sha256(json.dumps(res.canonical(), sort_keys=True).encode()).hexdigest(), ms >= 0.0)
        fp.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_grid_on_one_gpu_solves_like_the_oracle():
    """SURVEY 8e row 2, rehearsed on the 1-GPU box: two ranks share the device, each builds its half of the feasibility grid's rows there, ONE all-gather (gloo) of the bit-rows,
    each installs the other's -- and the Solve over the gathered grid is the oracle's, on both ranks (a row left out or misplaced changes InstanceTypeOptions)."""
    import hashlib, json
    from oracle import oracle_py
    p = W.config2(pods=3000, sizes=10, seed=43)
    want = hashlib.sha256(json.dumps(oracle_py.solve(p).canonical(), sort_keys=True).encode()).hexdigest()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grid_gpu_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == (want, True) and out[1] == (want, True)
