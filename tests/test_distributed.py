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
