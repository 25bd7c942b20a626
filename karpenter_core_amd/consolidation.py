"""Consolidation what-if fan-out across GPUs (BASELINE config #4).

Every deprovisioning what-if is an independent `simulateScheduling` (reference
pkg/controllers/deprovisioning/helpers.go:42-115): a pure function of (cluster snapshot, candidate set).
What-if i goes to rank i mod world; each rank solves its shard in ONE batched launch (one workgroup per
what-if, `scheduler.solve_batch`), and ONE all-gather of fixed-size result records returns everything the
reference's callers read from a simulation (consolidation.go:190-260):

    record = [whatif_id, n_new_nodes, n_unscheduled, first new node's InstanceTypeOptions as TW 64-bit words]

`allPodsScheduled` is n_unscheduled == 0; `len(newNodes)` gates replace-vs-delete; the option mask feeds
filterByPrice.  torch.distributed is plumbing only: backend "nccl" is RCCL over xGMI on the GPU box, "gloo"
in the CPU tests.  There is exactly one collective on the data path.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist

from .model import Problem, SolveResult


def shard(n: int, rank: int, world: int) -> List[int]:
    """What-if indices owned by `rank` (round-robin: neighbouring prefixes have neighbouring cost)."""
    return list(range(rank, n, world))


def record_width(n_instance_types: int) -> int:
    return 3 + (n_instance_types + 63) // 64


def to_record(whatif_id: int, res: SolveResult, type_index: dict, width: int) -> torch.Tensor:
    rec = torch.zeros(width, dtype=torch.int64)
    rec[0], rec[1], rec[2] = whatif_id, len(res.new_nodes), len(res.unscheduled)
    if res.new_nodes:
        for name in res.new_nodes[0].instance_types:
            t = type_index[name]
            w = 3 + t // 64
            v = int(rec[w]) & 0xFFFFFFFFFFFFFFFF
            v |= 1 << (t % 64)
            rec[w] = v - (1 << 64) if v >= (1 << 63) else v
    return rec


def solve_whatifs(problems: Sequence[Problem], solve_many: Callable[[List[Problem]], List[SolveResult]],
                  device: str = "cpu") -> torch.Tensor:
    """Solve all what-ifs across the process group; every rank returns the full [n, width] record table."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(problems)
    width = record_width(len(problems[0].instance_types)) if n else 3
    type_index = {it.name: i for i, it in enumerate(problems[0].instance_types)} if n else {}
    mine = shard(n, rank, world)
    results = solve_many([problems[i] for i in mine]) if mine else []
    per_rank = (n + world - 1) // world
    local = torch.full((per_rank, width), -1, dtype=torch.int64)
    for slot, (i, res) in enumerate(zip(mine, results)):
        local[slot] = to_record(i, res, type_index, width)
    local = local.to(device)
    if world > 1:
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)                 # the single exchange step of the path
        table = torch.stack(gathered, 0).reshape(world * per_rank, width)
    else:
        table = local
    table = table[table[:, 0] >= 0]
    order = torch.argsort(table[:, 0])
    return table[order].cpu()


def gpu_solve_many(problems: List[Problem]) -> List[SolveResult]:
    """One batched launch on this rank's GPU (no CPU path: raises without a gfx950 device)."""
    from . import scheduler
    flats = [scheduler.FlatProblem(p) for p in problems]
    try:
        res, _, _ = scheduler.solve_batch(flats)
        return res
    finally:
        for f in flats:
            f.close()
