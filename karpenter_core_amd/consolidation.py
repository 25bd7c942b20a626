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

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from .model import Problem, SolveResult


def shard(n: int, rank: int, world: int) -> List[int]:
    """What-if indices owned by `rank` (round-robin: neighbouring prefixes have neighbouring cost)."""
    return list(range(rank, n, world))


def deal(weights: Sequence[int], world: int) -> List[List[int]]:
    """Which rank solves which what-if, by predicted work (e.g. the pods of its candidate nodes): longest first, each to the least loaded rank (ties: the lower rank) --
    the LPT rule, the same as include/ksolve.h ks_deal_lpt.  Round-robin (`shard`) leaves a step as long as its longest what-if plus whatever i mod N put beside it;
    this puts the longest what-if (nearly) alone.  Returns the what-if ids of every rank, ascending."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(weights[i]) or 1
    return [sorted(x) for x in out]


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


def all_gather_records(records: torch.Tensor, per_rank: int) -> torch.Tensor:
    """The single exchange step of the sharded what-if path: this rank's [m, width] int64 result records (column 0 = what-if id, m <= per_rank)
    are padded to per_rank rows with id -1, all-gathered once, and returned as the full table ordered by what-if id (on `records`' device).
    Used by `solve_whatifs` and by `bench.py --gpus N`; a world of one returns the records sorted."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    width = records.shape[1]
    local = torch.full((per_rank, width), -1, dtype=torch.int64, device=records.device)
    local[: records.shape[0]] = records
    if world > 1:
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        table = torch.cat(gathered, 0)
    else:
        table = local
    table = table[table[:, 0] >= 0]
    return table[torch.argsort(table[:, 0])]


def all_gather_grid_rows(rows, device: str = "cpu"):
    """The one exchange step of the sharded static grid (scheduler.sharded_grid): this rank's bit-rows (numpy uint64 [m, TW]; the ranks' shares differ by at
    most one row) all-gathered ONCE over the process group -- RCCL over xGMI with device="cuda", gloo on the CPU -- and handed back as one numpy array per rank, in rank order."""
    import numpy as np
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [rows]
    tw = rows.shape[1]
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(counts, n)                                  # (the shares' sizes are a function of M * C and the world size; gathered so that a mismatch is loud, not silent)
    counts = [int(c.item()) for c in counts]
    per = max(counts)
    local = torch.zeros((per, tw), dtype=torch.int64, device=device)
    local[: rows.shape[0]] = torch.from_numpy(rows.view(np.int64)).to(device)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    return [g[:c].cpu().numpy().view(np.uint64) for g, c in zip(gathered, counts)]


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
    local = torch.full((len(mine), width), -1, dtype=torch.int64)
    for slot, (i, res) in enumerate(zip(mine, results)):
        local[slot] = to_record(i, res, type_index, width)
    return all_gather_records(local.to(device), per_rank).cpu()


def gpu_solve_many(problems: List[Problem], device: Optional[int] = None) -> List[SolveResult]:
    """One batched launch on this rank's GPU (no CPU path: raises without a gfx950 device).  `device` defaults to the calling
    rank's current HIP device (torch.cuda.current_device(), i.e. LOCAL_RANK after torch.cuda.set_device)."""
    from . import scheduler
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    flats = [scheduler.FlatProblem(p) for p in problems]
    try:
        for f in flats:
            f.upload(device)
        res, _, _ = scheduler.solve_batch(flats)
        return res
    finally:
        for f in flats:
            f.close()


# =====================================================================================================
# Consolidation decisions on top of the batched what-ifs.
#
# The reference runs ONE simulateScheduling per probe: ceil(log2 N) dependent Solves for multi-node
# consolidation's binary search (multinodeconsolidation.go:74-114), up to N sequential ones for single-node
# consolidation's first-success scan (singlenodeconsolidation.go:43-78).  Here every prefix / every singleton
# is solved in ONE ks_solve_batch launch, the price stage (filterByPrice over worstLaunchPrice,
# helpers.go:148-157,292-315) runs on the device over the results while they are still there, and the
# reference's control flow is replayed over the per-what-if commands.  Validation (validation.go: TTL wait,
# re-simulation) and events need the live cluster and stay in Go.
# =====================================================================================================
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

from .model import (InstanceType, LABEL_CAPACITY_TYPE, LABEL_INITIALIZED, LABEL_INSTANCE_TYPE, LABEL_ZONE, Provisioner,
                    RequirementOut, StateNode)

ACTION_DO_NOTHING, ACTION_DELETE, ACTION_REPLACE = "do-nothing", "delete", "replace"   # deprovisioning/types.go actions
CAPACITY_TYPE_SPOT, CAPACITY_TYPE_ON_DEMAND = "spot", "on-demand"


@dataclass
class Snapshot:
    """What candidateNodes / simulateScheduling read from the cluster (helpers.go:42-115, 165-230)."""
    instance_types: List[InstanceType]
    provisioner: Provisioner
    nodes: List[StateNode]
    bound: List[list]                      # pods bound to each node
    pending: list = field(default_factory=list)      # provisioner.GetPendingPods (helpers.go:76-79): they join the batch of every simulation
    deleting: tuple = ()                   # indices of nodes that are MarkedForDeletion (helpers.go:48-61): not state nodes, their pods join the batch


@dataclass
class CandidateNode:
    """deprovisioning CandidateNode (types.go): the node plus what pricing reads."""
    index: int
    name: str
    instance_type: str
    capacity_type: str
    zone: str


@dataclass
class Command:
    """deprovisioning Command as far as the decision goes: action, nodesToRemove, the replacement node's options."""
    action: str = ACTION_DO_NOTHING
    nodes_to_remove: List[str] = field(default_factory=list)
    replacement_types: List[str] = field(default_factory=list)                 # replacementNodes[0].InstanceTypeOptions
    replacement_requirements: Dict[str, tuple] = field(default_factory=dict)  # ... .Requirements (canonical tuples)
    error: Optional[str] = None     # computeConsolidation returned an error for this candidate set (consolidation.go:224-228)

    def canonical(self):
        return (self.action, tuple(self.nodes_to_remove), tuple(self.replacement_types), tuple(sorted(self.replacement_requirements.items())))


class _NoProblem:
    def close(self):
        pass


def candidate(snapshot: Snapshot, i: int) -> CandidateNode:
    lab = snapshot.nodes[i].labels
    return CandidateNode(i, snapshot.nodes[i].name, lab[LABEL_INSTANCE_TYPE], lab[LABEL_CAPACITY_TYPE], lab[LABEL_ZONE])


def _offering_price(it: InstanceType, capacity_type: str, zone: str) -> Optional[float]:
    for o in it.offerings:                 # Offerings.Get, cloudprovider/types.go:117-124 (availability is not consulted)
        if o.capacity_type == capacity_type and o.zone == zone:
            return o.price
    return None


def get_node_prices(types: Dict[str, InstanceType], cands: Sequence[CandidateNode]) -> float:
    """getNodePrices, consolidation.go:277-287."""
    price = 0.0
    for c in cands:
        p = _offering_price(types[c.instance_type], c.capacity_type, c.zone)
        if p is None:
            raise ValueError(f"unable to determine offering for {c.instance_type}/{c.capacity_type}/{c.zone}")
        price += p
    return price


def _has(r: Optional[RequirementOut], value: str) -> bool:
    """Requirements.Get(key).Has(value): a missing key reads as Exists (requirements.go:114-120, requirement.go:171-176)."""
    if r is None:
        return True
    inside = value in r.values
    if (inside if r.complement else not inside):
        return False
    if r.greater_than is None and r.less_than is None:
        return True
    try:
        v = int(value)
    except ValueError:
        return False
    return not ((r.greater_than is not None and r.greater_than >= v) or (r.less_than is not None and r.less_than <= v))


def _req_tuple(r: RequirementOut) -> tuple:
    return (r.complement, tuple(sorted(r.values)), r.greater_than, r.less_than)


def compute_consolidations(snapshot: Snapshot, candidate_sets: Sequence[Sequence[int]]) -> Tuple[List[Command], list, list]:
    """computeConsolidation (consolidation.go:190-274) for every candidate set at once: one batched Solve, one batched
    price stage.  Returns (commands, flat problems, results); the caller closes the flat problems."""
    from . import scheduler, workloads
    types = {it.name: it for it in snapshot.instance_types}
    tindex = {it.name: i for i, it in enumerate(snapshot.instance_types)}
    # one snapshot, flattened natively per what-if on all host cores (scheduler.open_whatifs)
    # simulateScheduling's batch is pending pods + the candidates' pods + the pods of nodes that are already deleting (helpers.go:76-84), and
    # deleting nodes are no state nodes (:48-55).  Over the shared snapshot: the pending pods sit on a node of their own that no provisioner owns
    # (it is never an existing node) and that every candidate set removes first; the deleting nodes are removed last.
    PENDING = "~pending~"
    deleting = [int(j) for j in snapshot.deleting]
    nodes, bound = list(snapshot.nodes), list(snapshot.bound)
    pend_idx = None
    if snapshot.pending:
        pend_idx = len(nodes)
        nodes.append(StateNode(name=PENDING)); bound.append(list(snapshot.pending))
    snap, pod_node = workloads.snapshot_problem(snapshot.instance_types, snapshot.provisioner, nodes, bound)
    snap.cluster_pods = [cp for cp in snap.cluster_pods if cp.node_name != PENDING]      # pending pods are bound nowhere: countDomains does not see them
    cmds = [Command() for _ in candidate_sets]
    live = [i for i, cs in enumerate(candidate_sets) if not (set(cs) & set(deleting))]
    for i in range(len(candidate_sets)):
        if i not in set(live):
            cmds[i] = Command(error="candidate node is deleting")       # errCandidateNodeDeleting, helpers.go:62-67
    sets = [([pend_idx] if pend_idx is not None else []) + list(candidate_sets[i]) + deleting for i in live]
    flats_live = scheduler.open_whatifs(snap, pod_node, sets)
    results_live, _, _ = scheduler.solve_batch(flats_live) if flats_live else ([], 0, 0)
    flats = [_NoProblem()] * len(candidate_sets); results = [None] * len(candidate_sets)       # (a refused candidate set has nothing on the device)
    for i, f, r in zip(live, flats_live, results_live):
        flats[i], results[i] = f, r
    need, prices = [], []
    for i, (cs, res) in enumerate(zip(candidate_sets, results)):
        if res is None:
            continue
        cands = [candidate(snapshot, j) for j in cs]
        # simulateScheduling, helpers.go:102-111: the simulation must not lean on a node that is not ready yet -- Solve returns EVERY
        # in-state existing node (scheduler.go:132), so one uninitialised node that stays in the cluster fails the simulation
        removed = set(cs) | set(deleting)
        if any(n.owned and n.labels.get(LABEL_INITIALIZED) != "true" for j, n in enumerate(snapshot.nodes) if j not in removed and n.in_state):
            continue
        if res.unscheduled:                                     # "not all pods would schedule"
            continue
        if not res.new_nodes:                                   # everything fits on the remaining nodes
            cmds[i] = Command(ACTION_DELETE, [c.name for c in cands])
            continue
        if len(res.new_nodes) != 1:                             # "we're not going to turn a single node into multiple nodes"
            continue
        try:
            price = get_node_prices(types, cands)
        except ValueError as e:                                 # "getting offering price from candidate node": an error of THIS computation only
            cmds[i] = Command(error=str(e))
            continue
        need.append(i)
        prices.append(price)
    kept = scheduler.price_filter([flats[i] for i in need], [0] * len(need), prices)
    for i, keep in zip(need, kept):
        res = results[i]
        node = res.new_nodes[0]
        keep = set(keep)
        options = [n for n in node.instance_types if tindex[n] in keep]   # filterByPrice preserves the option order
        if not options:                                         # "can't replace with a cheaper node"
            continue
        cands = [candidate(snapshot, j) for j in candidate_sets[i]]
        ct = node.requirements.get(LABEL_CAPACITY_TYPE)
        if all(c.capacity_type == CAPACITY_TYPE_SPOT for c in cands) and _has(ct, CAPACITY_TYPE_SPOT):
            continue                                            # "can't replace a spot node with a spot node"
        reqs = {k: _req_tuple(r) for k, r in node.requirements.items()}
        if _has(ct, CAPACITY_TYPE_SPOT) and _has(ct, CAPACITY_TYPE_ON_DEMAND):
            # OD -> [OD, spot] was priced on the spot assumption: pin the launch to spot (Requirements.Add = intersection)
            reqs[LABEL_CAPACITY_TYPE] = (False, (CAPACITY_TYPE_SPOT,), None, None)
        cmds[i] = Command(ACTION_REPLACE, [c.name for c in cands], options, reqs)
    return cmds, flats, results


def simulate_candidates(snapshot: Snapshot, candidate_sets: Sequence[Sequence[int]]) -> List[Optional[SolveResult]]:
    """simulateScheduling (helpers.go:42-99) for every candidate set in ONE batched launch over the shared snapshot; None where the reference returns
    errCandidateNodeDeleting."""
    from . import scheduler, workloads
    PENDING = "~pending~"
    deleting = [int(j) for j in snapshot.deleting]
    nodes, bound = list(snapshot.nodes), list(snapshot.bound)
    pend_idx = None
    if snapshot.pending:
        pend_idx = len(nodes)
        nodes.append(StateNode(name=PENDING)); bound.append(list(snapshot.pending))
    snap, pod_node = workloads.snapshot_problem(snapshot.instance_types, snapshot.provisioner, nodes, bound)
    snap.cluster_pods = [cp for cp in snap.cluster_pods if cp.node_name != PENDING]
    live = [i for i, cs in enumerate(candidate_sets) if not (set(cs) & set(deleting))]
    sets = [([pend_idx] if pend_idx is not None else []) + list(candidate_sets[i]) + deleting for i in live]
    flats = scheduler.open_whatifs(snap, pod_node, sets) if sets else []
    got, _, _ = scheduler.solve_batch(flats) if flats else ([], 0, 0)
    for f in flats:
        f.close()
    out: List[Optional[SolveResult]] = [None] * len(candidate_sets)
    for i, r in zip(live, got):
        out[i] = r
    return out


def replacement_command(snapshot: Snapshot, candidates: Sequence[int], simulate: Optional[Callable] = None):
    """Drift.ComputeCommand / Expiration.ComputeCommand (drift.go:59-98, expiration.go:68-113) after their candidate filters and sort.  The reference
    simulates candidate after candidate and stops at the first whose simulation runs; here every candidate is simulated in ONE batch (`simulate`, default
    `simulate_candidates`) and the same first candidate decides: delete if its pods fit the rest of the cluster, otherwise replace it with EVERY node the
    simulation opened (no price stage, any number of nodes; pods left unscheduled are only logged there).
    -> (action, [node name], [(instance type options, canonical requirements)] per replacement node)"""
    results = (simulate or simulate_candidates)(snapshot, [[i] for i in candidates])
    for i, res in zip(candidates, results):
        if res is None:                                        # errCandidateNodeDeleting: "just retry" with the next candidate
            continue
        name = snapshot.nodes[i].name
        if not res.new_nodes:
            return (ACTION_DELETE, [name], [])
        return (ACTION_REPLACE, [name], [(list(n.instance_types), tuple(sorted((k, _req_tuple(r)) for k, r in n.requirements.items()))) for n in res.new_nodes])
    return (ACTION_DO_NOTHING, [], [])


def validate_command(snapshot: Snapshot, cmd: Command, candidates: Sequence[int], simulate: Optional[Callable] = None) -> bool:
    """Validation.ValidateCommand (validation.go:109-172) on the cluster as it is NOW (the TTL wait before it stays with the caller): the command's nodes
    that are still candidates are simulated again -- valid iff every pod schedules and the simulation needs no new node where none was expected, or exactly
    one whose instance type options contain the command's (instanceTypesAreSubset: the simulation applies no price filter, so it may list more)."""
    names = {snapshot.nodes[i].name: i for i in candidates}
    idx = [names[n] for n in cmd.nodes_to_remove if n in names]
    if not idx:
        return False
    (res,) = (simulate or simulate_candidates)(snapshot, [idx])
    if res is None:
        raise ValueError("candidate node is deleting")          # "simulating scheduling, %w"
    removed = set(idx) | set(int(j) for j in snapshot.deleting)
    if any(n.owned and n.labels.get(LABEL_INITIALIZED) != "true" for j, n in enumerate(snapshot.nodes) if j not in removed and n.in_state):
        return False                                             # helpers.go:102-111: allPodsScheduled = false
    if res.unscheduled:
        return False
    if not res.new_nodes:
        return not cmd.replacement_types
    if len(res.new_nodes) > 1 or not cmd.replacement_types:
        return False
    return set(cmd.replacement_types) <= set(res.new_nodes[0].instance_types)


def _filter_out_same_type(snapshot: Snapshot, flat, result, cmd: Command, cands: Sequence[CandidateNode]) -> List[str]:
    """filterOutSameType, multinodeconsolidation.go:132-165 (the second filterByPrice runs on the device as well)."""
    from . import scheduler
    types = {it.name: it for it in snapshot.instance_types}
    tindex = {it.name: i for i, it in enumerate(snapshot.instance_types)}
    existing, by_type = set(), {}
    for c in cands:
        existing.add(c.instance_type)
        p = _offering_price(types[c.instance_type], c.capacity_type, c.zone)
        if p is None:
            continue
        if p < by_type.get(c.instance_type, float("inf")) or c.instance_type not in by_type:
            by_type[c.instance_type] = min(p, by_type.get(c.instance_type, 1.7976931348623157e308))
    max_price = 1.7976931348623157e308
    for n in cmd.replacement_types:
        if n in existing and by_type.get(n, 0.0) < max_price:   # a Go map miss reads 0.0
            max_price = by_type.get(n, 0.0)
    # computeConsolidation may have narrowed the replacement's capacity-type to spot after pricing; the device still holds the
    # Solve's requirements, so the second pricing is told about the narrowing
    narrowed = cmd.replacement_requirements.get(LABEL_CAPACITY_TYPE) == (False, (CAPACITY_TYPE_SPOT,), None, None)   # idempotent if it already was
    keep = set(scheduler.price_filter([flat], [0], [max_price], [narrowed])[0])
    return [n for n in cmd.replacement_types if tindex[n] in keep]


def first_n_node_consolidation_option(snapshot: Snapshot, candidates: Sequence[int], max_nodes: int = 100) -> Command:
    """firstNNodeConsolidationOption, multinodeconsolidation.go:74-114: every prefix the binary search could probe is
    solved in one launch, then the search is replayed over the commands."""
    if len(candidates) < 2:
        return Command()
    lo, hi = 1, max_nodes
    if len(candidates) <= hi:
        hi = len(candidates) - 1
    prefixes = [list(candidates[0:mid + 1]) for mid in range(lo, hi + 1)]
    cmds, flats, results = compute_consolidations(snapshot, prefixes)
    try:
        last = Command()
        while lo <= hi:
            mid = (lo + hi) // 2
            k = mid - 1
            action = cmds[k]
            if action.error is not None:                        # firstNNodeConsolidationOption returns the error of the prefix it probes (:92-95)
                raise ValueError(action.error)
            if action.action == ACTION_REPLACE:
                cands = [candidate(snapshot, j) for j in prefixes[k]]
                opts = _filter_out_same_type(snapshot, flats[k], results[k], action, cands)
                action = Command(ACTION_REPLACE if opts else ACTION_DO_NOTHING, action.nodes_to_remove if opts else [], opts,
                                 action.replacement_requirements if opts else {})
            if action.action in (ACTION_REPLACE, ACTION_DELETE):
                last = action
                lo = mid + 1
            else:
                hi = mid - 1
        return last
    finally:
        for f in flats:
            f.close()


def single_node_consolidation_option(snapshot: Snapshot, candidates: Sequence[int]) -> Command:
    """SingleNodeConsolidation.ComputeCommand's scan, singlenodeconsolidation.go:54-78 (validation stays in Go): every
    singleton in one launch, first replace/delete in candidate order wins."""
    cmds, flats, _ = compute_consolidations(snapshot, [[c] for c in candidates])
    for f in flats:
        f.close()
    for cmd in cmds:
        if cmd.error is not None:                               # logged, next candidate (singlenodeconsolidation.go:57-60)
            continue
        if cmd.action in (ACTION_REPLACE, ACTION_DELETE):
            return cmd
    return Command()
