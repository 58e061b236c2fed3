"""Multi-GPU sharding of the hot path: independent LP relaxations (branch-and-bound nodes) over ranks.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the MI355X node, "gloo" in the CPU
tests).  A node's relaxation is a pure function of (saved root tableau, cut list) (branch-and-cut.ts:33-37
always restores the root), so the unit of sharding is the node: every rank owns an engine holding the SAME
saved root -- each rank solves the root itself, which is cheaper than broadcasting a 7-23 MB tableau and keeps
the data path free of collectives -- and evaluates nodes rank, rank+world, ... of a batch.  The only exchange
step is the all-gather of the per-node outcomes (flags, evaluation, RHS column, row map: ~12 bytes per row per
node); afterwards every rank holds every outcome and replays the same deterministic tree, so the incumbent is
agreed on without a separate broadcast.
"""
import numpy as np

from ._capi import SimplexResult
from .branch_and_cut import _NodeEval

_FIELDS = [name for name, _ in SimplexResult._fields_]


def shard(items, rank, world):
    """round-robin: rank r takes items r, r + world, ..."""
    return items[rank::world]


def _pack(results, rhs, vibr, stride):
    """[n, len(_FIELDS) + 2 * stride] float64: result fields, RHS row, row map (exact in a double)"""
    n = len(results)
    out = np.zeros((n, len(_FIELDS) + 2 * stride), dtype=np.float64)
    for i, r in enumerate(results):
        out[i, :len(_FIELDS)] = [float(getattr(r, f)) for f in _FIELDS]
        h = r.height
        out[i, len(_FIELDS):len(_FIELDS) + h] = rhs[i, :h]
        out[i, len(_FIELDS) + stride:len(_FIELDS) + stride + h] = vibr[i, :h]
    return out


def _unpack(row, stride):
    r = SimplexResult()
    for j, (name, ctype) in enumerate(SimplexResult._fields_):
        v = row[j]
        setattr(r, name, float(v) if name in ("obj_cell", "evaluation") else int(v))
    h = r.height
    rhs = row[len(_FIELDS):len(_FIELDS) + h].copy()
    vibr = row[len(_FIELDS) + stride:len(_FIELDS) + stride + h].astype(np.int32)
    return _NodeEval(r, rhs, vibr)


def evaluate_nodes_sharded(tableau, cut_lists, check_cycles, group):
    """Every rank calls this with the same `cut_lists`; returns the outcomes of ALL nodes on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(cut_lists)
    stride = tableau.row_capacity
    per = (n + world - 1) // world  # equal-sized contributions (padded) for all_gather
    mine = shard(cut_lists, rank, world)
    width = len(_FIELDS) + 2 * stride
    local = np.zeros((per, width), dtype=np.float64)
    if mine:
        results, rhs, vibr = tableau.applyCutsBatch(mine, check_cycles=check_cycles)
        local[:len(mine)] = _pack(results, rhs, vibr, stride)
    use_cuda = dist.get_backend(group) == "nccl"
    t = torch.from_numpy(local)
    if use_cuda:
        t = t.cuda()
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)  # the one exchange step of the sharded path
    parts = [g.cpu().numpy() for g in gathered]
    out = []
    for i in range(n):
        out.append(_unpack(parts[i % world][i // world], stride))
    return out


def make_sharded_evaluator(tableau, check_cycles, group):
    return lambda cut_lists: evaluate_nodes_sharded(tableau, cut_lists, check_cycles, group)
