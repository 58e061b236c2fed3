"""Multi-GPU sharding of the hot path across PROCESSES: independent LP relaxations (branch-and-bound nodes) over ranks.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the MI355X node, "gloo" in the CPU tests).
A node's relaxation is a pure function of (saved root tableau, cut list) (branch-and-cut.ts:33-37 always restores the
root), so the unit of sharding is the node: every rank owns an engine holding the SAME saved root and evaluates nodes
rank, rank + world, ... of a batch with one engine call.  The one exchange step is the all-gather of the outcomes --
per node the state record (flags, evaluation = the bound the tree prunes with, pivot counts) and, round 5, only what the
tree reads between relaxations: the row and the RHS cell of every integer variable (mip-utils.ts:43-61, 100-126; 12 bytes
per integer variable -- `evaluate_nodes_sharded_watched`; the whole RHS column + row map, ~12 bytes per ROW, remain
available: `evaluate_nodes_sharded`) -- after which every rank holds every outcome and replays the same deterministic
tree (speculation with in-order commit), so the incumbent is agreed on without a further broadcast; the full column of the
leaf the tree commits to is re-evaluated locally at the end.

(Inside ONE process the same split over several GPUs is the engine's own device pool, jslp_pool_* in
include/jslp_engine.h: host threads and peer copies instead of ranks and collectives.)

The payload is assembled and taken apart with whole-array copies: the engine call leaves results, RHS columns and row
maps in three contiguous arrays (jslp_engine_relax_batch_pinned), which are laid side by side into one byte matrix
[node, RES | rhs | rows]; nothing here loops over nodes.
"""
import ctypes

import numpy as np

from ._capi import SimplexResult
from .branch_and_cut import _NodeEval

RES_BYTES = ctypes.sizeof(SimplexResult)


def shard(items, rank, world):
    """round-robin: rank r takes items r, r + world, ..."""
    return items[rank::world]


def node_bytes(stride):
    """bytes of one node's outcome in the exchange payload (8-byte aligned)"""
    return RES_BYTES + 8 * stride + 4 * stride + (4 * stride) % 8


class ShardedOutcomes:
    """every node's outcome after the all-gather; node i sits in rank (i % world)'s block at slot i // world"""

    def __init__(self, blocks, n, world, stride):
        self.blocks, self.n, self.world, self.stride = blocks, n, world, stride  # blocks: [world, per, node_bytes] uint8

    def __len__(self):
        return self.n

    def _row(self, i):
        return self.blocks[i % self.world, i // self.world]

    def result(self, i):
        return SimplexResult.from_buffer_copy(self._row(i)[:RES_BYTES].tobytes())

    def rhs(self, i, height):
        return self._row(i)[RES_BYTES:RES_BYTES + 8 * height].view(np.float64)

    def rows(self, i, height):
        o = RES_BYTES + 8 * self.stride
        return self._row(i)[o:o + 4 * height].view(np.int32)

    def node(self, i):
        r = self.result(i)
        return _NodeEval(r, self.rhs(i, r.height).copy(), self.rows(i, r.height).copy())

    def heights(self):
        """height of every node, vectorised (offset of `height` inside the result struct)"""
        off = SimplexResult.height.offset
        h = self.blocks[:, :, off:off + 4].copy().view(np.int32)[:, :, 0]  # [world, per]
        idx = np.arange(self.n)
        return h[idx % self.world, idx // self.world]


EXCHANGE_STATS = {"seconds": 0.0, "calls": 0, "bytes": 0}  # the exchange step since the caller last zeroed it (bench.py)


def exchange_outcomes(local, n, group):
    """all-gather of the per-rank payloads ([per, node_bytes] uint8, padded to equal size) -> ShardedOutcomes"""
    import time
    t0 = time.perf_counter()
    try:
        return _exchange_outcomes(local, n, group)
    finally:
        EXCHANGE_STATS["seconds"] += time.perf_counter() - t0
        EXCHANGE_STATS["calls"] += 1
        EXCHANGE_STATS["bytes"] += int(local.size)


def _exchange_outcomes(local, n, group):
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per, nb = local.shape
    t = torch.from_numpy(local)
    if dist.get_backend(group) == "nccl":  # RCCL moves device memory: one H2D, the collective, one D2H
        t = t.cuda(non_blocking=True)
    out = torch.empty((world, per, nb), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t.view(-1), group=group)
    return out.cpu().numpy() if out.is_cuda else out.numpy()


def pack_local(results, rhs, rows, n_mine, per, stride):
    """[per, node_bytes] uint8 from the engine's three output arrays (whole-array copies)"""
    nb = node_bytes(stride)
    local = np.zeros((per, nb), dtype=np.uint8)
    if n_mine:
        res = np.frombuffer(results, dtype=np.uint8, count=n_mine * RES_BYTES).reshape(n_mine, RES_BYTES)
        local[:n_mine, :RES_BYTES] = res
        # (the engine's own read-back buffers may be laid out with a wider, aligned row stride: keep the first `stride` entries)
        local[:n_mine, RES_BYTES:RES_BYTES + 8 * stride] = np.ascontiguousarray(rhs[:n_mine, :stride]).view(np.uint8).reshape(n_mine, 8 * stride)
        o = RES_BYTES + 8 * stride
        local[:n_mine, o:o + 4 * stride] = np.ascontiguousarray(rows[:n_mine, :stride]).view(np.uint8).reshape(n_mine, 4 * stride)
    return local


class ShardedOutcomesDevice:
    """every node's outcome after the all-gather of the DEVICE-resident payloads: rank r's block is one byte vector
    [per state records | per x stride doubles | per x stride int32], node i sits in rank (i % world)'s block at slot i // world"""

    def __init__(self, blocks, n, world, per, stride, rec, tableau):
        self.blocks, self.n, self.world, self.per, self.stride, self.rec = blocks, n, world, per, stride, rec  # blocks: [world, block_bytes] uint8
        self.o_rhs = _align(per * rec)
        self.o_rows = _align(self.o_rhs + per * stride * 8)
        # result structs of ALL nodes from the raw records, rank by rank (one library call per rank)
        self.results = []
        for r in range(world):
            k = len(range(r, n, world))
            self.results.append(tableau.results_from_states(blocks[r, :k * rec], k) if k else None)

    def __len__(self):
        return self.n

    def result(self, i):
        return self.results[i % self.world][i // self.world]

    def rhs(self, i, height):
        o = self.o_rhs + (i // self.world) * self.stride * 8
        return self.blocks[i % self.world, o:o + 8 * height].view(np.float64)

    def rows(self, i, height):
        o = self.o_rows + (i // self.world) * self.stride * 4
        return self.blocks[i % self.world, o:o + 4 * height].view(np.int32)

    def node(self, i):
        r = self.result(i)
        return _NodeEval(r, self.rhs(i, r.height).copy(), self.rows(i, r.height).copy())

    def heights(self):
        return np.array([self.result(i).height for i in range(self.n)], dtype=np.int32)


def _align(x, a=64):
    return (x + a - 1) // a * a


def evaluate_nodes_sharded_device(tableau, cut_lists, check_cycles, group, packed_mine=None):
    """evaluate_nodes_sharded without the host bounce: the engine leaves this rank's outcomes in a device tensor
    (jslp_engine_relax_batch_device) that IS the all-gather's input; the gathered block crosses PCIe once.  With the gloo backend
    (CPU tests: the oracle library's "device" memory is host memory) the same code runs on CPU tensors."""
    import time
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(cut_lists)
    stride = (tableau.row_capacity + 3) // 4 * 4  # 16-byte aligned node slices: the kernels' fast stores
    rec = tableau.state_record_bytes()
    per = max((n + world - 1) // world, 1)
    o_rhs = _align(per * rec)
    o_rows = _align(o_rhs + per * stride * 8)
    block = _align(o_rows + per * stride * 4)
    # the engine writes through the pointers: device memory for the HIP engine whatever the process group's backend is (the
    # test library's "device" memory is host memory)
    on_gpu = tableau.lib.backend.startswith("hip")
    local = torch.zeros(block, dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
    if on_gpu:
        torch.cuda.current_stream().synchronize()  # the fill runs on torch's stream, the engine's kernels on the engine's own
    n_mine = len(range(rank, n, world))
    if n_mine:
        packed = packed_mine if packed_mine is not None else tableau.pack_cut_lists(shard(cut_lists, rank, world))
        base = local.data_ptr()
        tableau.applyCutsBatchDevice(packed, check_cycles, base, base + o_rhs, base + o_rows, stride)
    t0 = time.perf_counter()
    if on_gpu and dist.get_backend(group) != "nccl":  # HIP engines under a CPU process group (N virtual shards on one GPU)
        local = local.cpu()
    out = torch.empty((world, block), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local, group=group)
    blocks = out.cpu().numpy() if out.is_cuda else out.numpy()
    EXCHANGE_STATS["seconds"] += time.perf_counter() - t0
    EXCHANGE_STATS["calls"] += 1
    EXCHANGE_STATS["bytes"] += int(block)
    return ShardedOutcomesDevice(blocks, n, world, per, stride, rec, tableau)


class ShardedOutcomesWatched:
    """every node's COMPACT outcome after the all-gather of the device-resident payloads: rank r's block is one byte vector
    [per state records | per x w int32 rows | per x w doubles], node i sits in rank (i % world)'s block at slot i // world; w = the watched
    (integer) variables.  What the tree reads between relaxations and nothing else: 128 + 12 w bytes per node (Monster_II: 1.5 KB instead
    of the 11.4 KB of the full RHS column + row map)."""

    def __init__(self, blocks, n, world, per, w, rec, tableau):
        self.blocks, self.n, self.world, self.per, self.w, self.rec = blocks, n, world, per, w, rec  # blocks: [world, block_bytes] uint8
        self.o_rows = _align(per * rec)
        self.o_vals = _align(self.o_rows + per * w * 4)
        self.results = []
        for r in range(world):
            k = len(range(r, n, world))
            self.results.append(tableau.results_from_states(blocks[r, :k * rec], k) if k else None)

    def __len__(self):
        return self.n

    def result(self, i):
        return self.results[i % self.world][i // self.world]

    def watched_rows(self, i):
        o = self.o_rows + (i // self.world) * self.w * 4
        return self.blocks[i % self.world, o:o + 4 * self.w].view(np.int32)

    def watched_values(self, i):
        o = self.o_vals + (i // self.world) * self.w * 8
        return self.blocks[i % self.world, o:o + 8 * self.w].view(np.float64)

    def node(self, i):
        from .branch_and_cut import _NodeEvalWatched
        return _NodeEvalWatched(self.result(i), self.watched_rows(i).copy(), self.watched_values(i).copy())

    def heights(self):
        return np.array([self.result(i).height for i in range(self.n)], dtype=np.int32)


_BUFS = {}  # (kind, device, bytes) -> tensor: the exchange's buffers are reused from call to call (round 6)


def _exchange_buffers(block, world, on_gpu, gather_on_gpu):
    """this rank's payload (device memory the engine writes through: never zeroed again -- every byte a reader looks at is written by the engine
    call of the same exchange, the alignment gaps are never read), the gathered block, and -- when the gather runs on the device -- a PINNED host
    tensor the gathered block lands in with one DMA (`out.cpu()` into pageable memory staged 3.5 MB through a bounce buffer: 0.33-0.40 ms per
    exchange of the 2416-node batch, and noisy; pinned: ~0.15 ms)"""
    import torch
    dev = "cuda" if on_gpu else "cpu"
    key = ("local", dev, block)
    local = _BUFS.get(key)
    if local is None:
        local = _BUFS[key] = torch.zeros(block, dtype=torch.uint8, device=dev)
        if on_gpu:
            torch.cuda.current_stream().synchronize()  # the fill runs on torch's stream, the engine's kernels on the engine's own
    gdev = "cuda" if gather_on_gpu else "cpu"
    key = ("out", gdev, world * block)
    out = _BUFS.get(key)
    if out is None:
        out = _BUFS[key] = torch.empty((world, block), dtype=torch.uint8, device=gdev)
    host = None
    if gather_on_gpu:
        key = ("host", "pinned", world * block)
        host = _BUFS.get(key)
        if host is None:
            host = _BUFS[key] = torch.empty((world, block), dtype=torch.uint8, pin_memory=True)
    return local, out, host


def watched_block_bytes(per, w, rec):
    """bytes one rank contributes to the compact exchange for `per` nodes and `w` watched variables"""
    return _align(_align(_align(per * rec) + per * w * 4) + per * w * 8)


def evaluate_nodes_sharded_watched(tableau, cut_lists, check_cycles, group, packed_mine=None, copy=True):
    """The compact exchange: the engine leaves, per node, the state record and the row / RHS cell of the watched
    variables in a device tensor (jslp_engine_relax_batch_watched_device) that IS the all-gather's input.  The tableau's watched
    variables must have been set (Tableau.set_watched_variables(model.integer_index_array)) -- identically on every rank."""
    import time
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(cut_lists)
    w = tableau.watched_count()
    if w <= 0:
        raise ValueError("evaluate_nodes_sharded_watched: Tableau.set_watched_variables first")
    rec = tableau.state_record_bytes()
    per = max((n + world - 1) // world, 1)
    o_rows = _align(per * rec)
    o_vals = _align(o_rows + per * w * 4)
    block = _align(o_vals + per * w * 8)
    on_gpu = tableau.lib.backend.startswith("hip")
    gather_on_gpu = on_gpu and dist.get_backend(group) == "nccl"
    local, out, host = _exchange_buffers(block, world, on_gpu, gather_on_gpu)
    n_mine = len(range(rank, n, world))
    if n_mine:
        packed = packed_mine if packed_mine is not None else tableau.pack_cut_lists(shard(cut_lists, rank, world))
        base = local.data_ptr()
        tableau.applyCutsBatchWatchedDevice(packed, check_cycles, base, base + o_rows, base + o_vals)
    t0 = time.perf_counter()
    send = local
    if on_gpu and not gather_on_gpu:  # HIP engines under a CPU process group (N virtual shards on one GPU)
        send = local.cpu()
    dist.all_gather_into_tensor(out.view(-1), send, group=group)
    if gather_on_gpu:
        host.copy_(out, non_blocking=True)  # ONE DMA into pinned memory
        torch.cuda.current_stream().synchronize()
        blocks = host.numpy()
    else:
        blocks = out.numpy()
    if copy:
        blocks = blocks.copy()  # (the buffers are reused by the next exchange; copy=False: the outcomes are VIEWS, valid until this process's next exchange)
    EXCHANGE_STATS["seconds"] += time.perf_counter() - t0
    EXCHANGE_STATS["calls"] += 1
    EXCHANGE_STATS["bytes"] += int(block)
    return ShardedOutcomesWatched(blocks, n, world, per, w, rec, tableau)


def evaluate_nodes_sharded(tableau, cut_lists, check_cycles, group, packed_mine=None):
    """Every rank calls this with the same `cut_lists`; returns the outcomes of ALL nodes on every rank
    (ShardedOutcomes).  `packed_mine`: this rank's share already flattened by Tableau.pack_cut_lists."""
    import os
    import torch.distributed as dist

    # RCCL: outcomes stay on the device until they have been gathered (JSLP_SHARD_DEVICE_PATH=0 / 1 forces the choice: the
    # CPU tests run both forms over gloo)
    forced = os.environ.get("JSLP_SHARD_DEVICE_PATH")
    if forced == "1" or (forced is None and dist.get_backend(group) == "nccl"):
        return evaluate_nodes_sharded_device(tableau, cut_lists, check_cycles, group, packed_mine)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(cut_lists)
    stride = tableau.row_capacity
    per = (n + world - 1) // world  # equal-sized contributions (padded) for the all-gather
    n_mine = len(range(rank, n, world))
    results = rhs = rows = None
    if n_mine:
        packed = packed_mine if packed_mine is not None else tableau.pack_cut_lists(shard(cut_lists, rank, world))
        results, rhs, rows = tableau.applyCutsBatch(None, check_cycles=check_cycles, packed=packed, copy=False)
    local = pack_local(results, rhs, rows, n_mine, per, stride)
    blocks = exchange_outcomes(local, n, group)  # the one exchange step of the sharded path
    return ShardedOutcomes(blocks, n, world, stride)


def make_sharded_evaluator(tableau, check_cycles, group, watched=None):
    """evaluate(cut_lists) for branch_and_cut(evaluate_batch=...).  `watched` = the variable indexes the tree reads between
    relaxations (the model's integer variables): the COMPACT exchange -- registered on the engine here, once (every rank passes the
    same list); None = whole RHS columns + row maps (JSLP_SHARD_COMPACT=0 forces that form)."""
    import os
    # (the engine takes at most row_capacity watched variables; a model with more integers than that exchanges whole columns)
    if watched is not None and os.environ.get("JSLP_SHARD_COMPACT", "1") != "0" and 0 < len(watched) <= tableau.row_capacity:
        tableau.set_watched_variables(watched)

        def evaluate_compact(cut_lists):
            out = evaluate_nodes_sharded_watched(tableau, cut_lists, check_cycles, group)
            return [out.node(i) for i in range(len(out))]
        return evaluate_compact

    def evaluate(cut_lists):
        out = evaluate_nodes_sharded(tableau, cut_lists, check_cycles, group)
        return [out.node(i) for i in range(len(out))]  # a speculative batch: a handful of nodes
    return evaluate
