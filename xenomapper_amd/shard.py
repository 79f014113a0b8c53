"""Read-block sharding across GPUs (one process per GPU, torch.distributed; "nccl" = RCCL on ROCm).

The classification path shards by contiguous record blocks: a unit at record i needs
only records i-1 and i of both files (ref xenomapper.py:402-418), so rank g takes records
[start_g, end_g) plus a one-record halo (start_g - 1) and owns every unit whose *second* record lies in
its block -- three equal QNAMEs in a row still give two overlapping pairs, whichever way the cut falls.
The only collective on the data path is one all-reduce (sum) of category_counts: 64 x int64 = 512 bytes,
latency-bound on xGMI.  Bin index lists are never exchanged: each rank keeps (or returns) its six stable
lists and shard order reproduces the reference's output order.
"""
from __future__ import annotations

import numpy as np


def plan_blocks(n_records, world):
    """Contiguous, 64-record-aligned blocks [(start, end)] covering [0, n_records); sizes differ by <= 64."""
    words = (n_records + 63) // 64
    per, extra = divmod(words, world)
    out, w0 = [], 0
    for g in range(world):
        w1 = w0 + per + (1 if g < extra else 0)
        out.append((min(w0 * 64, n_records), min(w1 * 64, n_records)))
        w0 = w1
    return out


def unpack_unit_bits(unit_bits, n_records):
    return np.unpackbits(np.ascontiguousarray(unit_bits, dtype=np.uint64).view(np.uint8), bitorder="little")[:n_records]


def pack_unit_bits(flags):
    flags = np.asarray(flags, dtype=np.uint8)
    pad = (-flags.shape[0]) % 64
    return np.packbits(np.concatenate([flags, np.zeros(pad, dtype=np.uint8)]), bitorder="little").view(np.uint64)


def take_block(columns, unit_bits, n_records, start, end):
    """Local view of block [start, end): the columns with the halo record in front (when start > 0) and the
    local unit mask (the halo closes no unit here: its unit belongs to the previous block).
    Returns (local_columns, local_unit_bits, halo) where halo is 0 or 1."""
    halo = 1 if start > 0 else 0
    lo = start - halo
    local = [np.ascontiguousarray(c[lo:end]) for c in columns]
    flags = unpack_unit_bits(unit_bits, n_records)[lo:end].copy()
    if halo and flags.shape[0]:
        flags[0] = 0
    return local, pack_unit_bits(flags), halo


def to_global_index(local_idx, start, halo):
    """Record indices of a block's units in whole-input numbering."""
    return local_idx.astype(np.int64) + (start - halo)


def allreduce_counts(counts):
    """Sum category_counts over all ranks, in place.  `counts`: a torch int64 tensor of 64 elements
    (device tensor under nccl/RCCL, CPU tensor under gloo).  No-op without an initialised process group.
    The same reduction without torch.distributed: Context.comm_init + Context.allreduce_counts (xm_allreduce_counts,
    RCCL inside the library)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def gather_bin_lists(local_lists, dst=0):
    """Collect every rank's six (global-index) bin lists on `dst` and concatenate them in shard order,
    which is input order.  Host-side (object gather); returns the six arrays on dst, None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size()
    gathered = [None] * world if dist.get_rank() == dst else None
    dist.gather_object([np.asarray(x) for x in local_lists], gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [np.concatenate([gathered[g][b] for g in range(world)]) for b in range(len(local_lists))]


def classify_block(ctx, mode, columns, unit_bits, n_records, start, end, min_score_floor):
    """Run the fused pass (K1 + K2) on one block through the C ABI.  Returns (global bin lists [6 (+1 error slot)], counts u64[64])."""
    local, bits, halo = take_block(columns, unit_bits, n_records, start, end)
    _, idx, off, counts = ctx.classify_compact(mode, *local, bits, min_score_floor, want_code=False)
    lists = [to_global_index(idx[int(off[b]):int(off[b + 1])], start, halo) for b in range(7)]
    return lists, counts
