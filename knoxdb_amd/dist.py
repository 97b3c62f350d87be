"""Pack-sharded multi-GPU plumbing: one process per GPU, packs partitioned contiguously, no
data-path collective -- only the final partial aggregates cross ranks.

The exchange runs over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in
the CPU tests): an all-gather of each rank's knox_agg_state bytes, then the SAME deterministic
merge on every rank (knox_agg_combine, rank order == global row order).  A host without torch
(the Go shim) uses knox_comm_* in the C ABI, which does the identical exchange with RCCL
directly.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import AggState
from .engine import agg_combine


def shard_range(n_packs, rank, world):
    """Contiguous pack range [first, last) of `rank`: GPU g owns packs [g*P/W, (g+1)*P/W)."""
    return (rank * n_packs) // world, ((rank + 1) * n_packs) // world


def states_to_bytes(states, count):
    buf = bytearray()
    for s in states:
        buf += bytes(s)
    buf += int(count).to_bytes(8, "little", signed=True)
    return bytes(buf)


def bytes_to_states(raw, n_states):
    sz = C.sizeof(AggState)
    states = [AggState.from_buffer_copy(raw[i * sz:(i + 1) * sz]) for i in range(n_states)]
    count = int.from_bytes(raw[n_states * sz:n_states * sz + 8], "little", signed=True)
    return states, count


def allgather_states(states, count, device=None, group=None):
    """All-gather every rank's partial states and merge them in rank order.
    Returns (merged_states, total_count), identical on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = states_to_bytes(states, count)
    t = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    per_rank = [bytes_to_states(bytes(o.cpu().numpy().tobytes()), len(states)) for o in out]
    merged = [agg_combine([per_rank[r][0][i] for r in range(world)]) for i in range(len(states))]
    total = sum(c for _, c in per_rank)
    return merged, total
