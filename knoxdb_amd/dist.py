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


class StateGather:
    """The per-step exchange with everything allocated once: a page-locked host buffer each way, one device tensor
    each way, ONE collective (all_gather_into_tensor) and ONE device-to-host copy per call -- the exchange is a few
    hundred bytes, so its cost is launches and synchronisations, not bandwidth.  device=None: CPU tensors (gloo)."""

    def __init__(self, n_states, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.n_states = n_states
        self.world = dist.get_world_size(group)
        self.nbytes = n_states * C.sizeof(AggState) + 8
        self.device = device
        if device is not None:
            self.h_in = torch.empty(self.nbytes, dtype=torch.uint8).pin_memory()
            self.h_out = torch.empty(self.nbytes * self.world, dtype=torch.uint8).pin_memory()
            self.d_in = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
            self.d_out = torch.empty(self.nbytes * self.world, dtype=torch.uint8, device=device)
        else:
            self.h_in = torch.empty(self.nbytes, dtype=torch.uint8)
            self.h_out = torch.empty(self.nbytes * self.world, dtype=torch.uint8)

    def __call__(self, states, count):
        torch, dist = self.torch, self.dist
        mine = states_to_bytes(states, count)
        assert len(mine) == self.nbytes
        self.h_in.numpy()[:] = np.frombuffer(mine, dtype=np.uint8)
        if self.device is not None:
            self.d_in.copy_(self.h_in, non_blocking=True)
            dist.all_gather_into_tensor(self.d_out, self.d_in, group=self.group)
            self.h_out.copy_(self.d_out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        else:
            parts = [self.h_out[r * self.nbytes:(r + 1) * self.nbytes] for r in range(self.world)]
            dist.all_gather(parts, self.h_in, group=self.group)
        raw = self.h_out.numpy().tobytes()
        per_rank = [bytes_to_states(raw[r * self.nbytes:(r + 1) * self.nbytes], self.n_states) for r in range(self.world)]
        merged = [agg_combine([per_rank[r][0][i] for r in range(self.world)]) for i in range(self.n_states)]
        return merged, sum(c for _, c in per_rank)


class SumCountReduce:
    """The literal collective of the north star for the order-free integer aggregates: ONE all_reduce(int64, SUM) of
    {low half of the wrapping sum, high half, count} (backend "nccl" = ncclAllReduce over xGMI).  The 64-bit sum
    travels as two 32-bit halves so that no lane overflows and the result is exactly the sum mod 2^64 in any reduction
    order -- bit-identical to StateGather + knox_agg_combine.  Everything is allocated once.

    start() / finish() split the exchange so that a host can keep it in flight behind the next scan (a query's answer
    is then read one step later, exactly like the scans' own result ring): `slots` exchanges may be outstanding."""

    def __init__(self, device=None, group=None, slots=2):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, device
        self.slots, self._next = slots, 0
        self.h = [torch.empty(3, dtype=torch.int64) for _ in range(slots)]
        if device is not None:
            self.h = [t.pin_memory() for t in self.h]
            self.d = [torch.empty(3, dtype=torch.int64, device=device) for _ in range(slots)]
            self.ev = [torch.cuda.Event() for _ in range(slots)]

    def start(self, isum, count):
        """enqueue one exchange; -> the handle finish() takes.  isum: any Python int, taken mod 2^64"""
        k = self._next
        self._next = (k + 1) % self.slots
        isum &= (1 << 64) - 1
        hv = self.h[k].numpy()
        hv[0], hv[1], hv[2] = isum & 0xFFFFFFFF, isum >> 32, count
        if self.device is not None:
            self.d[k].copy_(self.h[k], non_blocking=True)
            self.dist.all_reduce(self.d[k], op=self.dist.ReduceOp.SUM, group=self.group)  # the stream waits, not the host
            self.h[k].copy_(self.d[k], non_blocking=True)
            self.ev[k].record(self.torch.cuda.current_stream(self.device))
        else:
            self.dist.all_reduce(self.h[k], op=self.dist.ReduceOp.SUM, group=self.group)
        return k

    def finish(self, k):
        """-> (sum mod 2^64, total count) of exchange k"""
        if self.device is not None:
            self.ev[k].synchronize()
        hv = self.h[k].numpy()
        return (int(hv[0]) + (int(hv[1]) << 32)) & ((1 << 64) - 1), int(hv[2])

    def __call__(self, isum, count):
        return self.finish(self.start(isum, count))


# ---- the library's own RCCL exchange (knox_comm_*): what a Go host calls, no torch in the data path ----------------
def comm_unique_id():
    """ncclGetUniqueId through the ABI (rank 0 creates it, every rank passes the same bytes to comm_init)"""
    buf = (C.c_uint8 * 128)()
    rc = _lib.lib().knox_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError("knox_comm_unique_id: %s" % _lib.lib().knox_strerror(rc).decode())
    return bytes(buf)


def comm_init(ctx, n_ranks, rank, unique_id):
    """ncclCommInitRank on the context's device; returns the communicator's own size (ncclCommCount)"""
    buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
    _lib.check(ctx.L.knox_comm_init(ctx.h, n_ranks, rank, buf), "knox_comm_init")
    n, r = C.c_int(), C.c_int()
    _lib.check(ctx.L.knox_comm_ranks(ctx.h, C.byref(n), C.byref(r)), "knox_comm_ranks")
    return n.value


class AbiSumCountReduce:
    """SumCountReduce over knox_comm_allreduce_sum_count_start / _finish: ONE ncclAllReduce(int64 x 3, sum) per exchange on
    the context's collective stream, up to 4 in flight (FIFO)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def start(self, isum, count):
        _lib.check(self.ctx.L.knox_comm_allreduce_sum_count_start(self.ctx.h, isum & ((1 << 64) - 1), count), "knox_comm_allreduce_sum_count_start")
        return 0

    def finish(self, _k=0):
        s, c = C.c_uint64(), C.c_int64()
        _lib.check(self.ctx.L.knox_comm_allreduce_sum_count_finish(self.ctx.h, C.byref(s), C.byref(c)), "knox_comm_allreduce_sum_count_finish")
        return int(s.value), int(c.value)

    def __call__(self, isum, count):
        self.start(isum, count)
        return self.finish()


class AbiStateGather:
    """StateGather over knox_comm_allgather_states: ncclAllGather of every rank's states + the same deterministic
    knox_agg_combine in rank order on every rank."""

    def __init__(self, ctx, n_states):
        self.ctx, self.n_states = ctx, n_states

    def __call__(self, states, count):
        mine = (AggState * self.n_states)(*states)
        merged = (AggState * self.n_states)()
        c = C.c_int64(count)
        _lib.check(self.ctx.L.knox_comm_allgather_states(self.ctx.h, mine, self.n_states, merged, C.byref(c)), "knox_comm_allgather_states")
        return [merged[i] for i in range(self.n_states)], int(c.value)
