#!/usr/bin/env python3
"""What one {sum, count} exchange per step costs the headline scan on ONE GPU (1-rank communicator of the library's own,
no torch): steps per second and the scan kernel's duration with the exchange off / on, and with 1 or 2 scans queued ahead.
Usage: python3 tools/dist_overhead.py [steps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("KNOX_TOOL_TORCH") == "1":   # torch's bundled librccl is then the one the library's dlopen finds (as under bench.py)
    import torch                                          # noqa: E402,F401
if os.environ.get("KNOX_TOOL_TORCH") == "2":   # ... and a 1-rank nccl process group beside it, as bench.py has
    import torch                                          # noqa: E402
    import torch.distributed as tdist                     # noqa: E402
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    tdist.barrier()
import knoxdb_amd as K                                    # noqa: E402
from knoxdb_amd import And, Leaf, dist as kdist           # noqa: E402
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n, pack = 1 << 30, 1 << 20
with K.Context(0) as ctx:
    ctx.set_arena(64 << 30)
    col = ctx.generate(BlockType.Int64, 0, 2, n, pack_rows=pack, lo=-(1 << 40), span=1 << 41)
    bits = ctx.bits(n, for_col=col)
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, (-(1 << 38), 1 << 38))), [col], aggs=[(ReducerFunc.Sum, 0), (ReducerFunc.Count, 0)])
    ranks = kdist.comm_init(ctx, 1, 0, kdist.comm_unique_id())
    red = kdist.AbiSumCountReduce(ctx)
    print("pair check", bits.pair_check, "ranks in communicator", ranks)

    def run(k, depth, exchange):
        queued, pending, last = 0, 0, None
        for _ in range(min(depth, k)):
            ctx.scan_async(plan, bits_out=bits, pack_counts=True)
            queued += 1
        for _ in range(k):
            if queued < k:
                ctx.scan_async(plan, bits_out=bits, pack_counts=True)
                queued += 1
            res = ctx.scan_collect(plan, n // pack)
            if exchange:
                red.start(int(res.states[0].isum), res.count)
                pending += 1
                if pending > 1:
                    last = red.finish()
                    pending -= 1
        while pending:
            last = red.finish()
            pending -= 1
        return last

    for name, depth, exchange in (("depth 1, no exchange", 1, False), ("depth 2, no exchange", 2, False), ("depth 1, exchange", 1, True),
                                  ("depth 2, exchange", 2, True), ("depth 1, no exchange", 1, False)):
        run(12, depth, exchange)
        ctx.L.knox_prof_enable(ctx.h, 1)
        ctx.sync()
        t0 = time.perf_counter()
        run(steps, depth, exchange)
        ctx.sync()
        dt = time.perf_counter() - t0
        each = (C.c_float * 512)()
        k = C.c_int()
        ctx.L.knox_prof_read_each(ctx.h, each, 512, C.byref(k))
        ctx.L.knox_prof_enable(ctx.h, 0)
        v = [each[i] for i in range(k.value)]
        print("%-24s step %.4f ms  scan kernel %.4f ms (min %.4f max %.4f)  outside %.1f us" % (name, dt / steps * 1e3, sum(v) / len(v), min(v), max(v), (dt / steps * 1e3 - sum(v) / len(v)) * 1e3))
