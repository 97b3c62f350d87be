#!/usr/bin/env python3
"""The placement effect under the L2 / memory-interface counters: ONE 8 GiB int64 column scanned into two bitsets, one the
arena verified as FAST with it (another placement class), one that lies in the column's OWN class (the top of a second
column allocated right behind it in the same zone) and is SLOW with it.  The process runs, in this order,

    16 x scan(column -> fast bitset),  16 x scan(column -> slow bitset),  16 x read-only form

and prints the scan times it measured itself; under `rocprofv3 --pmc <counters>` (tools/prof_placement.sh) the dispatches of
scan_kernel<8,1,1u,...> split into the first and the second 16 by order (after the arena's probes, which are FLV 1 kernels).
Usage: python3 tools/placement_counters.py"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K                                                # noqa: E402
from knoxdb_amd import And, Leaf                                      # noqa: E402
from knoxdb_amd.engine import Bits                                    # noqa: E402
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R   # noqa: E402

n, pack = 1 << 30, 1 << 20
ctx = K.Context(0)
ctx.set_arena(120 << 30)
ctx.reserve(11 << 30)
col = ctx.generate(B.Int64, 0, 2, n, pack_rows=pack, lo=-(1 << 40), span=1 << 41)
neighbour = ctx.generate(B.Int64, 3, 0, 1 << 28, lo=0, pack_rows=pack)          # 2 GiB right behind it: same zone, same class
fast = ctx.bits(n, for_col=col)
slow = Bits(ctx, n, wrap_ptr=neighbour.device_ptr + (1 << 30))                    # 128 MiB inside the neighbour
print("column class", col.placement_class(), "neighbour class", neighbour.placement_class(), "fast bitset pair check", fast.pair_check, flush=True)
plan = ctx.plan(And(Leaf(0, F.Range, (-(1 << 38), 1 << 38))), [col], aggs=[(R.Sum, 0), (R.Count, 0)])
ro = ctx.plan(And(Leaf(0, F.Range, (-(1 << 38), 1 << 38))), [col], aggs=[])


def ms(p, bits, steps=16):
    ctx.L.knox_prof_enable(ctx.h, 1)
    for _ in range(steps):
        ctx.scan_async(p, bits_out=bits)
        ctx.scan_collect(p, 0)
    v = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, v, 512, C.byref(k))
    ctx.L.knox_prof_enable(ctx.h, 0)
    x = [v[i] for i in range(k.value)]
    return statistics.median(x[len(x) // 2:])


t_fast, t_slow, t_ro = ms(plan, fast), ms(plan, slow), ms(ro, None)
print("scan -> fast bitset %.4f ms   scan -> slow bitset %.4f ms   read-only %.4f ms   (fast %.3f, slow %.3f of read-only)" %
      (t_fast, t_slow, t_ro, t_ro / t_fast, t_ro / t_slow), flush=True)
