#!/usr/bin/env python3
"""Diagnostic: placement-aware allocation at work.  Allocates `ncol` columns of 2^lg int64 rows with
placement on (default) and reports, per column, the allocator's probe rate / candidates tried and the
steady-state rate of the real scan; then the same with placement off (fresh context).
usage: placement_check.py [lg=28] [ncol=24]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 1 << lg


def rate(ctx, col, bits, steps=14):
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [col], aggs=[(ReducerFunc.Sum, 0)])
    ctx.scan_async(plan, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits)
        ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = [ms[i] for i in range(k.value)]
    return n * 8.125 / statistics.median(v[len(v) // 2:]) / 1e6


for on in (True, False):
    ctx = K.Context(0)
    ctx.set_placement(on)
    ctx.L.knox_prof_enable(ctx.h, 1)
    cols = [ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20) for _ in range(ncol)]
    bits = ctx.bits(n, for_col=cols[0]) if on else ctx.bits(n)
    rate(ctx, cols[0], bits, 60)
    rates = [rate(ctx, c, bits) for c in cols]
    print("placement %s: held %s" % ("on " if on else "off", ctx.placement_info()))
    print("  probe GB/s : " + " ".join("%5.0f" % c.placement[0] for c in cols))
    print("  candidates : " + " ".join("%5d" % c.placement[1] for c in cols))
    print("  mix/ro     : " + " ".join("%5.3f" % c.placement[2] for c in cols))
    print("  scan GB/s  : " + " ".join("%5.0f" % r for r in rates))
    print("  min %.0f  median %.0f  max %.0f" % (min(rates), statistics.median(rates), max(rates)))
    for c in cols:
        c.free()
    bits.free()
    ctx.close()
