#!/usr/bin/env python3
"""Diagnostic: the context's device-memory arena at work (include/knoxgpu.h).  Allocates `ncol` columns of
2^lg int64 rows out of an arena of `budget_gib`, one bitset FOR each column, and reports per column the placement class
masks and the steady-state rate of the real scan; then the same with the arena off (plain hipMalloc, fresh context).
usage: placement_check.py [lg=28] [ncol=24] [budget_gib=160]"""
import ctypes as C, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 24
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 160
n = 1 << lg


def rate(ctx, col, bits, steps=14):
    """bits = None: the read-only form of the same scan (count + sum, no bitset)"""
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
    return n * (8.125 if bits is not None else 8.0) / statistics.median(v[len(v) // 2:]) / 1e6


for on in (True, False):
    ctx = K.Context(0)
    ctx.set_placement(on)
    if on:
        ctx.set_arena(budget << 30)
    ctx.L.knox_prof_enable(ctx.h, 1)
    cols = [ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20) for _ in range(ncol)]
    bitsets = [ctx.bits(n, for_col=c) for c in cols]
    rate(ctx, cols[0], bitsets[0], 60)
    rates = [rate(ctx, c, b) for c, b in zip(cols, bitsets)]
    ro = [rate(ctx, c, None) for c in cols]
    print("arena %s: %s  held outside: %s" % ("on " if on else "off", json.dumps(ctx.arena_info()), ctx.placement_info()))
    print("  candidates : " + " ".join("%5d" % b.pair_check[1] for b in bitsets))
    print("  pair check : " + " ".join("%5.3f" % b.pair_check[0] for b in bitsets))
    print("  scan GB/s  : " + " ".join("%5.0f" % r for r in rates))
    print("  read-only  : " + " ".join("%5.0f" % r for r in ro))
    print("  scan / ro  : " + " ".join("%5.3f" % (a / b) for a, b in zip(rates, ro)))
    print("  min %.0f  median %.0f  max %.0f" % (min(rates), statistics.median(rates), max(rates)))
    for o in cols + bitsets:
        o.free()
    ctx.close()
