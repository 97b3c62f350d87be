#!/usr/bin/env python3
"""Diagnostic: per-column speed with the bitset store enabled / disabled (variant 2)."""
import ctypes as C, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
n = 1 << 28
ctx = K.Context(0)
cols = [ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20) for _ in range(16)]
bits = ctx.bits(n)
ctx.L.knox_prof_enable(ctx.h, 1)
def run(col, b, steps, aggs):
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [col], aggs=aggs)
    for i in range(steps):
        ctx.scan_async(plan, bits_out=b, pack_counts=False)
        ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = [ms[i] for i in range(k.value)]
    return n * 8.125 / statistics.median(v[6:]) / 1e6
run(cols[0], bits, 80, [(ReducerFunc.Sum, 0)])
print("col: count | store flavours: nt, none, plain, sc1, sc0sc1, sc0, nt sc1, nt sc0 sc1")
for ci, col in enumerate(cols):
    ctx.diag_set_variant(1)  # needs a diagnostics library: KNOXGPU_SO=knoxdb_amd/diag/libknoxgpu_p7.so
    a = run(col, None, 16, [(ReducerFunc.Count, 0)])
    row = []
    for var in (1, 2, 3, 4, 5, 6, 7, 8):
        ctx.diag_set_variant(var)
        row.append(run(col, bits, 16, [(ReducerFunc.Sum, 0)]))
    print("col %2d: %5.0f | " % (ci, a) + " ".join("%5.0f" % x for x in row))
