#!/usr/bin/env python3
"""Diagnostic: spatial structure of the read+write placement effect.  Allocates `ncol` columns of
2^lg int64 rows (hipMalloc) and measures the steady-state scan rate (Between + sum + bitset) of every
2^seg-row segment of every column (knox_col_wrap on sub-ranges).
usage: placement_segments.py [lg=30] [ncol=3] [seg=25]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 25
n, m = 1 << lg, 1 << seg
ctx = K.Context(0)
ctx.L.knox_prof_enable(ctx.h, 1)
bits = ctx.bits(m)


def rate(col, steps=12):
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
    return m * 8.125 / statistics.median(v[len(v) // 2:]) / 1e6


cols = [ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20) for _ in range(ncol)]
w0 = ctx.wrap(BlockType.Int64, cols[0].device_ptr, m, 1 << 20)
rate(w0, 60)
for ci, c in enumerate(cols):
    row = []
    for s in range(n // m):
        w = ctx.wrap(BlockType.Int64, c.device_ptr + s * m * 8, m, 1 << 20)
        row.append(rate(w))
        w.free()
    print("col %d (va 0x%x): " % (ci, c.device_ptr) + " ".join("%4.0f" % (x / 10) for x in row))
