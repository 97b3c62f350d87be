#!/usr/bin/env python3
"""Diagnostic: is the read+write placement effect a property of the column, of the bitset buffer, or of
the pair?  `ncol` columns x `nbits` bitset buffers, steady-state scan rate for every pair.
usage: placement_pairs.py [lg=28] [ncol=6] [nbits=10]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nbits = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = 1 << lg
ctx = K.Context(0)
ctx.set_placement(False)
ctx.L.knox_prof_enable(ctx.h, 1)


def rate(col, bits, steps=12):
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


cols, bitss = [], []
for i in range(max(ncol, nbits)):  # interleave the allocations so both kinds spread over the address space
    if i < ncol:
        cols.append(ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20))
    if i < nbits:
        bitss.append(ctx.bits(n))
rate(cols[0], bitss[0], 60)
print("        " + " ".join("bits%-2d" % j for j in range(nbits)))
for i, c in enumerate(cols):
    print("col %2d: " % i + " ".join("%6.0f" % rate(c, b) for b in bitss))
