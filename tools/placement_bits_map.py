#!/usr/bin/env python3
"""Diagnostic: placement class of many small bitset buffers (32 MiB each) against ONE column, with their
device addresses -- how does the class vary along the address space?
usage: placement_bits_map.py [nbits=160] [lg=28]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 160
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 28
n = 1 << lg
ctx = K.Context(0)
ctx.set_placement(False)
ctx.L.knox_prof_enable(ctx.h, 1)
ctx.L.knox_bits_device_ptr.restype = C.c_void_p


def rate(col, bits, steps=10):
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


cols = [ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20) for _ in range(2)]
bitss = [ctx.bits(n) for _ in range(nbits)]
rate(cols[0], bitss[0], 60)
print("col0 va 0x%x  col1 va 0x%x" % (cols[0].device_ptr, cols[1].device_ptr))
rows = []
for b in bitss:
    va = ctx.L.knox_bits_device_ptr(b.h)
    rows.append((va, rate(cols[0], b), rate(cols[1], b)))
for va, r0, r1 in sorted(rows):
    print("0x%012x  %s %s   %5.0f %5.0f" % (va, "F" if r0 > 6450 else "s", "F" if r1 > 6450 else "s", r0, r1))
