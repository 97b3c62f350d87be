#!/usr/bin/env python3
"""Does the rate of a TWO-column scan (BASELINE configs[2]) depend on WHERE the two columns lie relative to each other?
Five 8 GiB float64 columns are placed by the arena (their zone = placement class is visible in the address: one
reservation per zone); every pair is scanned with the read-only form of the cfg-3 tree and with the full scan."""
import ctypes as C, itertools, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R
n, pack = 1 << 30, 1 << 20
ctx = K.Context(0)
ctx.set_arena(120 << 30)
big = ctx.generate(B.Int64, 3, 0, 3 << 30, lo=0, pack_rows=pack)   # 24 GiB: one zone grows to hold three columns
big.free()
cols = [ctx.generate(B.Float64, 1 if i % 2 == 0 else 2, 3 + i, n, pack_rows=pack) for i in range(5)]
zone = lambda c: c.device_ptr >> 36
print("columns at", [hex(c.device_ptr) for c in cols], "zones", [hex(zone(c)) for c in cols])
tree = And(Leaf(0, F.Ge, 0.25), Leaf(1, F.Lt, 0.5))


def ms(p, bits, steps=12):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(p, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(p, bits_out=bits); ctx.scan_collect(p, 0)
    ctx.scan_collect(p, 0)
    v = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, v, 512, C.byref(k)); ctx.L.knox_prof_enable(ctx.h, 0)
    x = [v[i] for i in range(k.value)]
    return statistics.median(x[len(x) // 2:])


ro0 = ctx.plan(tree, [cols[0], cols[1]], aggs=[])
ms(ro0, None, 40)
for i, j in itertools.combinations(range(5), 2):
    ro = ctx.plan(tree, [cols[i], cols[j]], aggs=[])
    t = ms(ro, None)
    print("pair (%d, %d) %s: read-only %.4f ms = %.0f GB/s" % (i, j, "SAME zone" if zone(cols[i]) == zone(cols[j]) else "different zones", t, n * 16 / t / 1e6), flush=True)
print(ctx.arena_info())
