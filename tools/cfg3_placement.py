#!/usr/bin/env python3
"""cfg 3 (two float64 columns, AND, min / max / mean) against several candidate bitsets in ONE arena: does the pairwise
verification (knox_bits_alloc_for / _for_cols) predict the rate of the two-column scan?"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R
n, pack = 1 << 30, 1 << 20
ctx = K.Context(0)
ctx.set_arena(160 << 30)
warm = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)   # the headline column first, as in bench.py
wb = ctx.bits(n, for_col=warm)
warm.free(); wb.free()
A = ctx.generate(B.Float64, 1, 3, n, pack_rows=pack)
Bc = ctx.generate(B.Float64, 2, 4, n, pack_rows=pack)
tree = And(Leaf(0, F.Ge, 256.0), Leaf(1, F.Lt, 0.5))
plan = ctx.plan(tree, [A, Bc], aggs=[(R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.Sum, 0), (R.Count, 0)])
ro = ctx.plan(tree, [A, Bc], aggs=[])


def ms(p, bits, steps=16):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(p, bits_out=bits, pack_counts=bits is not None)
    for _ in range(steps - 1):
        ctx.scan_async(p, bits_out=bits, pack_counts=bits is not None); ctx.scan_collect(p, n // pack if bits is not None else 0)
    ctx.scan_collect(p, n // pack if bits is not None else 0)
    v = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, v, 512, C.byref(k)); ctx.L.knox_prof_enable(ctx.h, 0)
    x = [v[i] for i in range(k.value)]
    return statistics.median(x[len(x) // 2:])


ms(ro, None, 40)
print("read-only two-column form: %.4f ms = %.0f GB/s" % (ms(ro, None), n * 16 / ms(ro, None) / 1e6))
for name, mk in (("for [A, B]", lambda: ctx.bits(n, for_col=[A, Bc])), ("for A", lambda: ctx.bits(n, for_col=A)), ("for B", lambda: ctx.bits(n, for_col=Bc)),
                 ("plain 1", lambda: ctx.bits(n)), ("plain 2", lambda: ctx.bits(n)), ("plain 3", lambda: ctx.bits(n)), ("for [A, B] again", lambda: ctx.bits(n, for_col=[A, Bc]))):
    b = mk()
    t = ms(plan, b)
    print("%-18s pair check %-22s scan %.4f ms = %.0f GB/s = %.3f of peak" % (name, b.pair_check, t, n * 16.125 / t / 1e6, n * 16.125 / t / 8e9), flush=True)
print(ctx.arena_info())
# occupancy x burst iterations for the two-leaf float scan, on the bitset verified against both columns
b = ctx.bits(n, for_col=[A, Bc])
for it in (1, 2, 4, 8):
    os.environ["KNOX_ITERS"] = str(it)
    row = []
    for bpc in (2, 3, 4, 6, 8):
        ctx.set_tuning(bpc)
        row.append(n * 16.125 / ms(plan, b) / 1e6)
    print("cfg3 iters %d  " % it + " ".join("b%d:%5.0f" % (x, r) for x, r in zip((2, 3, 4, 6, 8), row)), flush=True)
