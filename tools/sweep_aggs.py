#!/usr/bin/env python3
"""Occupancy / tile-iteration sweep of the aggregate-heavy instantiations: rates (GB/s) for blocks-per-CU x KNOX_ITERS."""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R
n, pack = 1 << 30, 1 << 20
ctx = K.Context(0)


def rate(bits, plan, bpr, steps=14):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(plan, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits); ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k)); ctx.L.knox_prof_enable(ctx.h, 0)
    v = [ms[i] for i in range(k.value)]
    return n * bpr / statistics.median(v[len(v) // 2:]) / 1e6


ALL = [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.First, 0), (R.Last, 0)]
cases = [(B.Int32, 4, dict(lo=-(1 << 30), span=1 << 31), Leaf(0, F.Range, -(1 << 28), 1 << 28), ALL, 0),
         (B.Int64, 8, dict(lo=-(1 << 40), span=1 << 41), Leaf(0, F.Range, -(1 << 38), 1 << 38), ALL, 0),
         (B.Float32, 4, {}, Leaf(0, F.Lt, 0.25), [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)], 2),
         (B.Float64, 8, {}, Leaf(0, F.Lt, 0.25), [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)], 2),
         (B.Int16, 2, dict(lo=-(1 << 15), span=1 << 16), Leaf(0, F.Range, -(1 << 13), 1 << 13), ALL, 0)]
for bt, w, gen, leaf, aggs, kind in cases:
    col = ctx.generate(bt, kind, 2 if kind == 0 else 4, n, pack_rows=pack, **gen)
    bits = ctx.bits(n, for_col=col)
    plan = ctx.plan(And(leaf), [col], aggs=aggs)
    ctx.set_tuning(0)
    os.environ.pop("KNOX_ITERS", None)
    print(bt.name, "default: %5.0f GB/s" % rate(bits, plan, w + .125, 30), flush=True)
    for it in (1, 2, 4):
        os.environ["KNOX_ITERS"] = str(it)
        row = []
        for bpc in (2, 3, 4, 5, 6, 8):
            ctx.set_tuning(bpc)
            row.append(rate(bits, plan, w + .125))
        print("  iters", it, " ".join("b%d:%5.0f" % (b_, r) for b_, r in zip((2, 3, 4, 5, 6, 8), row)), flush=True)
    col.free(); bits.free()
