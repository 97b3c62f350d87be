#!/usr/bin/env python3
"""What bounds the 1- / 2-byte scans: for uint8 / int16 / int32 columns of 2^30 rows, Between + sum,
  * the read-only form (no bitset written) against the full scan (1/8 B written per row: 11 % / 6 % / 3 % of the traffic),
  * wave iterations per bitset burst (KNOX_ITERS 1 .. 16: 0.5 .. 8 KiB contiguous per wave for 1-byte values) x workgroups per CU.
usage: sweep_narrow.py [rows_log2=30] [all]      (all: every aggregate instead of the wrapping sum)"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
AGGS = [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.First, 0), (R.Last, 0)] if "all" in sys.argv[2:] else [(R.Sum, 0)]
n, pack = 1 << lg, 1 << 20
ctx = K.Context(0)


def rate(plan, bits, bpr, steps=14):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(plan, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits); ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k)); ctx.L.knox_prof_enable(ctx.h, 0)
    v = [ms[i] for i in range(k.value)]
    return n * bpr / statistics.median(v[len(v) // 2:]) / 1e6


NARROW = ((B.Uint8, 1, 0, 256, 64, 127), (B.Int8, 1, -128, 256, -32, 31), (B.Int16, 2, -(1 << 15), 1 << 16, -(1 << 13), 1 << 13), (B.Int32, 4, -(1 << 30), 1 << 31, -(1 << 28), 1 << 28))
for bt, w, lo, span, a, b in (NARROW if "wide" not in sys.argv[2:] else ()):
    col = ctx.generate(bt, 0, 2, n, lo=lo, span=span, pack_rows=pack)
    bits = ctx.bits(n, for_col=col)
    plan = ctx.plan(And(Leaf(0, F.Range, a, b)), [col], aggs=AGGS)
    rate(plan, bits, w + .125, 40)
    for it in (1, 2, 4):
        os.environ["KNOX_ITERS"] = str(it)
        full, ro = [], []
        for bpc in (2, 3, 4, 6, 8, 12):
            ctx.set_tuning(bpc)
            full.append(rate(plan, bits, w + .125))
            ro.append(rate(plan, None, w))
        print("%-6s iters %2d  scan      " % (bt.name, it) + " ".join("b%d:%5.0f" % (b_, r) for b_, r in zip((2, 3, 4, 6, 8, 12), full)))
        print("%-6s iters %2d  read-only " % (bt.name, it) + " ".join("b%d:%5.0f" % (b_, r) for b_, r in zip((2, 3, 4, 6, 8, 12), ro)), flush=True)
    ctx.set_tuning(0)
    os.environ.pop("KNOX_ITERS", None)
    col.free(); bits.free()
if "all" in sys.argv[2:]:   # the wide aggregate-heavy instantiations beside them
    FL = [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)]
    for bt, w, kind, tree, aggs in ((B.Int64, 8, 0, And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), AGGS), (B.Float64, 8, 2, And(Leaf(0, F.Lt, 0.25)), FL),
                                    (B.Float32, 4, 2, And(Leaf(0, F.Lt, 0.25)), FL)):
        col = ctx.generate(bt, kind, 2, n, lo=-(1 << 40) if kind == 0 else 0, span=(1 << 41) if kind == 0 else 0, pack_rows=pack)
        bits = ctx.bits(n, for_col=col)
        plan = ctx.plan(tree, [col], aggs=aggs)
        rate(plan, bits, w + .125, 40)
        for it in (2, 4, 8):
            os.environ["KNOX_ITERS"] = str(it)
            full = []
            for bpc in (2, 3, 4, 6, 8, 12):
                ctx.set_tuning(bpc)
                full.append(rate(plan, bits, w + .125))
            print("%-7s iters %2d  scan      " % (bt.name, it) + " ".join("b%d:%5.0f" % (b_, r) for b_, r in zip((2, 3, 4, 6, 8, 12), full)), flush=True)
        ctx.set_tuning(0)
        os.environ.pop("KNOX_ITERS", None)
        col.free(); bits.free()
