#!/usr/bin/env python3
"""The aggregate-heavy scan instantiations: steady-state kernel time and achieved GB/s of
Between + {sum | ALL aggregates} + bitset for int32 / int16 / uint8 and Lt + sum/min/max/mean for float64 / float32.
Also the command the SQ counter pass is taken over:
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -- python3 tools/bench_narrow.py
usage: bench_narrow.py [rows_log2=30] [steps=24]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n, pack = 1 << lg, 1 << 20
ctx = K.Context(0)
L = ctx.L


def scan_ms(plan, bits):
    L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(plan, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits)
        ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)()
    k = C.c_int()
    L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    L.knox_prof_enable(ctx.h, 0)
    v = [ms[i] for i in range(k.value)]
    return statistics.median(v[len(v) // 3:])


def rec(name, ms, nbytes):
    print("%-52s %9.4f ms %9.1f GB/s  (%.1f%% of 8 TB/s)" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80), flush=True)


ALL = [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.First, 0), (R.Last, 0)]
for bt, w, lo, span, a, b in ((B.Int64, 8, -(1 << 40), 1 << 41, -(1 << 38), 1 << 38), (B.Int32, 4, -(1 << 30), 1 << 31, -(1 << 28), 1 << 28),
                              (B.Uint32, 4, 0, 1 << 32, 1 << 30, 3 << 30), (B.Int16, 2, -(1 << 15), 1 << 16, -(1 << 13), 1 << 13),
                              (B.Uint16, 2, 0, 1 << 16, 1 << 14, 3 << 14), (B.Int8, 1, -128, 256, -32, 31), (B.Uint8, 1, 0, 256, 64, 127)):
    col = ctx.generate(bt, 0, 2, n, lo=lo, span=span, pack_rows=pack)
    bits = ctx.bits(n, for_col=col)
    rec("scan %s Between + sum + bitset" % bt.name, scan_ms(ctx.plan(And(Leaf(0, F.Range, a, b)), [col], aggs=[(R.Sum, 0)]), bits), n * (w + 0.125))
    rec("scan %s Between + ALL aggs + bitset" % bt.name, scan_ms(ctx.plan(And(Leaf(0, F.Range, a, b)), [col], aggs=ALL), bits), n * (w + 0.125))
    col.free()
    bits.free()
for bt, w in ((B.Float64, 8), (B.Float32, 4)):
    col = ctx.generate(bt, 2, 4, n, pack_rows=pack)
    bits = ctx.bits(n, for_col=col)
    plan = ctx.plan(And(Leaf(0, F.Lt, 0.25)), [col], aggs=[(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)])
    rec("scan %s Lt + sum/min/max/mean + bitset" % bt.name, scan_ms(plan, bits), n * (w + 0.125))
    col.free()
    bits.free()
