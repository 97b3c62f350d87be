#!/usr/bin/env python3
"""Where the time between scan kernels goes: whole-step time of the headline scan with / without per-pack counts,
with / without aggregates, against the kernel's own duration (one-deep software pipeline as in bench.py)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

ctx = K.Context(0)
n, pack = 1 << 30, 1 << 20
col = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
bits = ctx.bits(n, for_col=col)
tree = And(Leaf(0, F.Range, -(1 << 38), 1 << 38))
for name, aggs, pc, wb in (("sum+count, bitset, pack counts", [(R.Sum, 0), (R.Count, 0)], True, True), ("sum+count, bitset", [(R.Sum, 0), (R.Count, 0)], False, True),
                           ("count only, bitset", [], False, True), ("count only, no bitset", [], False, False)):
    plan = ctx.plan(tree, [col], aggs=aggs)
    def enq():
        ctx.scan_async(plan, bits_out=bits if wb else None, pack_counts=pc)
    def col_():
        ctx.scan_collect(plan, n // pack if pc else 0)
    for _ in range(10):
        enq(); col_()
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.sync(); t0 = time.perf_counter()
    steps = 50
    enq()
    for _ in range(steps - 1):
        enq(); col_()
    col_(); ctx.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    ms, k = C.c_double(), C.c_int()
    ctx.L.knox_prof_read(ctx.h, C.byref(ms), C.byref(k)); ctx.L.knox_prof_enable(ctx.h, 0)
    print("%-34s step %.4f ms  kernel %.4f ms  outside the kernel %.1f us" % (name, dt, ms.value / k.value, (dt - ms.value / k.value) * 1e3))
