#!/usr/bin/env python3
"""Time-window aggregation timings (whole calls): ordered clock at 2^30 rows, shuffled clock (stable-sort path) at 2^28."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

ctx = K.Context(0)
PACK = 1 << 20


def timed(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for lg, shuffled in ((30, False), (28, False), (28, True), (26, True)):
    n = 1 << lg
    col = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=PACK)
    clock = ctx.generate(B.Int64, 0, 9, n, lo=1000, span=1 << 20, pack_rows=PACK) if shuffled else ctx.generate(B.Int64, 3, 0, n, lo=1000, pack_rows=PACK)
    bits = ctx.bits(n, for_col=col)
    ctx.scan(And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), [col], bits_out=bits)
    for nb in (1024, 16384, 1 << 20):
        step = ((1 << 20) if shuffled else n) // nb
        ms = timed(lambda: ctx.reduce_buckets(clock, col, 1000, max(step, 1), nb, mask=bits))
        ms_v = timed(lambda: ctx.reduce_buckets(clock, col, 1000, max(step, 1), nb, mask=bits, moments=True))
        print("2^%d rows, %s clock, %7d windows: %8.2f ms  (%.1f G rows/s); with var/std %8.2f ms" % (
            lg, "shuffled" if shuffled else "ordered ", nb, ms, n / ms / 1e6, ms_v))
    col.free(); clock.free(); bits.free()
