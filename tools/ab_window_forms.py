#!/usr/bin/env python3
"""The three launch forms of the time-window reduce step (KNOX_WINDOW_FORM=block | wave | group, knox_windows.cpp) on the same
resident columns: ordered clock, values path (Sum finalised on the device, 9 B per window back), order known.
Whole calls, wall clock.  usage: ab_window_forms.py [rows_log2 ...] (default 30 28 26)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

ctx = K.Context(0)
lgs = [int(x) for x in sys.argv[1:]] or [30, 28, 26]


def timed(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for lg in lgs:
    n = 1 << lg
    col = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20)
    clock = ctx.generate(B.Int64, 3, 0, n, lo=1000, pack_rows=1 << 20)
    bits = ctx.bits(n, for_col=col)
    ctx.scan(And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), [col], bits_out=bits)
    for nb in (1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
        if nb * 8 > n:
            continue
        step = n // nb
        line = "2^%d rows, %8d windows (%6d rows each):" % (lg, nb, step)
        for form in ("block", "wave", "group", ""):
            if form:
                os.environ["KNOX_WINDOW_FORM"] = form
            else:
                os.environ.pop("KNOX_WINDOW_FORM", None)
            for m, tag in ((bits, "masked"), (None, "all")):
                ms = timed(lambda: ctx.reduce_buckets_values(clock, col, R.Sum, 1000, step, nb, mask=m, assume_sorted=True))
                line += "  %s/%s %6.2f" % (form or "default", tag, ms)
            if form in ("block", "group"):
                ms = timed(lambda: ctx.reduce_buckets_values(clock, col, R.Std, 1000, step, nb, mask=bits, assume_sorted=True))
                line += "  %s/std %6.2f" % (form, ms)
        print(line + "  ms", flush=True)
    col.free(); clock.free(); bits.free()
