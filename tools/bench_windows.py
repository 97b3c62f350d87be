#!/usr/bin/env python3
"""Time-window aggregation and selection-vector timings (whole calls, wall clock around the C call):
ordered clock at 2^30 / 2^28 rows, shuffled clock (hand-written stable radix sort path) at 2^28 / 2^26, each as
  states  knox_reduce_buckets          144-byte states to the caller's (pageable) memory
  values  knox_reduce_buckets_values   Reducer.Value() finalised on the device, 8+1 B/window into the pinned result buffer
and knox_bits_indexes / knox_bits_indexes_view of 16.8 M hits, and the flag / count calls blocking vs through the ring."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

from knoxdb_amd._lib import AggState

ctx = K.Context(0)
PACK = 1 << 20
quick = "--quick" in sys.argv
profile = "--profile" in sys.argv      # under rocprofv3: 2 repetitions, the shapes the review names
REPS = 2 if profile else 5


def timed(fn, reps=None):
    reps = reps or REPS
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for lg, shuffled in ((28, False), (28, True)) if quick else ((30, False), (26, False), (28, True)) if profile else ((30, False), (28, False), (26, False), (28, True), (26, True)):
    n = 1 << lg
    col = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=PACK)
    clock = ctx.generate(B.Int64, 0, 9, n, lo=1000, span=1 << 20, pack_rows=PACK) if shuffled else ctx.generate(B.Int64, 3, 0, n, lo=1000, pack_rows=PACK)
    bits = ctx.bits(n, for_col=col)
    ctx.scan(And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), [col], bits_out=bits)
    for nb in (1024, 16384, 1 << 20):
        step = ((1 << 20) if shuffled else n) // nb
        out = (AggState * nb)()      # the caller's (pageable) destination, kept across calls
        rb = lambda **kw: ctx.reduce_buckets(clock, col, 1000, max(step, 1), nb, mask=bits, out=out, **kw)
        rv = lambda f, **kw: ctx.reduce_buckets_values(clock, col, f, 1000, max(step, 1), nb, mask=bits, **kw)
        # "checked": the order check of the time column runs on every call (flags bit4), as it did before the answer was
        # kept with the column; the other figures are what a repeated query of a resident time column costs
        ms_chk = timed(lambda: rb(recheck=True))
        ms = timed(lambda: rb())
        ms_v = timed(lambda: rb(moments=True))
        ms_view = timed(lambda: ctx.reduce_buckets_view(clock, col, 1000, max(step, 1), nb, mask=bits))
        ms_sum_chk = timed(lambda: rv(R.Sum, recheck=True))
        ms_sum = timed(lambda: rv(R.Sum))
        ms_std = timed(lambda: rv(R.Std))
        print("2^%d rows, %s clock, %7d windows: states %7.2f ms checked (%.1f G rows/s), %7.2f ms order known (%.1f G rows/s), with var/std %7.2f ms, as a pinned view %7.2f ms | values: sum %7.2f ms checked (%.1f G rows/s), %7.2f ms order known (%.1f G rows/s), std %7.2f ms" % (
            lg, "shuffled" if shuffled else "ordered ", nb, ms_chk, n / ms_chk / 1e6, ms, n / ms / 1e6, ms_v, ms_view, ms_sum_chk, n / ms_sum_chk / 1e6, ms_sum, n / ms_sum / 1e6, ms_std), flush=True)
    if not shuffled and lg <= 28:
        # the literal reducers per window (knox_reduce_buckets_sequential): one lane per window against one wave per window
        for nb in (1024, 16384, 65536, 1 << 20):
            step = n // nb
            for fn in (R.Mean, R.StdJoin):
                t = {}
                for form in ("lane", "wave"):
                    if form == "lane" and nb <= 1024 and lg > 26:
                        t[form] = float("nan")     # seconds per call in this corner: measured once at 2^26 rows
                        continue
                    t[form] = timed(lambda: ctx.reduce_buckets_sequential(clock, col, fn, 1000, step, nb, mask=bits, assume_sorted=True, form=form), reps=2)
                print("2^%d rows, literal %-8s %7d windows: one lane per window %8.2f ms (%.0f GB/s) | one wave per window %8.2f ms (%.0f GB/s)" % (
                    lg, fn.name, nb, t["lane"], n * 16.125 / t["lane"] / 1e6, t["wave"], n * 16.125 / t["wave"] / 1e6), flush=True)
    col.free(); clock.free(); bits.free()
if profile:
    sys.exit(0)

# selection vector of 16.8 M hits out of 2^26 rows
n = 1 << 26
col = ctx.generate(B.Int64, 0, 2, n, lo=0, span=1 << 20, pack_rows=PACK)
bits = ctx.bits(n, for_col=col)
res = ctx.scan(And(Leaf(0, F.Lt, 1 << 18)), [col], bits_out=bits)
pinned = ctx.host_array(res.count + 8, np.uint32)
print("indexes of %.1f M hits: view (pinned result buffer) %.2f ms | into knox_host_alloc memory %.2f ms | into a fresh pageable numpy array %.2f ms" % (
    res.count / 1e6, timed(lambda: bits.indexes_view()), timed(lambda: bits.indexes(out=pinned)), timed(lambda: bits.indexes(capacity=res.count + 8))), flush=True)

# flags / counts: blocking calls vs four enqueued + four collected
a, b = ctx.bits(n), ctx.bits(n)
a.one(); b.one()


def ring4():
    for _ in range(4):
        a.and_flag_async(b)
    for _ in range(4):
        ctx.bits_collect()


def ring4c():
    for _ in range(4):
        a.count_async()
    for _ in range(4):
        ctx.bits_collect()


print("2^26-bit bitsets: and_flag blocking %.1f us/call, through the ring %.1f us/call | popcount blocking %.1f us/call, through the ring %.1f us/call" % (
    timed(lambda: a.and_flag(b), 50) * 1e3, timed(ring4, 50) * 1e3 / 4, timed(lambda: a.count(), 50) * 1e3, timed(ring4c, 50) * 1e3 / 4), flush=True)
small = ctx.bits(1 << 16)
small.one()
print("2^16-bit bitset (one pack): popcount blocking %.1f us/call, through the ring %.1f us/call" % (
    timed(lambda: small.count(), 200) * 1e3,
    timed(lambda: ([small.count_async() for _ in range(4)], [ctx.bits_collect() for _ in range(4)]), 200) * 1e3 / 4), flush=True)
