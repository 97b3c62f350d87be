#!/usr/bin/env python3
"""Shuffled clock only (the stable radix sort path): 2^28 rows -> 1024 / 16384 windows, 25 % selected and unselected (mask=None),
values path.  The command behind a per-kernel table of exactly this shape (tools/prof_cmd.sh <tag> shuffled tools/bench_shuffled.py).
usage: bench_shuffled.py [rows_log2=28] [reps=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = K.Context(0)
n = 1 << lg
col = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20)
clock = ctx.generate(B.Int64, 0, 9, n, lo=1000, span=1 << 20, pack_rows=1 << 20)
bits = ctx.bits(n, for_col=col)
sel = int(os.environ.get("KNOX_SHUFFLED_SEL", "25"))   # per cent of the rows the bitset selects (values are uniform in [-2^40, 2^40))
half = (1 << 40) * sel // 100
ctx.scan(And(Leaf(0, F.Range, -half, half)), [col], bits_out=bits)


if os.environ.get("KNOX_SHUFFLED_DTYPE") == "f64":     # Sum over a float64 column (uniform in [0, 1)): the exact fixed-point adders
    vals = ctx.generate(B.Float64, 2, 5, n, pack_rows=1 << 20)
else:
    vals = col


FUNC = {"sum": R.Sum, "mean": R.Mean, "min": R.Min, "max": R.Max, "first": R.First, "last": R.Last, "count": R.Count, "var": R.Var, "std": R.Std}[
    os.environ.get("KNOX_SHUFFLED_FUNC", "sum")]


def timed(fn):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


only = os.environ.get("KNOX_SHUFFLED_ONLY")   # "masked" | "all": one variant (a profile of one shape)
for nb in [int(x) for x in os.environ.get("KNOX_SHUFFLED_NB", "1024,16384").split(",")]:   # window counts (powers of two up to 2^20)
    step = (1 << 20) // nb
    for m, tag in ((bits, "%d %% selected" % sel), (None, "all rows")):
        if only and (tag == "all rows") != (only == "all"):
            continue
        ms = timed(lambda: ctx.reduce_buckets_values(clock, vals, FUNC, 1000, step, nb, mask=m))
        print("2^%d rows, shuffled clock, %6d windows, %-13s: %6.2f ms (%.1f G rows/s) [path %d]" % (lg, nb, tag, ms, n / ms / 1e6, ctx.windows_last_path()), flush=True)
