#!/usr/bin/env python3
"""IN / NOT IN set matchers (numInSetMatcher, match_num.go:672-808) as scan leaves: whole scan calls and kernel time per set size.
usage: bench_inset.py [rows_log2=28]"""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n, pack = 1 << lg, 1 << 20
ctx = K.Context(0)
rng = np.random.default_rng(5)
for bt, w, lo, span in ((B.Int64, 8, 0, 1 << 20), (B.Int32, 4, 0, 1 << 20), (B.Uint16, 2, 0, 1 << 16)):
    col = ctx.generate(bt, 0, 2, n, lo=lo, span=span, pack_rows=pack)
    bits = ctx.bits(n, for_col=col)
    for nk in (1, 4, 16, 64, 1024, 65536):
        keys = np.unique(rng.integers(lo, lo + span, size=nk)).tolist()
        for mode in (F.In, F.NotIn):
            plan = ctx.plan(And(Leaf(0, mode, keys)), [col], aggs=[])
            res = ctx.scan_async(plan, bits_out=bits) or ctx.scan_collect(plan, 0)   # warm
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(8):
                ctx.scan_async(plan, bits_out=bits)
                res = ctx.scan_collect(plan, 0)
            t = (time.perf_counter() - t0) / 8 * 1e3   # whole call: the matcher, the merge into the result bitset, the count
            print("%-7s %-6s %6d keys: %8.4f ms  %7.1f G rows/s  %7.0f GB/s   count %d" % (bt.name, mode.name, len(keys), t, n / t / 1e6, n * (w + 0.125) / t / 1e6, res.count), flush=True)
    col.free(); bits.free()
