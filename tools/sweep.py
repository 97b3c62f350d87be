#!/usr/bin/env python3
"""Interleaved A/B sweep of the streaming scan kernel in ONE process (methodology rule 24):
blocks-per-CU x load variant x output mode, median / min kernel time from HIP events."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

ap = argparse.ArgumentParser()
ap.add_argument("--rows-log2", type=int, default=30)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--bpc", type=str, default="2,4,8,12,16,32")
ap.add_argument("--variants", type=str, default="1")
ap.add_argument("--modes", type=str, default="count,bits,bits+sum")
ap.add_argument("--dtype", type=str, default="i64")
args = ap.parse_args()

n = 1 << args.rows_log2
ctx = K.Context(0)
if args.dtype == "i64":
    col = ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20)
    tree = And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38))
    bpr = 8.125
elif args.dtype in ("i32", "i16", "u8", "f32"):
    bt, w, lo, span, a, b = {"i32": (BlockType.Int32, 4, -(1 << 30), 1 << 31, -(1 << 28), 1 << 28),
                             "i16": (BlockType.Int16, 2, -(1 << 15), 1 << 16, -(1 << 13), 1 << 13),
                             "u8": (BlockType.Uint8, 1, 0, 256, 64, 127),
                             "f32": (BlockType.Float32, 4, 0, 0, 0.25, 0.5)}[args.dtype]
    col = ctx.generate(bt, 2 if args.dtype == "f32" else 0, 2, n, lo=lo, span=span, pack_rows=1 << 20)
    tree = And(Leaf(0, FilterMode.Range, a, b))
    bpr = w + 0.125
else:
    col = ctx.generate(BlockType.Float64, 1, 3, n, pack_rows=1 << 20)
    tree = And(Leaf(0, FilterMode.Ge, 256.0))
    bpr = 8.125
bits = ctx.bits(n)
plans = {
    "count": (ctx.plan(tree, [col], aggs=[(ReducerFunc.Count, 0)]), None),
    "bits": (ctx.plan(tree, [col], aggs=[(ReducerFunc.Count, 0)]), bits),
    "bits+sum": (ctx.plan(tree, [col], aggs=[(ReducerFunc.Sum, 0)]), bits),
    "bits+all": (ctx.plan(tree, [col], aggs=[(ReducerFunc.Sum, 0), (ReducerFunc.Min, 0), (ReducerFunc.Max, 0), (ReducerFunc.Mean, 0)]), bits),
}
configs = [(int(b), int(v), m) for b in args.bpc.split(",") for v in args.variants.split(",") for m in args.modes.split(",")]
times = {c: [] for c in configs}
ctx.L.knox_prof_enable(ctx.h, 1)
for r in range(args.rounds + 1):
    for c in configs:
        bpc, var, mode = c
        ctx.set_tuning(bpc)
        if var != 1:
            ctx.diag_set_variant(var)  # diagnostics library only (KNOXGPU_SO=knoxdb_amd/diag/...)
        plan, b = plans[mode]
        ctx.scan_async(plan, bits_out=b)
        for _ in range(args.steps - 1):
            ctx.scan_async(plan, bits_out=b)   # one-deep pipeline: device never idles
            ctx.scan_collect(plan, 0)
        ctx.scan_collect(plan, 0)
        ms, k = C.c_double(), C.c_int()
        ctx.L.knox_prof_read(ctx.h, C.byref(ms), C.byref(k))
        if r > 0:
            times[c].append(ms.value / k.value)
print("%-6s %-4s %-10s %10s %10s %10s %8s" % ("bpc", "var", "mode", "median_ms", "min_ms", "GB/s(med)", "frac8T"))
for c in configs:
    med, mn = statistics.median(times[c]), min(times[c])
    gbs = n * ((bpr - 0.125) if c[2] == "count" else bpr) / (med * 1e-3) / 1e9
    print("%-6d %-4d %-10s %10.4f %10.4f %10.1f %8.3f" % (c[0], c[1], c[2], med, mn, gbs, gbs / 8000))
