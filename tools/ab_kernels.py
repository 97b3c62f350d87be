#!/usr/bin/env python3
"""A/B two builds of libknoxgpu.so on IDENTICAL device buffers (columns and bitset are allocated --
and placed -- once by build A and borrowed by build B through knox_col_wrap / knox_bits_wrap), so
that differences are kernel differences, not placement or box differences.  Alternates A/B `rounds`
times per case and prints the per-build median GB/s.
usage: ab_kernels.py <libB.so> [rows_log2=28] [rounds=3]      (A = the in-tree product library)
       ab_kernels.py env:NAME=VALUE [rows_log2] [rounds]      (B = the same library, its context opened with that environment
                                                               variable set, e.g. env:KNOX_DEFER=0)
       ab_kernels.py envrun:NAME=VALUE [rows_log2] [rounds]   (B = the same library and context kind, the variable set while B's
                                                               scans are enqueued, e.g. envrun:KNOX_ITERS=4)"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
import knoxdb_amd._lib as LL
from knoxdb_amd import And, Leaf, Or
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

lib_b = sys.argv[1]
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, pack = 1 << lg, 1 << 20
A = LL.lib()
env_b = envrun_b = None
if lib_b.startswith("env:"):
    env_b = lib_b[4:].split("=", 1)
    Bl = A
elif lib_b.startswith("envrun:"):   # the variable is set while B's scans are enqueued (knobs the library reads per call, e.g. KNOX_ITERS)
    envrun_b = lib_b[7:].split("=", 1)
    Bl = A
else:
    Bl = C.CDLL(lib_b)
    for name, (res, args) in LL.SIGNATURES.items():
        fn = getattr(Bl, name)
        fn.restype, fn.argtypes = res, args

ctxA = K.Context(0)
spec = {"i64": (B.Int64, 8, -(1 << 40), 1 << 41, 0), "i64b": (B.Int64, 8, -(1 << 40), 1 << 41, 0), "i64c": (B.Int64, 8, -(1 << 40), 1 << 41, 0),
        "i64d": (B.Int64, 8, -(1 << 40), 1 << 41, 0), "f64": (B.Float64, 8, 0, 0, 2), "f64b": (B.Float64, 8, 0, 0, 2),
        "i32": (B.Int32, 4, -(1 << 30), 1 << 31, 0), "f32": (B.Float32, 4, 0, 0, 2), "i16": (B.Int16, 2, -(1 << 15), 1 << 16, 0),
        "u8": (B.Uint8, 1, 0, 256, 0)}
colsA = {k: ctxA.generate(bt, kind, 2 + i, n, lo=lo, span=span, pack_rows=pack) for i, (k, (bt, w, lo, span, kind)) in enumerate(spec.items())}
bitsA = ctxA.bits(n, for_col=colsA["i64"])
ctxA.sync()
LL._lib = Bl
if env_b:
    os.environ[env_b[0]] = env_b[1]
ctxB = K.Context(0)
if env_b:
    os.environ.pop(env_b[0], None)
ctxB.set_placement(False)
colsB = {k: ctxB.wrap(spec[k][0], colsA[k].device_ptr, n, pack) for k in spec}
LL._lib = A
ptr = bitsA.device_ptr
LL._lib = Bl
bitsB = K.engine.Bits(ctxB, n, wrap_ptr=ptr)
LL._lib = A

ALLA = [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.First, 0), (R.Last, 0)]
cases = [
    ("i64 Between+sum", ["i64"], lambda: And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), [(R.Sum, 0)], 8.125),
    ("i64 Between+ALL", ["i64"], lambda: And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), ALLA, 8.125),
    ("f64 Lt+sum/min/max/mean", ["f64"], lambda: And(Leaf(0, F.Lt, 0.25)), [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)], 8.125),
    ("f64 x2 AND + min/max/mean (cfg 3)", ["f64", "f64b"], lambda: And(Leaf(0, F.Ge, 0.25), Leaf(1, F.Lt, 0.5)), [(R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.Sum, 0), (R.Count, 0)], 16.125),
    ("i64 x2 AND + sum", ["i64", "i64b"], lambda: And(Leaf(0, F.Gt, 0), Leaf(1, F.Lt, 0)), [(R.Sum, 0)], 16.125),
    ("i32 Between+sum", ["i32"], lambda: And(Leaf(0, F.Range, -(1 << 28), 1 << 28)), [(R.Sum, 0)], 4.125),
    ("f32 Lt+sum/min/max/mean", ["f32"], lambda: And(Leaf(0, F.Lt, 0.25)), [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)], 4.125),
    ("i32 Between+ALL", ["i32"], lambda: And(Leaf(0, F.Range, -(1 << 28), 1 << 28)), ALLA, 4.125),
    ("i16 Between+sum", ["i16"], lambda: And(Leaf(0, F.Range, -(1 << 13), 1 << 13)), [(R.Sum, 0)], 2.125),
    ("i16 Between+ALL", ["i16"], lambda: And(Leaf(0, F.Range, -(1 << 13), 1 << 13)), ALLA, 2.125),
    ("i32 0 % selected + ALL", ["i32"], lambda: And(Leaf(0, F.Range, (1 << 30) + 5, (1 << 30) + 6)), ALLA, 4.125),   # min / max pre-filter with nothing to adopt
    ("i16 0 % selected + ALL", ["i16"], lambda: And(Leaf(0, F.Range, 1 << 14, (1 << 14) - 1)), ALLA, 2.125),
    ("i16 Between (count)", ["i16"], lambda: And(Leaf(0, F.Range, -(1 << 13), 1 << 13)), [], 2.125),
    ("u8 Between (count)", ["u8"], lambda: And(Leaf(0, F.Range, 64, 127)), [], 1.125),
    ("u8 Between+sum", ["u8"], lambda: And(Leaf(0, F.Range, 64, 127)), [(R.Sum, 0)], 1.125),
    ("u8 Between+ALL", ["u8"], lambda: And(Leaf(0, F.Range, 64, 127)), ALLA, 1.125),
    ("i64 x2 OR (count)", ["i64", "i64b"], lambda: Or(Leaf(0, F.Gt, 1 << 39), Leaf(1, F.Lt, -(1 << 39))), [], 16.125),
    ("f64 x2 + i64 x1: 3 leaves + sum", ["f64", "f64b", "i64"], lambda: And(Leaf(0, F.Ge, 0.25), Leaf(1, F.Lt, 0.5), Leaf(2, F.Gt, 0)), [(R.Sum, 2)], 24.125),
    # three / four leaves with every aggregate (round 4: these instantiations kept their loads in scratch memory)
    ("i64 x3: 3 leaves + ALL", ["i64", "i64b", "i64c"], lambda: And(Leaf(0, F.Gt, -(1 << 39)), Or(Leaf(1, F.Lt, 0), Leaf(2, F.Ge, 0))), ALLA, 24.125),
    ("f64 x2 + i64 x1: 3 leaves + min/max/mean", ["f64", "f64b", "i64"], lambda: And(Leaf(0, F.Ge, 0.25), Leaf(1, F.Lt, 0.5), Leaf(2, F.Gt, 0)), [(R.Min, 0), (R.Max, 0), (R.Mean, 0)], 24.125),
    ("i64 x4: 4 leaves + ALL", ["i64", "i64b", "i64c", "i64d"], lambda: And(Or(Leaf(0, F.Gt, 0), Leaf(1, F.Lt, 0)), Or(Leaf(2, F.Ge, 0), Leaf(3, F.Lt, 0))), ALLA, 32.125),
]


def run(lib, ctx, cols, bits, names, tree, aggs, bpr, steps=16):
    LL._lib = lib
    try:
        if envrun_b and ctx is ctxB:
            os.environ[envrun_b[0]] = envrun_b[1]
        plan = ctx.plan(tree(), [cols[k] for k in names], aggs=aggs)
        ctx.L.knox_prof_enable(ctx.h, 1)
        ctx.scan_async(plan, bits_out=bits)
        for _ in range(steps - 1):
            ctx.scan_async(plan, bits_out=bits)
            ctx.scan_collect(plan, 0)
        res = ctx.scan_collect(plan, 0)
        ms = (C.c_float * 512)()
        k = C.c_int()
        ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
        ctx.L.knox_prof_enable(ctx.h, 0)
        v = [ms[i] for i in range(k.value)]
        return n * bpr / statistics.median(v[len(v) // 2:]) / 1e6, res.count
    finally:
        if envrun_b:
            os.environ.pop(envrun_b[0], None)
        LL._lib = A


run(A, ctxA, colsA, bitsA, *cases[0][1:], steps=60)
print("%-28s %10s %10s %8s" % ("case", "A GB/s", "B GB/s", "B/A"))
for name, names, tree, aggs, bpr in cases:
    ra, rb = [], []
    for _ in range(rounds):
        ga, ca = run(A, ctxA, colsA, bitsA, names, tree, aggs, bpr)
        gb, cb = run(Bl, ctxB, colsB, bitsB, names, tree, aggs, bpr)
        assert ca == cb, "builds disagree on %s: %d vs %d" % (name, ca, cb)
        ra.append(ga)
        rb.append(gb)
    a, b = statistics.median(ra), statistics.median(rb)
    print("%-28s %10.0f %10.0f %8.3f" % (name, a, b, b / a))
