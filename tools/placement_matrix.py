#!/usr/bin/env python3
"""Diagnostic: steady-state kernel GB/s per (column placement x wave-iterations-per-tile x WGs/CU)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 1 << lg
ctx = K.Context(0)
cols = []
for i in range(ncol):
    if len(sys.argv) > 4:   # alternate allocation flags per column: argv[4] = flag for odd columns
        os.environ["KNOX_COL_MALLOC_FLAGS"] = sys.argv[4] if (i & 1) else "0"
    cols.append(ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20))
bits = ctx.bits(n)
ctx.L.knox_prof_enable(ctx.h, 1)
def run(col, steps):
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [col], aggs=[(ReducerFunc.Sum, 0)])
    ctx.scan_async(plan, bits_out=bits)
    for i in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits); ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = [ms[i] for i in range(k.value)]
    return n * 8.125 / statistics.median(v[len(v) // 2:]) / 1e6
run(cols[0], 60)
mode = sys.argv[3] if len(sys.argv) > 3 else "iters"
if mode == "iters":
    configs = [(it, bpc, 1) for bpc in (2, 3) for it in (1, 2, 4, 8, 16)]
elif mode == "xcd":  # default vs XCD-contiguous tile mapping (variant 5) at iters 1, 2, 4
    configs = [(it, 2, v) for it in (1, 2, 4) for v in (1, 5)]
else:  # store flavours at iters=2, 2 WG/CU: nt (1), none (2), plain (3), write-through (4)
    configs = [(2, 2, v) for v in (1, 2, 3, 4)]
print("col | " + " ".join("i%-2d/b%d/v%d" % c for c in configs))
res = []
for ci, col in enumerate(cols):
    row = []
    for it, bpc, var in configs:
        os.environ["KNOX_ITERS"] = str(it)
        ctx.set_tuning(bpc)
        if var != 1:
            ctx.diag_set_variant(var)  # diagnostics library only
        row.append(run(col, 14))
    res.append(row)
    print("%3d | " % ci + " ".join("%6.0f" % x for x in row))
print("min | " + " ".join("%6.0f" % min(r[i] for r in res) for i in range(len(configs))))
print("med | " + " ".join("%6.0f" % statistics.median(r[i] for r in res) for i in range(len(configs))))
