#!/usr/bin/env python3
"""Does the rate of a TWO-column scan depend on the DISTANCE between the two columns inside one placement class?
One 18 GiB float64 region of one arena zone; column A = its first 8 GiB, column B = 8 GiB starting at 8 GiB + skew, for a
list of skews (0 = what two back-to-back arena allocations give).  Read-only form of the cfg-3 tree and the full scan
(bitset verified against both views), each the median of the second half of 16 launches, every skew visited twice."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K                                                # noqa: E402
from knoxdb_amd import And, Leaf                                      # noqa: E402
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R   # noqa: E402

n, pack = 1 << 30, 1 << 20
GIB = 1 << 30
ctx = K.Context(0)
ctx.set_arena(100 << 30)
ctx.reserve(19 * GIB)
big = ctx.generate(B.Float64, 2, 5, (18 * GIB) // 8, pack_rows=pack)   # uniform [0, 1)
base = big.device_ptr
print("region at", hex(base), "class", big.placement_class(), flush=True)
tree = And(Leaf(0, F.Ge, 0.25), Leaf(1, F.Lt, 0.5))


def ms(p, bits, steps=16):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(p, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(p, bits_out=bits)
        ctx.scan_collect(p, 0)
    ctx.scan_collect(p, 0)
    v = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, v, 512, C.byref(k))
    ctx.L.knox_prof_enable(ctx.h, 0)
    x = [v[i] for i in range(k.value)]
    return statistics.median(x[len(x) // 2:])


A = ctx.wrap(B.Float64, base, n, pack_rows=pack)
skews = [0, 4 << 10, 64 << 10, 1 << 20, (1 << 20) + (4 << 10), 33 << 20, 128 << 20, (256 << 20) + (64 << 10), 512 << 20, (1 << 30) + (1 << 20)]
if len(sys.argv) > 1 and sys.argv[1] == "fine":
    K4 = 4 << 10
    skews = [0, 256, 1024, 2048, K4, 2 * K4, 3 * K4, 4 * K4, 5 * K4, 8 * K4, 16 * K4, 17 * K4, 32 * K4, 64 * K4, 65 * K4, 128 * K4, 256 * K4, 257 * K4, 512 * K4,
             (2 << 20) + K4, (4 << 20) + K4, (8 << 20) + K4, (16 << 20) + K4, (64 << 20) + K4, (128 << 20) + K4, (256 << 20) + K4, 256 << 20, (256 << 20) + (64 << 10) + K4,
             (384 << 20), (384 << 20) + K4, (768 << 20) + K4]
bits = None
for rep in range(2):
    for sk in skews:
        Bc = ctx.wrap(B.Float64, base + 8 * GIB + sk, n, pack_rows=pack)
        if bits is None:
            bits = ctx.bits(n, for_col=[A, Bc])
            print("bitset pair check", bits.pair_check, flush=True)
        ro = ctx.plan(tree, [A, Bc], aggs=[])
        full = ctx.plan(tree, [A, Bc], aggs=[(R.Min, 0), (R.Max, 0), (R.Mean, 0)])
        if rep == 0 and sk == 0:
            ms(ro, None, 40)
        t_ro, t_full = ms(ro, None), ms(full, bits)
        print("skew %10d B (%8.3f MiB): read-only %.4f ms = %.0f GB/s   full %.4f ms = %.0f GB/s (%.4f of peak)" %
              (sk, sk / 2**20, t_ro, n * 16 / t_ro / 1e6, t_full, n * 16.125 / t_full / 1e6, n * 16.125 / t_full / 1e6 / 8000), flush=True)
        Bc.free()
