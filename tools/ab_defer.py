#!/usr/bin/env python3
"""A/B of the plain scans' DEFERRED bitset bursts (knox_scan.hip kDefer: every workgroup parks the words of up to 120 consecutive
tiles in LDS and writes them together, all workgroups at the same rounds) against bursts tile by tile (a context opened with
KNOX_DEFER=0), on IDENTICAL buffers: one 8 GiB int64 column, a bitset of another placement class (fast pair) and 128 MiB inside
a second column of the column's own class (slow pair).  Other builds of the library can be added: ab_defer.py [<lib.so> ...]"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K                                                # noqa: E402
import knoxdb_amd._lib as LL                                          # noqa: E402
from knoxdb_amd import And, Leaf                                      # noqa: E402
from knoxdb_amd.engine import Bits                                    # noqa: E402
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R   # noqa: E402

n, pack = 1 << 30, 1 << 20
A = LL.lib()
libs = [("bursts tile by tile", A), ("deferred, unaligned", A), ("deferred (product)", A)]
for path in sys.argv[1:]:
    L = C.CDLL(path)
    for name, (res, args) in LL.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    libs.append((os.path.basename(path), L))

ctxA = K.Context(0)
ctxA.set_arena(120 << 30)
ctxA.reserve(11 << 30)
col = ctxA.generate(B.Int64, 0, 2, n, pack_rows=pack, lo=-(1 << 40), span=1 << 41)
neighbour = ctxA.generate(B.Int64, 3, 0, 1 << 28, lo=0, pack_rows=pack)
fast = ctxA.bits(n, for_col=col)
p_col, p_fast, p_slow = col.device_ptr, fast.device_ptr, neighbour.device_ptr + (1 << 30)
print("fast bitset pair check", fast.pair_check, flush=True)
ctxA.sync()


def ms(ctx, p, bits, steps=16):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(p, bits_out=bits, pack_counts=True)
    for _ in range(steps - 1):
        ctx.scan_async(p, bits_out=bits, pack_counts=True)
        ctx.scan_collect(p, n // pack)
    res = ctx.scan_collect(p, n // pack)
    v = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, v, 512, C.byref(k))
    ctx.L.knox_prof_enable(ctx.h, 0)
    x = [v[i] for i in range(k.value)]
    return statistics.median(x[len(x) // 2:]), res


ctxs = []
for name, L in libs:
    LL._lib = L
    if name == "bursts tile by tile":
        os.environ["KNOX_DEFER"] = "0"
    elif name == "deferred, unaligned":
        os.environ["KNOX_DEFER"] = "2"
    else:
        os.environ.pop("KNOX_DEFER", None)
    ctx = K.Context(0)
    ctx.set_placement(False)
    c = ctx.wrap(B.Int64, p_col, n, pack)
    bf, bs = Bits(ctx, n, wrap_ptr=p_fast), Bits(ctx, n, wrap_ptr=p_slow)
    plan = ctx.plan(And(Leaf(0, F.Range, (-(1 << 38), 1 << 38))), [c], aggs=[(R.Sum, 0), (R.Count, 0)])
    ro = ctx.plan(And(Leaf(0, F.Range, (-(1 << 38), 1 << 38))), [c], aggs=[])
    ctxs.append((name, L, ctx, plan, ro, bf, bs))
ref = None
for rep in range(3):
    for name, L, ctx, plan, ro, bf, bs in ctxs:
        LL._lib = L
        t_f, r_f = ms(ctx, plan, bf)
        words = bf.bytes().view("uint64")
        sig = (int(r_f.count), int(r_f.states[0].isum), int(words.sum(dtype="uint64")), int(r_f.pack_counts.sum()))
        t_s, r_s = ms(ctx, plan, bs)
        sig_s = (int(r_s.count), int(r_s.states[0].isum), int(bs.bytes().view("uint64").sum(dtype="uint64")), int(r_s.pack_counts.sum()))
        t_r, _ = ms(ctx, ro, None)
        if ref is None:
            ref = sig
        assert sig == ref and sig_s == ref, (name, sig, sig_s, ref)
        print("%-26s fast pair %.4f ms (%.3f of peak)   slow pair %.4f ms (%.3f)   read-only %.4f ms" %
              (name, t_f, n * 8.125 / t_f / 8e9, t_s, n * 8.125 / t_s / 8e9, t_r), flush=True)
LL._lib = A
