#!/usr/bin/env python3
"""Scan on bit-packed containers vs the materialized column: kernel time, rows/s, GB/s of packed bytes.
usage: bench_packed.py [bits=20] [rows_log2=30]"""
import ctypes as C, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
from oracle import oracle as O  # container encoder only (test infrastructure): builds the input

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pack = 1 << 20
n_packs = (1 << lg) // pack
rng = np.random.default_rng(5)
uniq = min(n_packs, 16)  # distinct packs, repeated: the container encoder on the host is the slow part
base = 1_700_000_000_000
conts = []
for p in range(uniq):
    v = base + rng.integers(0, 1 << bits, size=pack, dtype=np.int64)
    v[0], v[-1] = base, base + (1 << bits) - 1
    conts.append(bytes(O.bitpack_container_store(O.I64, v)))
conts = [conts[p % uniq] for p in range(n_packs)]
ctx = K.Context(0)
n = n_packs * pack
pk = ctx.packed_upload(BlockType.Int64, conts, pack)
a, b = base + (1 << bits) // 4, base + (1 << bits) // 2
ctx.L.knox_prof_enable(ctx.h, 1)


def kernel_ms(fn, steps=20):
    for _ in range(5):
        fn()
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    for _ in range(steps):
        fn()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    return statistics.median([ms[i] for i in range(k.value)])


col = ctx.upload_raw_containers(BlockType.Int64, conts, pack)
bits_buf = ctx.bits(n, for_col=col)
out = {"bits": bits, "rows": n, "packed_bytes": pk.nbytes}
for name, kw in (("count", {}), ("count+sum", {"want_sum": True}), ("count+sum+bitset", {"want_sum": True, "bits_out": bits_buf})):
    ms = kernel_ms(lambda: pk.scan(FilterMode.Range, a, b, **kw))
    bpr = bits / 8 + (0.125 if "bits_out" in kw else 0)
    out["packed " + name] = {"kernel_ms": ms, "rows_per_s": n / ms * 1e3, "GBs_packed": n * bpr / ms / 1e6}
plan = ctx.plan(And(Leaf(0, FilterMode.Range, a, b)), [col], aggs=[(R.Sum, 0)])


def mat():
    ctx.scan_async(plan, bits_out=bits_buf)
    ctx.scan_collect(plan, 0)


ms = kernel_ms(mat)
out["materialized count+sum+bitset"] = {"kernel_ms": ms, "rows_per_s": n / ms * 1e3, "GBs": n * 8.125 / ms / 1e6}
c1 = pk.scan(FilterMode.Range, a, b)[0]
c2 = ctx.scan(And(Leaf(0, FilterMode.Range, a, b)), [col]).count
out["counts_agree"] = c1 == c2
print(json.dumps(out, indent=1))
