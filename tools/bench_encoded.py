#!/usr/bin/env python3
"""Scan on dictionary [5] / run-end [3] containers kept encoded in HBM (knox_enc_*) against the same containers decoded
to a materialized column: kernel time (the library's HIP events), rows/s, GB/s and fraction of the 8 TB/s peak by the
ENCODED bytes the kernel has to move (dictionary: 2 B of codes + 1/8 B of bitset per row; run-end: (W + 4) B per run +
1/8 B per row).  usage: bench_encoded.py [rows_log2=30] [dict_entries=4096] [mean_run=64]"""
import ctypes as C, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
from oracle import oracle as O  # container encoder only (test infrastructure): builds the input

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_dict = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mean_run = int(sys.argv[3]) if len(sys.argv) > 3 else 64
pack = 1 << 20
n_packs = max(1, (1 << lg) // pack)
n = n_packs * pack
rng = np.random.default_rng(5)
uniq = min(n_packs, 8)   # distinct packs, repeated: building containers on the host is the slow part
base = 1_700_000_000_000
ctx = K.Context(0)
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


out = {"rows": n}
for kind in ("dictionary", "run-end"):
    conts = []
    for p in range(uniq):
        if kind == "dictionary":
            dvals = np.sort(base + rng.choice(1 << 30, size=n_dict, replace=False)).astype(np.int64)
            codes = rng.integers(0, n_dict, size=pack).astype(np.uint16)
            conts.append(bytes([5]) + bytes(O.raw_container_store(O.I64, dvals)) + bytes(O.raw_container_store(O.U16, codes)))
        else:
            runs = np.maximum(1, rng.geometric(1.0 / mean_run, size=2 * pack // mean_run))
            runs = runs[np.cumsum(runs) <= pack]
            runs = np.append(runs, pack - int(runs.sum())) if runs.sum() < pack else runs
            vals = (base + rng.integers(0, 1 << 30, size=runs.size)).astype(np.int64)
            ends = (np.cumsum(runs) - 1).astype(np.uint32)
            conts.append(bytes([3]) + bytes(O.raw_container_store(O.I64, vals)) + bytes(O.raw_container_store(O.U32, ends)))
    conts = [conts[p % uniq] for p in range(n_packs)]
    enc = ctx.enc_upload(BlockType.Int64, conts, pack)
    col = ctx.upload_raw_containers(BlockType.Int64, conts, pack)
    bits_buf = ctx.bits(n, for_col=col)
    a, b = base + (1 << 28), base + (1 << 29)      # ~25 % of the value range
    res = {"resident_bytes": enc.nbytes, "container_bytes": enc.container_bytes, "materialized_bytes": n * 8}
    for name, kw in (("count", {}), ("count+sum", {"want_sum": True}), ("count+sum+bitset", {"want_sum": True, "bits_out": bits_buf})):
        ms = kernel_ms(lambda: enc.scan(FilterMode.Range, a, b, **kw))
        moved = enc.nbytes + (n / 8 if "bits_out" in kw else 0)      # what the scan has to read + write
        t0 = time.perf_counter()
        for _ in range(20):
            enc.scan(FilterMode.Range, a, b, **kw)
        call_ms = (time.perf_counter() - t0) / 20 * 1e3              # the whole blocking call: launches, result copy, synchronisation
        res[name] = {"kernel_ms": ms, "call_ms": call_ms, "rows_per_s": n / ms * 1e3, "GBs_encoded": moved / ms / 1e6, "frac_of_8TBs": moved / ms / 1e6 / 8000.0}
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, a, b)), [col], aggs=[(R.Sum, 0)])

    def mat():
        ctx.scan_async(plan, bits_out=bits_buf)
        ctx.scan_collect(plan, 0)

    ms = kernel_ms(mat)
    res["materialized count+sum+bitset"] = {"kernel_ms": ms, "rows_per_s": n / ms * 1e3, "GBs": n * 8.125 / ms / 1e6}
    c1, st, _ = enc.scan(FilterMode.Range, a, b, want_sum=True)
    r2 = ctx.scan(And(Leaf(0, FilterMode.Range, a, b)), [col], aggs=[(R.Sum, 0)])
    res["agree_with_materialized"] = bool(c1 == r2.count and st.isum == r2.states[0].isum)
    out[kind] = res
    for o in (enc, col, bits_buf):
        o.free()
print(json.dumps(out, indent=1))
