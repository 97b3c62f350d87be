#!/usr/bin/env python3
"""Secondary configs of BASELINE.json (not the headline): steady-state kernel time + achieved GB/s.
  cfg3: 2^30 float64 rows x 2 columns, `A >= 256 AND B < 0.5`, min/max/mean(A) + count
  cfg4: 2^30 uint64 ids + 2^30 32-byte []byte keys (one StringPool per pack), per-pack bloom (factor 2) probe then EQ, count
"""
import argparse, ctypes as C, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", type=int, default=3)
ap.add_argument("--rows-log2", type=int, default=30)
ap.add_argument("--steps", type=int, default=40)
args = ap.parse_args()
n, pack = 1 << args.rows_log2, 1 << 20
ctx = K.Context(0)
ctx.L.knox_prof_enable(ctx.h, 1)

def timed(plan, bits, steps, pack_counts=True, skip=None):
    ctx.scan_async(plan, bits_out=bits, pack_counts=pack_counts, pack_skip=skip)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits, pack_counts=pack_counts, pack_skip=skip)
        res = ctx.scan_collect(plan, n // pack if pack_counts else 0)
    res = ctx.scan_collect(plan, n // pack if pack_counts else 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = [ms[i] for i in range(k.value)]
    return statistics.median(v[len(v) // 3:]), res

if args.cfg == 0:
    # PCIe-inclusive ingest: pageable host memory -> pinned ring -> HBM (knox_col_upload), then one scan
    m = 1 << min(args.rows_log2, 28)
    host = np.arange(m, dtype=np.int64)
    t0 = time.perf_counter(); col = ctx.upload(host, pack); ctx.sync(); t_up = time.perf_counter() - t0
    bits = ctx.bits(m)
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, 5, m // 4)), [col], aggs=[(R.Sum, 0)])
    t0 = time.perf_counter(); ctx.scan_async(plan, bits_out=bits); res = ctx.scan_collect(plan, 0); t_scan = time.perf_counter() - t0
    t0 = time.perf_counter(); b, c = ctx.cmp(BlockType.Int64, FilterMode.Lt, host[:1 << 16], 127); t_cmp = time.perf_counter() - t0
    pin = ctx.host_array(m, np.int64)
    pin[:] = host
    t0 = time.perf_counter(); colp = ctx.upload(pin, pack); ctx.sync(); t_pin = time.perf_counter() - t0
    print(json.dumps({"cfg": "pcie-pinned (knox_host_alloc)", "rows": m, "upload_s": t_pin, "upload_GBs": m * 8 / t_pin / 1e9,
                      }))
    print(json.dumps({"cfg": "pcie", "rows": m, "upload_s": t_up, "upload_GBs": m * 8 / t_up / 1e9, "first_scan_s": t_scan,
                      "pcie_inclusive_rows_per_s": m / (t_up + t_scan), "knox_cmp_2^16_rows_s": t_cmp}))
elif args.cfg == 3:
    a = ctx.generate(BlockType.Float64, 1, 3, n, pack_rows=pack)
    b = ctx.generate(BlockType.Float64, 2, 4, n, pack_rows=pack)
    bits = ctx.bits(n, for_col=a)
    tree = And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5))
    for name, aggs in (("count", [(R.Count, 0)]), ("min/max/mean", [(R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.Count, 0)])):
        plan = ctx.plan(tree, [a, b], aggs=aggs)
        ms, res = timed(plan, bits, args.steps)
        gbs = n * 16.125 / ms / 1e6
        vals = [res.value(i)[0] for i in range(len(aggs))]
        print(json.dumps({"cfg": 3, "aggs": name, "kernel_ms": ms, "rows_per_s": n / ms * 1e3, "achieved_GBs": gbs,
                          "frac_8TBs": gbs / 8000, "count": res.count, "values": [float(v) for v in vals]}))
    # the read-only form of the same tree (no bitset, no per-pack counts): the two-stream ceiling on this box
    plan = ctx.plan(tree, [a, b], aggs=[])
    ctx.L.knox_prof_enable(ctx.h, 1)
    for _ in range(10):
        ctx.scan_async(plan); ctx.scan_collect(plan, 0)
    import ctypes as C
    kms, kn = C.c_double(), C.c_int()
    ctx.L.knox_prof_read(ctx.h, C.byref(kms), C.byref(kn))
    ctx.L.knox_prof_enable(ctx.h, 0)
    ro = kms.value / max(1, kn.value)
    print(json.dumps({"cfg": 3, "aggs": "read-only ceiling (count, no bitset)", "kernel_ms": ro, "achieved_GBs": n * 16.0 / ro / 1e6,
                      "bits_pair_check": bits.pair_check}))
elif args.cfg == 6:
    # time-window aggregation (next row f1): 2^30 rows, time = row id, windows of 2^16 / 2^10 rows
    tcol = ctx.generate(BlockType.Int64, 3, 0, n, lo=1_000_000, pack_rows=pack)
    vcol = ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
    bits = ctx.bits(n)
    ctx.scan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [vcol], bits_out=bits)
    for step in (1 << 16, 1 << 10):
        nb = n // step
        for sorted_flag in (False, True):
            ctx.reduce_buckets(tcol, vcol, 1_000_000, step, nb, mask=bits, assume_sorted=sorted_flag)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                st = ctx.reduce_buckets(tcol, vcol, 1_000_000, step, nb, mask=bits, assume_sorted=sorted_flag)
            dt = (time.perf_counter() - t0) / reps
            byts = n * (8.125 + (0 if sorted_flag else 8))
            print(json.dumps({"cfg": "buckets", "windows": nb, "rows_per_window": step, "order_check": not sorted_flag, "call_ms": dt * 1e3,
                              "rows_per_s": n / dt, "GBs_incl_host": byts / dt / 1e9, "count0": st[0].count}))
elif args.cfg == 4:
    ids = ctx.generate(BlockType.Uint64, 0, 5, n, lo=0, span=0, pack_rows=pack)   # raw 64-bit ids
    bits = ctx.bits(n, for_col=ids)
    t0 = time.perf_counter()
    bloom = ctx.bloom_build_u64(ids, 2)
    ctx.sync()
    t_build = time.perf_counter() - t0
    present = ids.download(777 * pack + 12345, 1)[0]
    for name, key in (("present", present), ("absent", np.uint64(0x123456789ABCDEF))):
        h = K.hash_value(BlockType.Uint64, key)
        t0 = time.perf_counter()
        maybe = bloom.probe(h)
        t_probe = time.perf_counter() - t0
        plan = ctx.plan(And(Leaf(0, FilterMode.Equal, key)), [ids], aggs=[(R.Count, 0)])
        ms_full, r_full = timed(plan, bits, args.steps)
        ms_pruned, r_pruned = timed(plan, bits, args.steps, skip=(1 - maybe).astype(np.uint8))
        scanned = int(maybe.sum())
        print(json.dumps({"cfg": 4, "key": name, "packs_maybe": scanned, "packs": n // pack, "count_full": r_full.count,
                          "count_pruned": r_pruned.count, "kernel_ms_full": ms_full, "kernel_ms_pruned": ms_pruned,
                          "full_GBs": n * 8.125 / ms_full / 1e6, "pruned_GBs_over_scanned_packs": scanned * pack * 8.125 / ms_pruned / 1e6 if scanned else None,
                          "effective_rows_per_s_incl_pruned": n / ms_pruned * 1e3, "bloom_build_s": t_build, "bloom_probe_s": t_probe,
                          "filter_bytes": int(ctx.L.knox_bloom_bytes(bloom.h))}))
    # the []byte leg (BASELINE configs[3]): 32-byte `addr` values from 2^24 distinct keys, each key inside one window of
    # 4 packs; one StringPool per pack; per-pack bloom over the bytes column (XXH3-64 of every row, on the device)
    ids.free(); bloom.free()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    t0 = time.perf_counter()
    addr = ctx.generate_bytes(6, n, pack, 32, 16, 4)
    ctx.sync()
    bits.free()
    bits = ctx.bits(n, for_col=addr)   # verified against the pool of the bytes column
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    bloom = ctx.bloom_build(addr, 2)
    ctx.sync()
    t_build = time.perf_counter() - t0
    ptr, buf = ctx.download_bytes_pack(addr, min(777, n // pack - 1))
    present = bytes(buf[12345 * 32:12346 * 32])
    absent = bytes(range(32))
    B_ROW = 8 + 32 + 0.125
    for name, key in (("present", present), ("absent", absent)):
        t0 = time.perf_counter()
        maybe = bloom.probe(K.hash_bytes(key))
        t_probe = time.perf_counter() - t0
        plan = ctx.plan(And(Leaf(0, FilterMode.Equal, key)), [addr], aggs=[])
        ms_full, r_full = timed(plan, bits, args.steps)
        ms_pruned, r_pruned = timed(plan, bits, args.steps, skip=(1 - maybe).astype(np.uint8))
        scanned = int(maybe.sum())
        print(json.dumps({"cfg": 4, "leg": "bytes", "key": name, "packs_maybe": scanned, "packs": n // pack, "count_full": r_full.count,
                          "count_pruned": r_pruned.count, "rows_scanned_pruned": r_pruned.rows_scanned, "bytes_per_row": B_ROW,
                          "kernel_ms_full": ms_full, "kernel_ms_pruned": ms_pruned, "full_GBs": n * B_ROW / ms_full / 1e6,
                          "full_frac_8TBs": n * B_ROW / ms_full / 1e6 / 8000, "full_rows_per_s": n / ms_full * 1e3,
                          "pruned_GBs_over_scanned_packs": scanned * pack * B_ROW / ms_pruned / 1e6 if scanned else None,
                          "effective_rows_per_s_incl_pruned": n / ms_pruned * 1e3, "generate_s": t_gen, "bloom_build_s": t_build,
                          "bloom_probe_s": t_probe, "filter_bytes": int(ctx.L.knox_bloom_bytes(bloom.h)),
                          "pool_bytes": int(ctx.L.knox_col_pool_bytes(addr.h))}))
