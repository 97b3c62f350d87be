#!/usr/bin/env python3
"""bench.py -- headline benchmark: 1B-row int64 range filter + sum per GPU (BASELINE.json
configs[1]; configs[4] when launched with N > 1 ranks).

A "step" is one pass of the hot path over the whole resident shard: the fused scan kernel
evaluates `Between(-2^38, 2^38)` on 2^30 int64 rows (1024 packs of 2^20), writes the LSB-first
selection bitset, the per-pack popcounts and the wrapping int64 sum + count.  Inputs are
generated on the device (counter-based generator) and are resident in HBM when the
timed region starts.  For N > 1 each rank owns a contiguous pack range (no data-path
collective); only {sum, count} cross xGMI each step: ONE ncclAllReduce(int64 x 3) on the library's own
communicator (knox_comm_*, `--collective abi`, the default) or through torch.distributed (`--collective torch`).

Setup, outside the timed region and reported under "placement" in the JSON line: the context's
device-memory arena gets an explicit budget (KNOX_BENCH_ARENA_GIB, default 160: the headline maps ~13-17 GiB of it, the
secondary configs the rest), the column is generated into it, the bitset is allocated FOR that column (a write buffer
verified against it, knox_bits_alloc_for) and the pair is checked ONCE with the real kernel against its own read-only
form (before it the read-only form is sampled every 0.5 s until two consecutive samples agree within 0.4 % -- stability only,
never a target rate -- or 10 s have passed: a process that has just exited leaves the driver clearing its memory, which costs
every kernel ~5 % while it lasts; "placement.device_quiet_ms" holds the samples, "device_quiet_wait_s" the seconds waited and
"device_quiet_ended_by" "stable" or "timeout").  The timed region contains only the K steps.

After the timed region (all outside it): the read-only ceiling, the 0 % / 100 % selectivity runs of the same scan, the
in-run parity against the oracle (bitset memcmp on a 64-pack sample, XOR checksum of ALL bitset words, count and
wrapping sum of the whole run), BASELINE configs[2] and configs[3] at the same size in the same context ("configs":
cfg3 = 2 x float64 AND + min / max / mean, cfg4_u64 / cfg4_bytes = per-pack bloom -> EQ -> count; each with kernel_ms,
GB/s, frac and its own parity verdict; f1_windows = SURVEY.md 8(f) row 1, time windows over an out-of-order clock: 2^28 rows ->
1024 windows as whole calls, with the path each took and the sorted path's time beside it), "placement_off" (the headline, cfg3 and cfg4_bytes kernels once more in a second
context with knox_ctx_set_placement(ctx, 0), and the points of peak the arena gains on each; "placement.probe_ms" is what
its probes cost), and the CPU baseline (scalar port @ 1 thread, AVX2 @ 1 thread, AVX2 @ all cores).

Environment (diagnostics, none needed): KNOX_PLACEMENT=0 plain hipMalloc; KNOX_BENCH_DEPTH scans queued ahead;
KNOX_BENCH_FORCE_DIST=1 / KNOX_BENCH_BACKEND=gloo rehearsal modes of the multi-rank path on a 1-GPU box.

Prints ONE JSON line on stdout (rank 0); progress notes go to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
BYTES_PER_ROW = 8.125  # 8 B value read + 1/8 B bitset written


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(sample_rows, pack_rows, a, b, gpu_pack_counts, min_seconds=10.0):
    """The oracle's literal restatement of the reference flow (cmp -> Indexes -> SumReducer per
    row), one thread like the reference's one-goroutine pack loop (pack/table/query.go:195-219),
    on the first `sample_rows` rows of the same workload, repeated until >= min_seconds of CPU
    work have been timed; beside it the AVX2 port of the compare kernel with the masked sum fused
    (the strategy of internal/cmp/avx2/int64_AVX2_amd64.s:1154-1330) on ONE thread -- the figure
    closest to the reference's published 1.8-2.8 G rows/s per core --, the AVX-512 form (k-mask compare,
    internal/cmp/avx512/int64_AVX512_amd64.s:655) on one thread where the host has it, and the best of them on every host core."""
    import numpy as np
    from oracle import oracle as O
    cores = host_cores()
    x = O.generate_mt(O.I64, 0, 2, 0, sample_rows, lo=-(1 << 40), span=1 << 41, threads=cores)
    passes, dt = 0, 0.0
    while dt < min_seconds and passes < 64:
        t0 = time.perf_counter()
        bits, cnt, s = O.scan_i64_bw_sum(x, pack_rows, a, b)
        dt += time.perf_counter() - t0
        passes += 1
    ok = None
    if gpu_pack_counts is not None:
        ok = int(gpu_pack_counts[: sample_rows // pack_rows].sum()) == int(cnt)
    rows = sample_rows * passes

    best = O.simd_best_level()

    def simd(threads, seconds, level):
        try:
            level = min(level, best)
            sbits = np.zeros((sample_rows + 7) >> 3, dtype=np.uint8)
            O.simd_scan_i64_bw_sum(x, pack_rows, a, b, threads, sbits, level)  # warm the pages
            sp, sdt = 0, 0.0
            while sdt < seconds and sp < 64:
                t0 = time.perf_counter()
                _, scnt, ssum = O.simd_scan_i64_bw_sum(x, pack_rows, a, b, threads, sbits, level)
                sdt += time.perf_counter() - t0
                sp += 1
            how = {3: "AVX-512 compare into a k-mask (8 lanes = one bitset byte), POPCNT per 64 rows, masked add (the strategy of internal/cmp/avx512/int64_AVX512_amd64.s:655)",
                   2: "AVX2 cmp + movemask + popcnt with the masked sum fused (internal/cmp/avx2/int64_AVX2_amd64.s:1154-1330)", 1: "scalar"}[level]
            return {"value": sample_rows * sp / sdt, "unit": "rows/s", "cores": threads, "isa": O.SIMD_LEVELS[level],
                    "gb_per_s": sample_rows * sp * BYTES_PER_ROW / sdt / 1e9,
                    "matches_port": bool(scnt == cnt and ssum == s and (sbits == bits).all()),
                    "note": "%s; oracle/knox_cpu_simd.c, pack-parallel over %d thread(s)" % (how, threads)}
        except Exception as e:  # the baseline must never break the bench line
            return {"error": str(e)}

    # AVX2 on one thread (rounds 1-2's figure), the best ISA of the host on one thread (AVX-512 where it exists), the best on every core
    return {"value": rows / dt, "unit": "rows/s", "cores": 1, "kind": "port", "simd_1_thread": simd(1, 3.0, 2),
            "simd_best_isa_1_thread": simd(1, 3.0, 3) if best >= 3 else None, "simd_all_cores": simd(cores, 3.0, 3),
            "sample": "first %d rows (%d packs of 2^20) of the same workload x %d passes, oracle/knox_oracle.c "
                      "ko_scan_i64_bw_sum (cmp_bw -> Indexes -> SumReducer per row), %.1f s timed, gcc -O2, 1 thread"
                      % (sample_rows, sample_rows // pack_rows, passes, dt),
            "gb_per_s": rows * BYTES_PER_ROW / dt / 1e9, "matches_gpu_counts": ok}


def in_run_parity(ctx, col_seed, n_rows, pack_rows, a, b, bits, res, row_seed_first=0):
    """SURVEY.md 8(d) "parity checks in-run", outside the timed region: (1) the bitset bytes of a 64-pack sample equal the
    oracle's cmp_bw output (memcmp), their counts and wrapping sum too; (2) the XOR of ALL 64-bit bitset words, the total
    count and the wrapping sum of the WHOLE run equal the CPU's over all rows (AVX2 port on every core -- itself checked
    against the literal port on the sample -- over rows regenerated on the host from the generator's spec)."""
    import numpy as np
    from oracle import oracle as O
    cores = host_cores()
    n_packs = n_rows // pack_rows
    got = bits.bytes()
    pb = pack_rows >> 3
    sample = sorted(set(int(p) for p in np.linspace(0, n_packs - 1, 64)))
    ok_bits = ok_counts = True
    ssum = 0
    for p in sample:
        x = O.generate(O.I64, 0, col_seed, p * pack_rows, pack_rows, lo=-(1 << 40), span=1 << 41)
        wb, wc = O.cmp(O.I64, O.RG, x, a, b)
        ok_bits = ok_bits and bool((got[p * pb:(p + 1) * pb] == wb).all())
        ok_counts = ok_counts and int(res.pack_counts[p]) == int(wc)
        ssum += int(O.reduce(O.I64, O.R_SUM, x, wb)[0])
    # the whole run, chunk by chunk (bounded host memory)
    chunk = min(n_rows, 1 << 27)
    xbuf = np.empty(chunk, dtype=np.int64)
    cbits = np.zeros(chunk >> 3, dtype=np.uint8)
    xor_cpu, cnt_cpu, sum_cpu = np.uint64(0), 0, 0
    for r0 in range(0, n_rows, chunk):
        O.generate_mt(O.I64, 0, col_seed, r0, chunk, lo=-(1 << 40), span=1 << 41, threads=cores, out=xbuf)
        _, c, s_ = O.simd_scan_i64_bw_sum(xbuf, pack_rows, a, b, cores, cbits, 0)   # the host's best ISA
        xor_cpu ^= np.bitwise_xor.reduce(cbits.view(np.uint64))
        cnt_cpu += int(c)
        sum_cpu = (sum_cpu + int(s_)) % (1 << 64)
    xor_gpu = np.bitwise_xor.reduce(got.view(np.uint64))
    gsum = int(res.states[0].isum) % (1 << 64)
    return {"sample_packs": len(sample), "sample_bitsets_equal_oracle": ok_bits, "sample_pack_counts_equal": ok_counts,
            "whole_run_rows_checked": n_rows, "xor_of_all_bitset_words": "%016x" % int(xor_gpu),
            "xor_equals_cpu": bool(xor_gpu == xor_cpu), "count_equals_cpu": bool(res.count == cnt_cpu),
            "sum_equals_cpu": bool(gsum == sum_cpu), "sample_sum_mod_2^64": ssum % (1 << 64)}


def _timed_steps(ctx, C, plan, bits, n_packs, steps, warmup, pack_skip=None):
    """`steps` software-pipelined passes (scan i+1 enqueued before scan i is collected, like the headline loop) after
    `warmup` untimed ones: wall clock per step + the dominant kernel's mean duration (the library's HIP events on its
    own stream, knox_prof_*).  Every pass produces and collects the bitset, the per-pack counts and the aggregates."""
    def run(k):
        ctx.scan_async(plan, bits_out=bits, pack_counts=True, pack_skip=pack_skip)
        last = None
        for i in range(k):
            if i + 1 < k:
                ctx.scan_async(plan, bits_out=bits, pack_counts=True, pack_skip=pack_skip)
            last = ctx.scan_collect(plan, n_packs)
        return last
    # the first ~10 launches after the device sat idle (generation, bloom build, host-side checks) run at a lower clock
    # (DESIGN.md 7 "clock ramp"): at least 12 untimed passes, whatever --warmup says
    run(max(12, warmup))
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.sync()
    t0 = time.perf_counter()
    res = run(steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    each = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, each, 512, C.byref(k))
    ctx.L.knox_prof_enable(ctx.h, 0)
    v = [each[i] for i in range(k.value)]
    return res, dt / steps * 1e3, (sum(v) / len(v) if v else 0.0), len(v), (min(v) if v else 0.0, max(v) if v else 0.0)


def _cfg_line(n_rows, bytes_per_row, step_ms, kernel_ms, launches, kernel, spread=None):
    gbs = n_rows * bytes_per_row / kernel_ms / 1e6 if kernel_ms > 0 else 0.0
    return {"rows": n_rows, "bytes_per_row": bytes_per_row, "algorithmic_bytes_per_launch": n_rows * bytes_per_row, "kernel": kernel,
            "kernel_ms": kernel_ms, "kernel_ms_min_max": list(spread) if spread else None, "launches_timed": launches, "ms_per_step": step_ms, "rows_per_s": n_rows / step_ms * 1e3,
            "achieved_gb_per_s": gbs, "frac": gbs / HBM_PEAK_GBS, "whole_step_gb_per_s": n_rows * bytes_per_row / step_ms / 1e6}


def secondary_configs(K, ctx, n_rows, pack_rows, steps, warmup, parity=True):
    """BASELINE configs[2] and configs[3] at the same size as the headline, after its timed region and in the SAME context
    (its arena places the new columns; nothing goes back to the driver in between -- the driver clearing freed memory
    slows every scan on the device for seconds): kernel time from HIP events, achieved GB/s against the
    8 TB/s peak with SURVEY.md 8(d)'s algorithmic bytes per row, and an in-run parity verdict (the oracle on sampled
    packs + whole-run invariants).  Reported under "configs" in the one JSON line; the headline fields do not depend on
    any of this."""
    import ctypes as C
    import numpy as np
    from knoxdb_amd import And, Leaf
    from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
    O = None
    if parity:
        from oracle import oracle as O   # the checker, outside every timed region
    n_packs = n_rows // pack_rows
    sample = sorted(set(int(p) for p in np.linspace(0, n_packs - 1, 16)))
    pb = pack_rows >> 3
    out = {}

    def note(msg):
        print("[configs] " + msg, file=sys.stderr, flush=True)

    # ---- configs[2]: two float64 columns, `A >= 256 AND B < 0.5`, min / max / mean(A) (+ sum, count) ---------------
    t0 = time.perf_counter()
    ctx.reserve(2 * (n_rows * 8 + (1 << 20)))   # both columns of the tree in ONE placement class (knox_ctx_reserve), the bitset in another
    A = ctx.generate(BlockType.Float64, 1, 3, n_rows, pack_rows=pack_rows)   # k * 2^-10 grid: every summation order is exact
    B = ctx.generate(BlockType.Float64, 2, 4, n_rows, pack_rows=pack_rows)   # uniform [0, 1)
    bits = ctx.bits(n_rows, for_col=[A, B])   # verified against both read streams (knox_bits_alloc_for_cols)
    tree = And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5))
    plan = ctx.plan(tree, [A, B], aggs=[(R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.Sum, 0), (R.Count, 0)])
    res, step_ms, kms, kn, spread = _timed_steps(ctx, C, plan, bits, n_packs, steps, warmup)
    line = _cfg_line(n_rows, 16.125, step_ms, kms, kn, "knox::scan_kernel<8,2,FSUM|MINMAX|FIRSTLAST,float> (2 leaves, truth table, min/max/mean)", spread)
    vmin, vmax, vmean, vsum = (float(res.value(i)[0]) for i in range(4))
    line.update({"workload": "BASELINE configs[2]: 2^%d float64 rows x 2 columns, A >= 256 AND B < 0.5, min/max/mean(A) + sum + count, bitset + per-pack counts"
                             % (n_rows.bit_length() - 1), "selected_rows": int(res.count), "min": vmin, "max": vmax, "mean": vmean, "sum": vsum,
                 "bitset_pair_check_at_allocation": bits.pair_check[0]})
    if parity:
        got = bits.bytes()
        ok_bits = ok_cnt = True
        for p in sample:
            xa = O.generate(O.F64, 1, 3, p * pack_rows, pack_rows)
            xb = O.generate(O.F64, 2, 4, p * pack_rows, pack_rows)
            wa, _ = O.cmp(O.F64, O.GE, xa, 256.0, 0.0)
            wb, _ = O.cmp(O.F64, O.LT, xb, 0.5, 0.0)
            want = O.bits_op("and", wa, wb, pack_rows)
            ok_bits = ok_bits and bool((got[p * pb:(p + 1) * pb] == want).all())
            ok_cnt = ok_cnt and int(res.pack_counts[p]) == int(O.popcount(want, pack_rows))
        line["parity_in_run"] = {
            "sample_packs": len(sample), "sample_bitsets_equal_oracle": ok_bits, "sample_pack_counts_equal": ok_cnt,
            "pack_counts_sum_to_count": bool(int(res.pack_counts.sum()) == int(res.count)),
            "bitset_popcount_equals_count": bool(bits.count() == int(res.count)),
            # on the grid every partial sum is exact: the sum is THE sum, and the mean must be its one IEEE division
            "mean_equals_sum_over_count": bool(res.count > 0 and vmean == vsum / int(res.count)),
            "min_max_are_the_grid_ends": bool(vmin == 256.0 and vmax == ((1 << 20) - 1) / 1024.0) if n_rows >= (1 << 26) else None}
    for o in (A, B, bits):
        o.free()
    line["setup_and_run_s"] = time.perf_counter() - t0
    out["cfg3"] = line
    note("cfg3: kernel %.4f ms, %.0f GB/s, frac %.3f" % (line["kernel_ms"], line["achieved_gb_per_s"], line["frac"]))

    # ---- configs[3]: uint64 id + 32-byte []byte addr (one StringPool per pack), per-pack bloom -> EQ -> count ---------
    def bloom_then_eq(ctx, col, key, hash_of_key, bytes_per_row, kernel, oracle_pack_bits):
        bits = ctx.bits(n_rows, for_col=col)
        tb = time.perf_counter()
        bloom = ctx.bloom_build(col, 2)      # FilterTypeBloom2b: factor 2, built on the device
        ctx.sync()
        t_build = time.perf_counter() - tb
        maybe = bloom.probe(hash_of_key)
        plan = ctx.plan(And(Leaf(0, FilterMode.Equal, key)), [col], aggs=[(R.Count, 0)] if col.dtype != BlockType.Bytes else [])
        full, step_ms, kms, kn, spread = _timed_steps(ctx, C, plan, bits, n_packs, steps, warmup)
        line = _cfg_line(n_rows, bytes_per_row, step_ms, kms, kn, kernel, spread)
        got = bits.bytes() if parity else None
        skip = (1 - maybe).astype(np.uint8)
        pruned, pstep_ms, pkms, pkn, _ = _timed_steps(ctx, C, plan, bits, n_packs, steps, warmup, pack_skip=skip)
        scanned = int(maybe.sum())
        line.update({"selected_rows": int(full.count), "bloom_build_s": t_build, "bloom_filter_bytes": int(ctx.L.knox_bloom_bytes(bloom.h)),
                     "bitset_pair_check_at_allocation": bits.pair_check[0],
                     # packs the filter rejects cost 0 B/row: the roofline fraction above is over SCANNED packs (the full scan);
                     # this is the separately labelled figure with pruning
                     "bloom_pruned": {"packs_scanned": scanned, "packs": n_packs, "rows_scanned": int(pruned.rows_scanned), "kernel_ms": pkms,
                                      "ms_per_step": pstep_ms, "effective_rows_per_s_incl_pruned_packs": n_rows / pstep_ms * 1e3,
                                      "gb_per_s_over_scanned_packs": scanned * pack_rows * bytes_per_row / pkms / 1e6 if scanned and pkms > 0 else None,
                                      "selected_rows": int(pruned.count)}})
        if parity:
            ok_bits = ok_cnt = True
            for p in sorted(set(sample + [int(x) for x in np.flatnonzero(maybe)[:4]])):
                want, wc = oracle_pack_bits(p)
                ok_bits = ok_bits and bool((got[p * pb:(p + 1) * pb] == want).all())
                ok_cnt = ok_cnt and int(full.pack_counts[p]) == int(wc)
            line["parity_in_run"] = {
                "sample_packs": len(sample), "sample_bitsets_equal_oracle": ok_bits, "sample_pack_counts_equal": ok_cnt,
                "pack_counts_sum_to_count": bool(int(full.pack_counts.sum()) == int(full.count)),
                "pruned_equals_full": bool(pruned.count == full.count and (pruned.pack_counts == full.pack_counts).all()),
                "every_pruned_pack_is_empty_in_the_full_scan": bool((full.pack_counts[skip != 0] == 0).all()),
                "present_key_found": bool(full.count >= 1)}
        bloom.free()
        bits.free()
        return line

    t0 = time.perf_counter()
    ids = ctx.generate(BlockType.Uint64, 0, 5, n_rows, lo=0, span=0, pack_rows=pack_rows)   # unique-in-practice 64-bit ids
    present = int(ids.download(min(777, n_packs - 1) * pack_rows + 12345 % pack_rows, 1)[0])

    def want_u64(p):
        x = O.generate(O.U64, 0, 5, p * pack_rows, pack_rows, lo=0, span=0)
        return O.cmp(O.U64, O.EQ, x, present)
    line = bloom_then_eq(ctx, ids, present, K.hash_value(BlockType.Uint64, np.uint64(present)), 8.125,
                         "knox::scan_kernel<8,1,0> (uint64 EQ, count)", want_u64)
    line["workload"] = "BASELINE configs[3], uint64 leg: 2^%d ids, per-pack bloom (factor 2) probe, EQ present key, count" % (n_rows.bit_length() - 1)
    ids.free()
    line["setup_and_run_s"] = time.perf_counter() - t0
    out["cfg4_u64"] = line
    note("cfg4_u64: kernel %.4f ms, %.0f GB/s, frac %.3f" % (line["kernel_ms"], line["achieved_gb_per_s"], line["frac"]))

    t0 = time.perf_counter()
    key_len, wbits, ppw = 32, 16, 4
    addr = ctx.generate_bytes(6, n_rows, pack_rows, key_len, wbits, ppw)   # 2^24 keys, every key inside one window of 4 packs
    ptr, buf = ctx.download_bytes_pack(addr, min(777, n_packs - 1))
    r = 12345 % pack_rows
    present = bytes(buf[r * key_len:(r + 1) * key_len])

    def want_bytes(p):
        wp, wbuf, _ = O.generate_bytes_pack(6, p, pack_rows, pack_rows, key_len, wbits, ppw)
        return O.bytes_match(O.EQ, wp, wbuf, present)
    line = bloom_then_eq(ctx, addr, present, K.hash_bytes(present), 8 + key_len + 0.125,
                         "knox::bytes_eq_kernel (8 B pointer + 32 B payload per row)", want_bytes)
    line["workload"] = ("BASELINE configs[3], []byte leg: 2^%d rows of 32-byte addr values from 2^24 keys, one StringPool per pack, per-pack bloom "
                        "(factor 2, XXH3-64 on the device) probe, EQ present key, count") % (n_rows.bit_length() - 1)
    line["pool_bytes"] = int(ctx.L.knox_col_pool_bytes(addr.h))
    addr.free()
    line["setup_and_run_s"] = time.perf_counter() - t0
    out["cfg4_bytes"] = line
    note("cfg4_bytes: kernel %.4f ms, %.0f GB/s, frac %.3f" % (line["kernel_ms"], line["achieved_gb_per_s"], line["frac"]))
    # ---- time windows over an out-of-order clock (NativeBucket.Push's find / insert) ---------------------
    # 2^28 rows -> 1024 windows behind a 25 % selection, whole calls (wall clock around the C call, order of the clock known):
    # Value() per window without a sort, whole states, float64 sum / std through the exact adders
    try:
        t0 = time.perf_counter()
        nw = min(n_rows, 1 << 28)
        clock = ctx.generate(BlockType.Int64, 0, 9, nw, lo=1000, span=1 << 20, pack_rows=pack_rows)
        ival = ctx.generate(BlockType.Int64, 0, 2, nw, lo=-(1 << 40), span=1 << 41, pack_rows=pack_rows)
        fval = ctx.generate(BlockType.Float64, 2, 5, nw, pack_rows=pack_rows)
        wbits = ctx.bits(nw, for_col=ival)
        ctx.scan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [ival], bits_out=wbits)
        nb, wstep = 1024, (1 << 20) // 1024

        def wall(fn, reps=5):
            fn(); ctx.sync()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.sync()
            return (time.perf_counter() - t1) / reps * 1e3

        f1 = {"workload": "2^%d rows, shuffled clock -> %d windows, 25 %% selected, whole calls" % (nw.bit_length() - 1, nb), "rows": nw}
        for name, colv, fn in (("int64_sum", ival, R.Sum), ("int64_std", ival, R.Std), ("float64_sum", fval, R.Sum), ("float64_std", fval, R.Std)):
            ms = wall(lambda: ctx.reduce_buckets_values(clock, colv, fn, 1000, wstep, nb, mask=wbits))
            f1[name] = {"ms": ms, "rows_per_s": nw / ms * 1e3, "path": ctx.windows_last_path()}
        ms = wall(lambda: ctx.reduce_buckets_view(clock, ival, 1000, wstep, nb, mask=wbits))
        f1["int64_states"] = {"ms": ms, "rows_per_s": nw / ms * 1e3, "path": ctx.windows_last_path()}
        # parity in the run: the sort-free fold against the sorted path, byte for byte
        a, b = ctx.reduce_buckets_values(clock, ival, R.Sum, 1000, wstep, nb, mask=wbits)
        direct = (a.tobytes(), b.tobytes())
        os.environ["KNOX_WINDOW_DIRECT"] = "0"
        try:
            a, b = ctx.reduce_buckets_values(clock, ival, R.Sum, 1000, wstep, nb, mask=wbits)
            f1["sorted_int64_sum_ms"] = wall(lambda: ctx.reduce_buckets_values(clock, ival, R.Sum, 1000, wstep, nb, mask=wbits))
        finally:
            os.environ.pop("KNOX_WINDOW_DIRECT", None)
        f1["parity"] = "green" if direct == (a.tobytes(), b.tobytes()) else "RED"
        f1["setup_and_run_s"] = time.perf_counter() - t0
        for o in (clock, ival, fval, wbits):
            o.free()
        out["f1_windows"] = f1
        note("f1_windows: int64 sum %.3f ms (sorted %.3f), states %.3f ms, float64 sum %.3f ms, std %.3f ms, parity %s" % (
            f1["int64_sum"]["ms"], f1["sorted_int64_sum_ms"], f1["int64_states"]["ms"], f1["float64_sum"]["ms"], f1["float64_std"]["ms"], f1["parity"]))
    except Exception as e:   # never at the cost of the configs above
        out["f1_windows"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def placement_off_ab(K, device, n_rows, pack_rows, on):
    """What the arena's placement buys TODAY, in the driver's own record: the headline, configs[2] and the configs[3] []byte
    leg once more in a SECOND context of this process with placement switched off (knox_ctx_set_placement(ctx, 0): plain
    hipMalloc / hipFree, no probes, bitsets wherever the driver puts them) -- same generators, same kernels, 12 ramp launches
    and 10 timed ones each (the library's HIP events), outside every timed region, while the first context (idle, nothing
    freed: a driver clearing memory would slow both) is still open.  `on` = the kernel_ms the run measured with the arena."""
    import ctypes as C
    from knoxdb_amd import And, Leaf
    from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
    n_packs = n_rows // pack_rows
    out = {}
    ctx = K.Context(device)
    try:
        ctx.set_placement(False)

        def kms(plan, bits):
            for _ in range(12):
                ctx.scan_async(plan, bits_out=bits, pack_counts=True)
                ctx.scan_collect(plan, n_packs)
            ctx.L.knox_prof_enable(ctx.h, 1)
            for _ in range(10):
                ctx.scan_async(plan, bits_out=bits, pack_counts=True)
                ctx.scan_collect(plan, n_packs)
            ms, k = C.c_double(), C.c_int()
            ctx.L.knox_prof_read(ctx.h, C.byref(ms), C.byref(k))
            ctx.L.knox_prof_enable(ctx.h, 0)
            return ms.value / max(1, k.value)

        def entry(name, ms_off, bytes_per_row):
            f_off = n_rows * bytes_per_row / ms_off / 1e6 / HBM_PEAK_GBS
            e = {"kernel_ms": ms_off, "frac": f_off}
            if on.get(name):
                f_on = n_rows * bytes_per_row / on[name] / 1e6 / HBM_PEAK_GBS
                e.update({"kernel_ms_with_arena": on[name], "frac_with_arena": f_on, "gain_points": (f_on - f_off) * 100.0})
            out[name] = e

        col = ctx.generate(BlockType.Int64, 0, 2, n_rows, lo=-(1 << 40), span=1 << 41, pack_rows=pack_rows)
        bits = ctx.bits(n_rows)
        entry("headline", kms(ctx.plan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [col], aggs=[(R.Sum, 0), (R.Count, 0)]), bits), BYTES_PER_ROW)
        col.free()
        A = ctx.generate(BlockType.Float64, 1, 3, n_rows, pack_rows=pack_rows)
        B = ctx.generate(BlockType.Float64, 2, 4, n_rows, pack_rows=pack_rows)
        entry("cfg3", kms(ctx.plan(And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5)), [A, B],
                                   aggs=[(R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.Sum, 0), (R.Count, 0)]), bits), 16.125)
        A.free(); B.free()
        addr = ctx.generate_bytes(6, n_rows, pack_rows, 32, 16, 4)
        _, buf = ctx.download_bytes_pack(addr, min(777, n_packs - 1))
        key = bytes(buf[:32])
        entry("cfg4_bytes", kms(ctx.plan(And(Leaf(0, FilterMode.Equal, key)), [addr], aggs=[]), bits), 40.125)
        addr.free(); bits.free()
        out["note"] = ("second context, knox_ctx_set_placement(ctx, 0): plain allocations, no probes; 12 ramp + 10 timed launches per config; "
                       "gain_points = (frac with the arena, from this run's own lines) - (frac without), in points of the 8 TB/s peak")
    finally:
        ctx.close()
    return out


def measure_traffic_live(rows_log2, timeout_s=90):
    """roofline.traffic measured in THIS run: two child processes under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate
    passes, as MI355X_MICROARCH.md prescribes; no trace flags) run the headline scan a few times (`bench.py --pmc-child`), the
    per-dispatch counters of scan_kernel<8,1,ISUM> are read back from the CSVs.  Called after this process has closed its
    context; the children are ordinary child processes (nothing is exec'ed over a process that touched the GPU).  Returns
    (hbm_bytes_per_launch, detail) or (None, reason): the caller then falls back to profiles/traffic.json."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="knox_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", out_dir, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--rows-log2", str(rows_log2), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-configs"]
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "KNOX_BENCH_FORCE_DIST", "KNOX_BENCH_BACKEND", "KNOX_ARENA_LOG")}
                env["TMPDIR"] = "/tmp"   # the child is a plain single-process run whatever launched this one
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 pass %s timed out" % ctr
            if r.returncode != 0:
                return None, "rocprofv3 pass %s failed (rc %d): %s" % (ctr, r.returncode, r.stderr.decode(errors="replace")[-200:])
            got = []
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "scan_kernel<8, 1, 1u, false, false, 0u>" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None, "no %s rows for the headline kernel" % ctr
            vals[ctr] = (sum(got) / len(got), len(got))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a 16 B/lane streaming read (the guide's
    # gfx950 correction): doubled.  WRITE_SIZE is exact for 16-B-per-lane streaming stores.
    fetch_b, write_b = vals["FETCH_SIZE"][0] * 1024 * 2, vals["WRITE_SIZE"][0] * 1024
    return fetch_b + write_b, {"FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"][0], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"][0],
                               "dispatches": [vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]], "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b}


def launch_ranks(n):
    """`bench.py --gpus N` without a launcher around it: start N rank processes (one per GPU, RCCL over xGMI between
    them) and wait.  They are FRESH interpreters -- this parent has not touched the GPU and never does (a process
    that has initialised HIP must not fork or exec its ranks) -- given RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* the
    way torch.distributed.run would; rank 0 inherits stdout and prints the one JSON line."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and failed_at is None:
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 30:   # a rank died: the others wait in a barrier for ever
            for p in procs:
                if p.poll() is None:
                    p.kill()                                          # exactly the processes started above
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows-log2", type=int, default=30, help="rows per GPU = 2^k (default 2^30 = the 1B-row config)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and every oracle-backed parity check")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip BASELINE configs[2] / configs[3] after the timed region")
    ap.add_argument("--no-placement-ab", action="store_true", help="skip the placement-off repetition of the three big configs (diagnostics, outside the timed region)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from profiles/traffic.json instead of two rocprofv3 --pmc child passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the child of measure_traffic_live: headline scans only, no output
    ap.add_argument("--cpu-sample-log2", type=int, default=29)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--collective", choices=["abi", "torch"], default=None,
                    help="who runs the per-step exchange at N > 1: the library's own RCCL calls (knox_comm_*, what a Go host uses; default "
                         "for the nccl backend) or torch.distributed (default for the gloo rehearsal)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # exactly ONE line goes to stdout: libraries that print banners there (RCCL's version header) are sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # KNOX_BENCH_BACKEND=gloo is a rehearsal mode only (several ranks sharing one GPU on a 1-GPU box; on a box without
    # a GPU it still brings the process group up before the device check below ends the run)
    backend = os.environ.get("KNOX_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()   # counting devices does not initialise the GPU
    if backend == "nccl" and n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    device_index = local_rank % max(1, n_dev)
    # KNOX_BENCH_FORCE_DIST=1: run the multi-rank code path (process group, barriers, the RCCL collectives) even
    # with one rank -- the rehearsal a 1-GPU box allows for the nccl backend
    if world > 1 or os.environ.get("KNOX_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")   # only the one-rank rehearsal comes without a launcher
        if backend == "nccl":
            torch.cuda.set_device(device_index)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        print("[rank %d] process group up: backend %s, %d rank(s)" % (rank, backend, dist.get_world_size()), file=sys.stderr, flush=True)
        collective = args.collective or ("abi" if backend == "nccl" else "torch")
        if collective == "abi":
            # the library's own communicator: rank 0 makes the ncclUniqueId, the process group only carries its 128 bytes
            from knoxdb_amd import dist as kdist0
            box = [None]
            if rank == 0:
                try:
                    box[0] = kdist0.comm_unique_id()
                except RuntimeError as e:
                    if backend == "nccl":
                        raise
                    box[0] = bytes(128)   # gloo rehearsal on a box without a GPU: the exchange itself is what is rehearsed
                    print("[rank 0] abi collective: %s -- rehearsal id" % e, file=sys.stderr, flush=True)
            dist.broadcast_object_list(box, src=0)
            comm_id = box[0]
            print("[rank %d] abi collective: unique id exchanged (%d bytes)" % (rank, len(comm_id)), file=sys.stderr, flush=True)
        if args.gpus > 1 and dist.get_world_size() != args.gpus:
            print("[rank %d] note: --gpus %d but the launcher started %d rank(s)" % (rank, args.gpus, dist.get_world_size()), file=sys.stderr)
    if not torch.cuda.is_available():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(device_index)
    local_rank = device_index

    import numpy as np
    # the arena's decisions (chunk labels, every write-buffer candidate and what it measured) of THIS run, kept beside the
    # other outputs: a placement that went wrong on some box can be read afterwards (knox_arena.h dbg)
    if "KNOX_ARENA_LOG" not in os.environ and not args.pmc_child:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            os.environ["KNOX_ARENA_LOG"] = os.path.join(ROOT, "gpurun_out", "arena_rank%d.log" % rank)
            open(os.environ["KNOX_ARENA_LOG"], "w").close()
        except OSError:
            os.environ.pop("KNOX_ARENA_LOG", None)
    import knoxdb_amd as K
    from knoxdb_amd import And, Leaf
    from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
    from knoxdb_amd import dist as kdist

    n_rows = 1 << args.rows_log2
    pack_rows = 1 << 20 if n_rows >= (1 << 20) else n_rows
    n_packs = n_rows // pack_rows
    a, b = -(1 << 38), 1 << 38
    seed = 2 if world == 1 else 20 + rank  # cfg 2 seed 2, cfg 5 seeds 20..27
    ctx = K.Context(local_rank)
    if args.blocks_per_cu:
        ctx.set_tuning(args.blocks_per_cu)
    # rank r owns global packs [r*n_packs, (r+1)*n_packs): row ids are offset by row_base
    first_pack, _ = kdist.shard_range(n_packs * world, rank, world)
    row_base = first_pack * pack_rows
    import ctypes as C

    arena_gib = float(os.environ.get("KNOX_BENCH_ARENA_GIB", "160"))   # headline 8 GiB; configs[2] 16 GiB; configs[3] 8 + 42 GiB + 2 x 2 GiB of filters
    ctx.set_arena(int(arena_gib * 2**30))   # explicit budget of the context's device-memory arena (include/knoxgpu.h)

    def kernel_ms(p, out_bits, reps):
        """mean duration of the scan kernel over `reps` launches (the library's own HIP events on its stream)"""
        ctx.L.knox_prof_enable(ctx.h, 1)
        for _ in range(reps):
            ctx.scan_async(p, bits_out=out_bits, pack_counts=out_bits is not None, row_base=row_base)
            ctx.scan_collect(p, n_packs if out_bits is not None else 0)
        ms, k = C.c_double(), C.c_int()
        ctx.L.knox_prof_read(ctx.h, C.byref(ms), C.byref(k))
        ctx.L.knox_prof_enable(ctx.h, 0)
        return ms.value / max(1, k.value)

    tree = And(Leaf(0, FilterMode.Range, a, b))
    t_setup = time.perf_counter()
    col = ctx.generate(BlockType.Int64, 0, seed, n_rows, first_row=0, lo=-(1 << 40), span=1 << 41, pack_rows=pack_rows)
    bits = ctx.bits(n_rows, for_col=col)   # the selection bitset: a write buffer verified against this very column
    plan = ctx.plan(tree, [col], aggs=[(ReducerFunc.Sum, 0), (ReducerFunc.Count, 0)])
    ctx.sync()
    # the ONE setup check, with the real kernel, outside the timed region: the scan against its own read-only form.  A low
    # ratio right after start is almost always a device still busy clearing somebody else's freed memory: look once more
    # after 2.5 s and report both; nothing else is tried, nothing is re-allocated.
    checks = []
    quiet = []
    quiet_wait_s, quiet_ended_by = 0.0, None
    if n_rows >= (1 << 26):
        ro_plan0 = ctx.plan(tree, [col], aggs=[])
        # A process that has just exited (the GPU tests, the smoke run, a previous bench) leaves the driver clearing the
        # memory it freed, which costs EVERY kernel on the device ~5 % for as long as it lasts (measured: three bench runs back
        # to back, the second and third read 6.82 instead of 7.16 TB/s for their whole 5 s).  The read-only form
        # of the scan is sampled every 0.5 s until two consecutive samples agree within 0.4 %: setup, outside the
        # timed region, reported as placement.device_quiet_ms.
        kernel_ms(ro_plan0, None, 12)                     # clock ramp
        quiet.append(kernel_ms(ro_plan0, None, 6))
        # The wait ends on STABILITY alone (two consecutive samples within 0.4 %), never on a target rate -- a benchmark that
        # waits until the device is as fast as it would like to see it has chosen its state (round-3 review) -- or after
        # 10 s; the line reports the seconds waited, the samples, and which of the two ended the wait.
        t_wait = time.perf_counter()
        quiet_ended_by = "timeout"
        while True:
            time.sleep(0.5)
            kernel_ms(ro_plan0, None, 6)                  # clock ramp after the pause
            quiet.append(kernel_ms(ro_plan0, None, 6))
            waited = time.perf_counter() - t_wait
            if abs(quiet[-1] - quiet[-2]) <= 0.004 * quiet[-2]:   # launch-to-launch noise is ~0.2 %
                quiet_ended_by = "stable"
                break
            if waited > 10.0:
                break
        quiet_wait_s = time.perf_counter() - t_wait
        for attempt in range(2):
            kernel_ms(plan, bits, 4)                      # clock ramp
            ro_ms, mix_ms = kernel_ms(ro_plan0, None, 6), kernel_ms(plan, bits, 8)
            checks.append({"read_only_ms": ro_ms, "scan_ms": mix_ms, "scan_over_read_only": ro_ms / mix_ms})
            print("[setup] check %d: read-only %.4f ms, scan %.4f ms, ratio %.3f, bitset verified at allocation: %s" %
                  (attempt, ro_ms, mix_ms, ro_ms / mix_ms, bits.pair_check), file=sys.stderr)
            if ro_ms / mix_ms >= 0.925:
                break
            time.sleep(2.5)
    t_setup = time.perf_counter() - t_setup

    # the final aggregate is the ONLY thing that crosses ranks: one all-reduce(int64, sum) of {sum, count} per step
    # (backend nccl = ncclAllReduce over xGMI); the all-gather of the full 144-byte states (what min / max / first /
    # last / float sums need) runs in the warmup and must give the same answer
    tdev = torch.device("cuda", local_rank) if backend == "nccl" else None
    gather = reduce_sc = None
    ranks_in_comm = None
    if dist is not None and collective == "abi":
        # knox_comm_init = ncclCommInitRank on this rank's device; every step's exchange is then
        # knox_comm_allreduce_sum_count_start / _finish (ONE ncclAllReduce(int64 x 3) on the context's collective stream)
        up, why = 1, ""
        try:
            ranks_in_comm = kdist.comm_init(ctx, dist.get_world_size(), rank, comm_id)
        except Exception as e:   # no RCCL library to dlopen, ncclCommInitRank refused ...: every rank must take the same way out
            up, why = 0, str(e)
        flag = torch.tensor([up], dtype=torch.int32, device=tdev if tdev is not None else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            gather, reduce_sc = kdist.AbiStateGather(ctx, 2), kdist.AbiSumCountReduce(ctx)
            print("[rank %d] abi collective: communicator up, ncclCommCount = %d" % (rank, ranks_in_comm), file=sys.stderr, flush=True)
        else:
            print("[rank %d] abi collective unavailable on some rank (%s): torch.distributed carries the exchange" % (rank, why or "another rank failed"), file=sys.stderr, flush=True)
            collective, ranks_in_comm = "torch", None
    if dist is not None and collective != "abi":
        gather, reduce_sc = kdist.StateGather(2, tdev), kdist.SumCountReduce(tdev)

    def enqueue():
        ctx.scan_async(plan, bits_out=bits, pack_counts=True, row_base=row_base)

    pending = []  # the previous step's all-reduce, still in flight behind this step's scan

    def collect():
        res = ctx.scan_collect(plan, n_packs)
        if dist is not None:
            # this step's exchange is enqueued, the previous step's is read: the collective never stalls the host
            # between two scans (every step's whole-table answer is still produced inside the timed region -- the
            # last one is read by run() before it returns)
            pending.append(reduce_sc.start(int(res.states[0].isum), res.count))
            if len(pending) > 1:
                gsum, total = reduce_sc.finish(pending.pop(0))
                return res, gsum, total
            return res, None, None
        return res, int(res.states[0].isum), res.count

    # scans queued ahead of the one being collected.  With ranks > 1 every collect() ends in the RCCL all-gather of
    # the states, whose kernel can only start when the running scan's workgroups retire: a second queued scan keeps
    # the device busy over that wait (the library's result ring holds 4 scans in flight).
    depth = int(os.environ.get("KNOX_BENCH_DEPTH", "2" if dist is not None else "1"))

    def run(k):
        """k steps, software-pipelined: up to `depth` later steps are enqueued before step i's results are
        collected (every step's bitset, pack counts and aggregates are produced and collected; the
        device just never idles between steps, as under a host with several queries in flight)."""
        queued = 0
        for _ in range(min(depth, k)):
            enqueue()
            queued += 1
        last = None
        for _ in range(k):
            if queued < k:
                enqueue()
                queued += 1
            last = collect()
        if dist is not None:
            last = (last[0],) + tuple(reduce_sc.finish(pending.pop(0)))
            assert not pending
        return last

    wres, wsum, wtotal = run(max(1, args.warmup))
    if args.pmc_child:   # under rocprofv3 --pmc: a few more scans for the counters, then out (no JSON line, no baselines)
        run(args.steps)
        ctx.sync()
        col.free()
        bits.free()
        ctx.close()
        return
    collectives_agree = None
    if dist is not None:
        merged, gtotal = gather(wres.states, wres.count)
        collectives_agree = bool(gtotal == wtotal and int(merged[0].isum) == wsum and int(merged[1].count) == wtotal)
        assert collectives_agree, "all-reduce {sum,count} != all-gather + combine"
    if dist is not None:
        dist.barrier()
        # The barrier (and the all-gather check before it) leaves the device idle for milliseconds, and the first ~10 scans
        # after such a pause run up to 15 % slower (per-launch times of a 1-rank rehearsal: 1.26 1.33 1.41 1.44 1.40 1.38 1.33
        # 1.31 1.29 1.27 ... 1.245 ms; DESIGN.md 7 "clock ramp") -- with one rank and no barrier the W warm-up steps absorb
        # that.  So the untimed passes that bring the clocks back run AFTER the barrier; every rank does the same 12, the
        # synchronize below drains them, and the timed region is still K steps between barrier + synchronize on both sides.
        run(12)
    ctx.L.knox_prof_enable(ctx.h, 1)
    torch.cuda.synchronize()
    ctx.sync()
    t0 = time.perf_counter()
    res, gsum, total = run(args.steps)
    ctx.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    each, kn = (C.c_float * 512)(), C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, each, 512, C.byref(kn))   # every timed launch of the scan kernel, in order
    ctx.L.knox_prof_enable(ctx.h, 0)
    per_launch_ms = [float(each[i]) for i in range(kn.value)]
    kms = C.c_double(sum(per_launch_ms))
    if os.environ.get("KNOX_BENCH_VERBOSE") == "1":
        print("[rank %d] scan kernel per launch (ms): %s" % (rank, " ".join("%.4f" % x for x in per_launch_ms)), file=sys.stderr, flush=True)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # empirical ceiling: the same column through the read-only form of the kernel (count only:
    # no bitset, no pack counts, no sum -- scan_kernel<8,1,AGG 0>, its own symbol in profiles), outside the timed region
    ro_plan = ctx.plan(And(Leaf(0, FilterMode.Range, a, b)), [col], aggs=[])
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(ro_plan)
    for _ in range(9):
        ctx.scan_async(ro_plan)
        ctx.scan_collect(ro_plan, 0)
    ro_res = ctx.scan_collect(ro_plan, 0)
    ro_ms, ro_n = C.c_double(), C.c_int()
    ctx.L.knox_prof_read(ctx.h, C.byref(ro_ms), C.byref(ro_n))
    ctx.L.knox_prof_enable(ctx.h, 0)
    read_only_gbs = n_rows * 8.0 * ro_n.value / (ro_ms.value * 1e6) if ro_ms.value > 0 else 0.0
    assert ro_res.count == res.count

    scan_ms = kms.value / max(1, kn.value)
    scan_ms_ranks = [scan_ms, scan_ms]
    if dist is not None:
        t = torch.tensor([scan_ms, -scan_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        scan_ms_ranks = [-float(t[1].item()), float(t[0].item())]   # [fastest rank, slowest rank]
    achieved = n_rows * BYTES_PER_ROW / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    total_rows = n_rows * world * args.steps
    value = total_rows / elapsed
    sum_val = gsum - (1 << 64) if gsum >= (1 << 63) else gsum   # Value() of the int64 SumReducer: the sum mod 2^64 as int64

    # 0 % and 100 % selectivity runs of the same scan, outside the timed region: same bytes moved, the
    # aggregate work differs (no row / every row feeds the sum)
    sel_runs = {}
    if rank == 0 and n_rows >= (1 << 26):
        for name, lo_, hi_ in (("0%", 1 << 50, (1 << 50) + 5), ("100%", -(1 << 41), 1 << 41)):
            p_ = ctx.plan(And(Leaf(0, FilterMode.Range, lo_, hi_)), [col], aggs=[(ReducerFunc.Sum, 0), (ReducerFunc.Count, 0)])
            kernel_ms(p_, bits, 4)
            ms_ = kernel_ms(p_, bits, 10)
            ctx.scan_async(p_, bits_out=bits, pack_counts=True, row_base=row_base)
            r_ = ctx.scan_collect(p_, n_packs)
            sel_runs[name] = {"kernel_ms": ms_, "gb_per_s": n_rows * BYTES_PER_ROW / ms_ / 1e6, "frac": n_rows * BYTES_PER_ROW / ms_ / 1e6 / HBM_PEAK_GBS,
                              "selected_rows": int(r_.count)}
        # leave the headline result in the bitset for the parity check below
        ctx.scan_async(plan, bits_out=bits, pack_counts=True, row_base=row_base)
        res = ctx.scan_collect(plan, n_packs)
    parity = None
    if rank == 0 and not args.no_cpu_baseline and os.environ.get("KNOX_BENCH_SKIP_HEADLINE_PARITY") != "1":
        parity = in_run_parity(ctx, seed, n_rows, pack_rows, a, b, bits, res)
        assert parity["sample_bitsets_equal_oracle"] and parity["xor_equals_cpu"] and parity["count_equals_cpu"] and parity["sum_equals_cpu"], parity

    ai = ctx.arena_info()
    class_gib, classes_seen = ctx.arena_classes()
    out = {
        "metric": "scanned rows/sec + achieved HBM GB/s, 1B-row int64 range filter+sum",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: %s x 2^%d int64 rows, pack size 2^20, Between[-2^38,2^38] (25%% sel.) "
                               "+ sum + count, bitset + per-pack counts emitted" % (1 if world == 1 else 4, "%d GPU" % world, args.rows_log2),
                   "rows_per_gpu": n_rows, "pack_rows": pack_rows, "parallelism": "pack-sharded x%d" % world,
                   "ranks_in_process_group": dist.get_world_size() if dist is not None else 1,
                   "ranks_in_rccl_communicator": ranks_in_comm,   # ncclCommCount of the library's own communicator (--collective abi)
                   "collective": ("%s every step; all-gather of the states + knox_agg_combine agrees: %s" %
                                  ("knox_comm_allreduce_sum_count_start/_finish = ncclAllReduce(int64 x3, sum) on the library's communicator" if collective == "abi"
                                   else "torch.distributed all_reduce(int64 x3, sum) over %s" % backend, collectives_agree)) if dist is not None else None,
                   "selected_rows": int(total), "sum_mod_2^64": sum_val},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "knox::scan_kernel<8,1,ISUM> (int64 Between + sum)", "kernel_ms": scan_ms,
                     "kernel_ms_min_max_over_ranks": scan_ms_ranks,
                     "kernel_ms_min_max_over_launches": [min(per_launch_ms), max(per_launch_ms)] if per_launch_ms else None,
                     "launches_timed": kn.value, "algorithmic_bytes_per_launch": n_rows * BYTES_PER_ROW,
                     "read_only_ceiling_gb_per_s": read_only_gbs,
                     "frac_of_read_only_ceiling": achieved / read_only_gbs if read_only_gbs > 0 else None},
        "whole_step_gb_per_s": n_rows * BYTES_PER_ROW / (elapsed / args.steps) / 1e9,
        "selectivity_runs": sel_runs,
        "parity_in_run": parity,
        # the context's device-memory arena (include/knoxgpu.h knox_ctx_set_arena; KNOX_PLACEMENT=0 = plain hipMalloc): the
        # bitset is a write buffer chosen by running the scan on the (column, buffer) pair; nothing is held outside it
        "placement": {"arena_enabled": ai["enabled"], "arena_budget_gib": ai["budget_gib"], "arena_mapped_gib": ai["mapped_gib"],
                      "arena_used_gib": ai["used_gib"], "write_buffers_created": ai["write_buffers"], "spare_write_buffer_gib": ai["spare_write_buffer_gib"],
                      "chunks_created": ai["chunks"], "probe_measurements": ai["probes"], "mixed_zone_gib": ai["mixed_zone_gib"],
                      "classes_seen": classes_seen, "column_space_gib_per_class": class_gib,
                      "bitset_pair_check_at_allocation": {"scan_over_read_only": bits.pair_check[0], "candidate_windows": bits.pair_check[1]},
                      "held_outside_arena_bytes": ctx.placement_info()["held_bytes"], "setup_checks": checks, "device_quiet_ms": quiet, "setup_s": t_setup},
    }
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr) and args.rows_log2 == 30:
        try:
            tj = json.load(open(tr))
            out["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = "profiles/traffic.json (%s; rocprofv3 --pmc passes of this command, tools/profile_round.sh)" % tj.get("measured", "round 1")
            out["_traffic_kernels"] = tj.get("kernels", {})
        except Exception:
            pass
    pack_counts_host = res.pack_counts
    col.free()      # back to the arena's free lists, not to the driver
    bits.free()
    # BASELINE configs[2] / configs[3] at the same size, in the same context; one extra key of the same JSON line, nothing
    # above depends on it
    if rank == 0 and world == 1 and not args.no_configs:
        try:
            out["configs"] = secondary_configs(K, ctx, n_rows, pack_rows, args.steps, args.warmup, parity=not args.no_cpu_baseline)
            for name, t in out.pop("_traffic_kernels", {}).items():   # PMC-measured HBM bytes per launch of the same kernels (profiles/traffic.json)
                if name in out["configs"]:
                    out["configs"][name]["traffic"] = t.get("hbm_bytes_per_launch")
        except Exception as e:   # a secondary config must never cost the headline its line
            import traceback
            traceback.print_exc()
            out["configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # the arena's cost / benefit in the same record (round-3 review item 6): what the probes cost this context, and the three
    # big configs once more without placement
    ai2 = ctx.arena_info()
    out["placement"].update({"probe_launches": ai2["probe_launches"], "probe_ms": ai2["probe_ms"], "probe_measurements": ai2["probes"],
                             "probe_budget": "200 launches / 50 ms per context (the library's default, knox_ctx_set_probe_budget)",
                             "chunks_labelled_by_inheritance": ai2["unprobed_chunks"], "write_buffers_placed_by_label_only": ai2["unverified_write_buffers"],
                             "chunks_created": ai2["chunks"], "arena_mapped_gib": ai2["mapped_gib"],
                             "device_quiet_wait_s": quiet_wait_s if quiet else 0.0, "device_quiet_ended_by": quiet_ended_by if quiet else None})
    if rank == 0 and world == 1 and not args.no_configs and not args.no_placement_ab and n_rows >= (1 << 26):
        try:
            on = {"headline": scan_ms}
            for k_ in ("cfg3", "cfg4_bytes"):
                if isinstance(out.get("configs"), dict) and k_ in out["configs"]:
                    on[k_] = out["configs"][k_].get("kernel_ms")
            out["placement_off"] = placement_off_ab(K, local_rank, n_rows, pack_rows, on)
            print("[placement A/B] " + json.dumps({k_: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v_.items()}
                                                    for k_, v_ in out["placement_off"].items() if isinstance(v_, dict)}), file=sys.stderr, flush=True)
        except Exception as e:   # diagnostics must never cost the headline its line
            out["placement_off"] = {"error": "%s: %s" % (type(e).__name__, e)}
    ctx.close()
    if rank == 0 and world == 1 and args.rows_log2 == 30 and not args.no_live_traffic and not args.no_cpu_baseline:
        t_pmc = time.perf_counter()
        live, detail = measure_traffic_live(args.rows_log2)
        if live is not None:
            out["roofline"]["traffic"] = live
            out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over the same scan (bench.py "
                                                 "--pmc-child), FETCH x 2 (gfx950) + WRITE, %.1f s" % (time.perf_counter() - t_pmc))
            out["roofline"]["traffic_detail"] = detail
        else:
            out["roofline"]["traffic_live_failed"] = detail
        print("[traffic] %s" % (("%.4g B per launch = %.4f x algorithmic" % (live, live / (n_rows * BYTES_PER_ROW))) if live is not None else detail), file=sys.stderr, flush=True)
    if rank == 0 and not args.no_cpu_baseline:
        # rank 0 only, after the timed region and its closing barrier (the other ranks are done and wait in destroy)
        sample = min(n_rows, 1 << args.cpu_sample_log2)
        out["cpu_baseline"] = cpu_baseline(sample, pack_rows, a, b, pack_counts_host)
    out.pop("_traffic_kernels", None)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
