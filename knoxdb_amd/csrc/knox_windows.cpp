// knox_windows.cpp -- time-window (NativeBucket) aggregation entry points, the literal sequential reducers, and
// NativeBucket.Emit's gap filling (pure host code).
#include "knox_internal.h"

static const int kWindowsSumsOnly = 1 << 21;  // internal flag: only Sum / Count / Mean / Var / Std will be read from the states
static const int kWindowsNoRowIds = 1 << 20;  // internal flag (never from the ABI: callers' flags are masked): the states' row ids are not wanted

// windows -> states on the device: each window's row range [d_start[k], d_start[k+1]) of `vals` is cut into work
// items (planned on the device: no host round trip of the bounds), reduced, and merged per window in row order.
// *states_dev = AggStateDev[n_buckets] inside ctx->tmp_vals.  n_rows bounds the item count.
static int reduce_windows(knox_ctx *ctx, const void *vals, int dtype, const uint32_t *mask, size_t n_rows, uint64_t row_base,
                          const uint64_t *d_start, size_t n_buckets, int flags, char **states_dev) {
    const int width = dtype_width(dtype);
    const uint32_t chunk_shift = 16;
    const size_t max_items = n_buckets + (n_rows >> chunk_shift) + 1;
    if (max_items > 0xfffffff0ull) return KNOX_E_RANGE;
    uint32_t plan_per = 0;
    const uint32_t plan_blocks = window_plan_blocks((uint32_t)n_buckets, &plan_per);
    const size_t off_istart = 0, off_bsum = ((n_buckets + 1) * 4 + 63) & ~(size_t)63;
    const size_t off_items = (off_bsum + ((size_t)plan_blocks + 1) * 4 + 63) & ~(size_t)63;
    const size_t off_partials = (off_items + max_items * sizeof(BucketItem) + 63) & ~(size_t)63;
    const size_t off_states = (off_partials + max_items * sizeof(AggPartial) + 63) & ~(size_t)63;
    int rc = ensure(ctx, ctx->tmp_vals, off_states + n_buckets * sizeof(AggStateDev) + 64);
    if (rc) return rc;
    char *base = (char *)ctx->tmp_vals.p;
    uint32_t *d_istart = (uint32_t *)(base + off_istart);
    hipError_t e = launch_window_plan(ctx->stream, d_start, (uint32_t)n_buckets, chunk_shift, (uint32_t *)(base + off_bsum), d_istart);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    uint32_t n_items32 = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n_items32, d_istart + n_buckets, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n_items = n_items32;
    if (n_items > max_items) return KNOX_E_RANGE;  // cannot happen: the bound above is exact arithmetic
    e = launch_window_items(ctx->stream, d_start, d_istart, (uint32_t)n_buckets, chunk_shift, n_items32, (BucketItem *)(base + off_items));
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.agg_col = vals;
    a.agg_flavor = dtype_flavor(dtype);
    a.mask = mask;
    a.n_rows = n_rows;
    a.row_base = row_base;
    a.partials = (AggPartial *)(base + off_partials);
    const int grid = (int)std::min<size_t>(n_items, (size_t)ctx->n_cus * 8);
    // many small windows: one wave per item instead of one workgroup per item, or -- with enough windows to fill the chip four to a
    // wave -- 16 lanes per window (knox_window.hip).  Measured on the same resident columns (tools/ab_window_forms.py, values path,
    // 2^26 ... 2^30 rows): at 64 K rows per window the workgroup form wins, at 16 K the group form by a little, at 4 K and below wave < block and
    // group < wave (2^20 windows: 2^26 rows 2.21 -> 0.75 ms, 2^28 rows 2.34 -> 1.10 ms, 2^30 rows 3.42 -> 2.35 ms).
    // KNOX_WINDOW_FORM=block | wave | group pins the form (measurements, tests).
    const bool small_items = n_items > (size_t)ctx->n_cus * 8 && n_rows / n_items < 32768;
    const char *form_env = getenv("KNOX_WINDOW_FORM");
    const bool pin_wave = form_env && !strcmp(form_env, "wave"), pin_group = form_env && !strcmp(form_env, "group");
    const bool pin_block = form_env && !strcmp(form_env, "block");
    const bool by_group = !pin_block && !pin_wave && (pin_group || (small_items && n_items >= (size_t)ctx->n_cus * 32));
    const int group_grid = (int)std::min<size_t>((n_items + 15) / 16, (size_t)ctx->n_cus * 16);
    if (by_group)
        e = launch_bucket_reduce_groups(ctx->stream, width, group_grid, a,
                                        (const BucketItem *)(base + off_items), (uint32_t)n_items, (flags & kWindowsSumsOnly) != 0);
    else if (!pin_block && (pin_wave || small_items))
        e = launch_bucket_reduce_waves(ctx->stream, width, (int)std::min<size_t>((n_items + 3) / 4, (size_t)ctx->n_cus * 16), a,
                                       (const BucketItem *)(base + off_items), (uint32_t)n_items);
    else
        e = launch_bucket_reduce(ctx->stream, width, grid, a, (const BucketItem *)(base + off_items), (uint32_t)n_items);
    if (e == hipSuccess) {
        // values wanted and no var / std pass to come: the finalize kernel emits Value() itself and the states are never stored
        const bool fuse_values = ctx->win_values != nullptr && !(flags & 2);
        e = launch_bucket_finalize(ctx->stream, width, a.partials, (const uint32_t *)(base + off_istart), (uint32_t)n_buckets, vals, row_base, dtype,
                                   fuse_values ? nullptr : (AggStateDev *)(base + off_states), ctx->win_func, fuse_values ? ctx->win_values : nullptr,
                                   fuse_values ? ctx->win_seen : nullptr);
        ctx->win_values_done = fuse_values && e == hipSuccess;
    }
    if (e == hipSuccess && (flags & 2)) {
        // var/std: second pass over the same work items, each centred on its window's mean
        rc = ensure(ctx, ctx->mom_partials, (n_items + 1) * 4 * sizeof(double));
        if (rc) return rc;
        e = launch_bucket_moments(ctx->stream, width, a.agg_flavor, by_group ? group_grid : grid, vals, a.mask, by_group ? (uint64_t)n_rows : 0,
                                  (const BucketItem *)(base + off_items), (uint32_t)n_items,
                                  (const uint32_t *)(base + off_istart), (uint32_t)n_buckets, (AggStateDev *)(base + off_states),
                                  (double *)ctx->mom_partials.p);
    }
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    *states_dev = base + off_states;
    return KNOX_OK;
}

// out-of-order timestamps (bucket_native.go:123-141 finds / inserts the window's reducer by time): the selected rows
// are brought into window order by a STABLE radix sort of (window index, row id, value) triples (knox_sort.hip) -- rows
// of one window keep their stream order, which is all the reducers can observe -- and the ordered pipeline runs on the
// sorted copy of the values; the row ids inside the states are mapped back through the sorted row ids.
// smallest / largest finite non-zero biased float64 exponent of a float column: one pass over the values the first time, kept with the
// column afterwards (col_fexp_*); emax 0 = nothing finite and non-zero
static int col_float_exprange(knox_ctx *ctx, const knox_col *col, int *emax_out, int *emin_out) {
    int emax = col_fexp_cached(col), emin = col->fexp_min;
    if (emax < 0) {
        int rc = ensure(ctx, ctx->misc, 256);
        if (rc) return rc;
        uint32_t *d_exp = (uint32_t *)((char *)ctx->misc.p + 96);
        HIPCHK(ctx, launch_zero_bytes(ctx->stream, d_exp, 8));
        hipError_t e = launch_float_exprange(ctx->stream, col->width, col->dtype, col->dev, col->n_rows, d_exp);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        uint32_t h[2] = {0, 0};
        HIPCHK(ctx, hipMemcpyAsync(h, d_exp, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        emax = (int)h[0];
        emin = emax ? (int)(~h[1]) : 0;
        col_fexp_set(col, emax, emin);
    }
    *emax_out = emax;
    *emin_out = emin;
    return KNOX_OK;
}

static int reduce_buckets_unordered(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                    int64_t step, size_t n_buckets, uint64_t row_base, int flags, char **states_dev) {
    const size_t n_rows = value_col->n_rows;
    if (n_rows > 0xfffffff0ull) return KNOX_E_UNSUPPORTED;
    const size_t w = (size_t)value_col->width;
    // Value() of a reducer with an order-free form -- count, first / last (the value of the lowest / highest selected row), min / max
    // (floats: with the reference's NaN and signed-zero rules), and over integer columns the sum and the mean rounded from the exact sum -- over as many windows as a workgroup's LDS holds a table
    // for (8 B + a bit per window: ~19 Ki windows; Count ~39 Ki, Mean ~13 Ki / ~7.9 Ki for 8-byte values; up to four times that in ranges
    // of windows, each of which reads all rows -- window_direct_geom): every row is pushed into
    // its window where it lies (window_direct_kernel, knox_fold.hip), as NativeBucket.Push does; nothing is sorted.
    // KNOX_WINDOW_DIRECT=0 keeps the sort (measurements, the agreement test).
    if (ctx->win_values != nullptr && !(flags & 2) && n_buckets <= (1u << 18)) {
        const char *env = getenv("KNOX_WINDOW_DIRECT");
        const WindowDirectGeom dg = (env && atoi(env) == 0) ? WindowDirectGeom{0, 0, 0, 0, 1, 0, 0}
                                                            : window_direct_geom((int)w, value_col->dtype, ctx->win_func, (uint32_t)n_buckets, ctx->n_cus, mask != nullptr);
        if (dg.grid != 0) {
            int rc = ensure(ctx, ctx->sort_buf, dg.scratch);
            if (rc) return rc;
            hipError_t e = launch_window_direct(ctx->stream, (int)w, value_col->dtype, ctx->win_func, dg, (const int64_t *)time_col->dev, value_col->dev,
                                                mask ? (const uint8_t *)mask->dev : nullptr, n_rows, t_from, step, (uint32_t)n_buckets, ctx->sort_buf.p,
                                                ctx->win_values, ctx->win_seen);
            if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
            ctx->win_path = KNOX_WINDOWS_DIRECT_VALUES;
            ctx->win_values_done = true;
            *states_dev = nullptr;
            return KNOX_OK;
        }
    }
    // var / std: exact sums of the values and of their squares, n * S2 - S1^2 as an integer (window_fvar_kernel).  Integer columns enter
    // as Go's float64(v) -- what the reference's VarReducer works on: exact up to 2^53, rounded to nearest above
    if (ctx->win_values != nullptr && (ctx->win_func == KNOX_R_VAR || ctx->win_func == KNOX_R_STD) && n_buckets <= 8192 &&
        value_col->dtype >= KNOX_INT64 && value_col->dtype <= KNOX_FLOAT32) {
        const char *env = getenv("KNOX_WINDOW_DIRECT");
        if (!(env && atoi(env) == 0)) {
            int emax = 0, emin = 0;
            int rc = col_float_exprange(ctx, value_col, &emax, &emin);
            if (rc) return rc;
            const WindowFvarGeom vg = window_fvar_geom((uint32_t)emax, (uint32_t)emin, (uint32_t)n_buckets, ctx->n_cus);
            if (vg.grid != 0) {
                rc = ensure(ctx, ctx->sort_buf, vg.scratch);
                if (rc) return rc;
                hipError_t e = launch_window_fvar(ctx->stream, (int)w, value_col->dtype, ctx->win_func, vg, (const int64_t *)time_col->dev, value_col->dev,
                                                  mask ? (const uint8_t *)mask->dev : nullptr, n_rows, t_from, step, (uint32_t)n_buckets, ctx->sort_buf.p,
                                                  ctx->win_values, ctx->win_seen);
                if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
                ctx->win_path = KNOX_WINDOWS_DIRECT_FLOAT_SUMS;
                ctx->win_values_done = true;
                *states_dev = nullptr;
                return KNOX_OK;
            }
        }
    }
    // float sums / means: exact fixed-point adders sized by the column's smallest and largest exponent (one pass over the values the first
    // time, kept with the column); a column spanning more than 107 binades, or more windows than LDS holds adders for, takes the sort
    if (ctx->win_values != nullptr && !(flags & 2) && (ctx->win_func == KNOX_R_SUM || ctx->win_func == KNOX_R_MEAN) &&
        (value_col->dtype == KNOX_FLOAT64 || value_col->dtype == KNOX_FLOAT32) && n_buckets <= 32768) {
        const char *env = getenv("KNOX_WINDOW_DIRECT");
        if (!(env && atoi(env) == 0)) {
            int emax = 0, emin = 0;
            int rc = col_float_exprange(ctx, value_col, &emax, &emin);
            if (rc) return rc;
            const WindowFsumGeom fg = window_fsum_geom((uint32_t)emax, (uint32_t)emin, (uint32_t)n_buckets, ctx->n_cus, mask != nullptr);
            if (fg.grid != 0) {
                rc = ensure(ctx, ctx->sort_buf, fg.scratch);
                if (rc) return rc;
                hipError_t e = launch_window_fsum(ctx->stream, (int)w, value_col->dtype, ctx->win_func, fg, (const int64_t *)time_col->dev, value_col->dev,
                                                  mask ? (const uint8_t *)mask->dev : nullptr, n_rows, t_from, step, (uint32_t)n_buckets, ctx->sort_buf.p,
                                                  ctx->win_values, ctx->win_seen);
                if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
                ctx->win_path = KNOX_WINDOWS_DIRECT_FLOAT_SUMS;
                ctx->win_values_done = true;
                *states_dev = nullptr;
                return KNOX_OK;
            }
        }
    }
    // ... and the whole state of every window the same way (integer columns, no var / std pass to follow, ~3600-4400 windows at most):
    // window_state_kernel, one walk below 8 bytes, two for 8-byte values
    if (!(flags & 2) && n_buckets <= 8192) {
        const char *env = getenv("KNOX_WINDOW_DIRECT");
        const WindowStateGeom sg = (env && atoi(env) == 0) ? WindowStateGeom{0, 0, 0, 0, 0} : window_state_geom((int)w, value_col->dtype, (uint32_t)n_buckets, ctx->n_cus);
        if (sg.grid != 0) {
            int rc = ensure(ctx, ctx->sort_buf, sg.scratch);
            if (!rc) rc = ensure(ctx, ctx->tmp_vals, n_buckets * sizeof(AggStateDev) + 64);
            if (rc) return rc;
            hipError_t e = launch_window_state(ctx->stream, (int)w, value_col->dtype, sg, (const int64_t *)time_col->dev, value_col->dev,
                                               mask ? (const uint8_t *)mask->dev : nullptr, n_rows, t_from, step, (uint32_t)n_buckets, row_base, ctx->sort_buf.p,
                                               (AggStateDev *)ctx->tmp_vals.p);
            if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
            ctx->win_path = KNOX_WINDOWS_DIRECT_STATES;
            *states_dev = (char *)ctx->tmp_vals.p;
            return KNOX_OK;
        }
    }
    // ... float columns too: the exact adders for the sum, order keys with the reference's NaN / signed-zero rules for min / max
    if (!(flags & 2) && n_buckets <= 8192 && (value_col->dtype == KNOX_FLOAT64 || value_col->dtype == KNOX_FLOAT32)) {
        const char *env = getenv("KNOX_WINDOW_DIRECT");
        if (!(env && atoi(env) == 0)) {
            int emax = 0, emin = 0;
            int rc = col_float_exprange(ctx, value_col, &emax, &emin);
            if (rc) return rc;
            const WindowFsumGeom sg = window_fstate_geom((uint32_t)emax, (uint32_t)emin, (uint32_t)n_buckets, ctx->n_cus);
            if (sg.grid != 0) {
                rc = ensure(ctx, ctx->sort_buf, sg.scratch);
                if (!rc) rc = ensure(ctx, ctx->tmp_vals, n_buckets * sizeof(AggStateDev) + 64);
                if (rc) return rc;
                hipError_t e = launch_window_fstate(ctx->stream, (int)w, value_col->dtype, sg, (const int64_t *)time_col->dev, value_col->dev,
                                                    mask ? (const uint8_t *)mask->dev : nullptr, n_rows, t_from, step, (uint32_t)n_buckets, row_base,
                                                    ctx->sort_buf.p, (AggStateDev *)ctx->tmp_vals.p);
                if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
                ctx->win_path = KNOX_WINDOWS_DIRECT_STATES;
                *states_dev = (char *)ctx->tmp_vals.p;
                return KNOX_OK;
            }
        }
    }
    unsigned key_bits = 1;
    while ((1ull << key_bits) < n_buckets) key_bits++;  // keys 0..n_buckets-1 (rows outside every window leave in the first pass)
    uint32_t bits[4] = {0, 0, 0, 0};
    int n_passes = 0;
    window_sort_passes(key_bits, bits, &n_passes);
    const WindowSortGeom g0 = window_sort_geom(n_rows, ctx->n_cus);
    // scratch (idx_scratch): window bounds | histogram | digit totals;  idx_rows: keys of all rows;  sort_buf: the triples
    const size_t off_hist = ((n_buckets + 2) * 8 + 255) & ~(size_t)255;
    const size_t off_tot = off_hist + (((size_t)g0.nb * 256 * 4 + 255) & ~(size_t)255);
    int rc = ensure(ctx, ctx->idx_scratch, off_tot + 256 * 4 + 64);
    const size_t keys0_bytes = (n_rows * 4 + 255) & ~(size_t)255;
    if (!rc) rc = ensure(ctx, ctx->idx_rows, keys0_bytes);
    if (rc) return rc;
    char *sa = (char *)ctx->idx_scratch.p;
    uint64_t *d_start = (uint64_t *)sa;
    uint32_t *d_hist = (uint32_t *)(sa + off_hist), *d_tot = (uint32_t *)(sa + off_tot);
    // A selection of well under all rows: compact the selected rows' (window index, [row id], value) in row order with a streaming
    // pass first, so that the first scatter ranks the selected rows only (knox_sort.hip).  Decided by the selection's size -- one
    // pass over the bitset -- ; KNOX_WINDOW_COMPACT=0 / 1 pins the choice (measurements, tests).
    const bool want_rows = (flags & kWindowsNoRowIds) == 0;  // Value() per window never shows a row id
    hipError_t e = hipSuccess;
    size_t n_in = n_rows;                         // rows the first scatter walks
    const uint32_t *keys_in = (const uint32_t *)ctx->idx_rows.p;
    const void *vals_in = value_col->dev;
    const uint32_t *idx_in = nullptr;
    size_t cset = 0;                              // bytes of the compacted (row id, value) arrays at the head of sort_buf
    WindowSortGeom gfirst = g0;
    bool compacted = false;
    if (mask != nullptr) {
        const char *env = getenv("KNOX_WINDOW_COMPACT");
        const WindowCompactGeom cg = window_compact_geom(n_rows, ctx->n_cus);
        rc = ensure(ctx, ctx->misc, 256);
        if (!rc) rc = ensure(ctx, ctx->pack_skip, ((size_t)cg.n_units + 2) * 4 + 64);
        if (rc) return rc;
        uint32_t *d_units = (uint32_t *)ctx->pack_skip.p;
        unsigned long long *d_total = (unsigned long long *)((char *)ctx->misc.p + 64);
        e = launch_mask_unit_offsets(ctx->stream, (const uint8_t *)mask->dev, n_rows, cg, d_units, d_total);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        unsigned long long n_m = 0;
        HIPCHK(ctx, hipMemcpyAsync(&n_m, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        // up to a third selected (2^28 rows -> 16384 windows, whole calls without / with: 5 % 2.27 / 1.39 ms, 25 % 2.95 / 2.7, 40 % 3.74 / 3.80,
        // 50 % 4.10 / 4.38, 60 % 4.58 / 5.08)
        const bool use = env ? atoi(env) != 0 : (n_m * 3 <= (unsigned long long)n_rows);
        if (use) {
            const size_t ck = ((size_t)n_m * 4 + 255) & ~(size_t)255, cw = ((size_t)n_m * w + 255) & ~(size_t)255;
            cset = (want_rows ? ck : 0) + cw;
            const size_t set_max = ck + (want_rows ? ck : 0) + cw;   // a sort set holds at most the compacted rows
            rc = ensure(ctx, ctx->sort_buf, cset + (n_passes > 1 ? 2 : 1) * set_max + 256);
            if (rc) return rc;
            char *cb = (char *)ctx->sort_buf.p;
            uint32_t *idx_c = want_rows ? (uint32_t *)cb : nullptr;
            void *vals_c = cb + (want_rows ? ck : 0);
            e = launch_window_compact(ctx->stream, (int)w, (const int64_t *)time_col->dev, value_col->dev, (const uint8_t *)mask->dev, n_rows, cg, d_units,
                                      t_from, step, (uint32_t)n_buckets, (uint32_t *)ctx->idx_rows.p, idx_c, vals_c);
            gfirst = window_sort_geom((size_t)n_m, ctx->n_cus);
            if (e == hipSuccess) e = launch_compact_keys_hist(ctx->stream, (const uint32_t *)ctx->idx_rows.p, (size_t)n_m, gfirst, bits[0], d_hist, d_tot);
            n_in = (size_t)n_m;
            vals_in = vals_c;
            idx_in = idx_c;
            compacted = true;
            ctx->win_path = KNOX_WINDOWS_SORTED_COMPACTED;
        }
    }
    if (!compacted)
        e = launch_window_sort_begin(ctx->stream, (const int64_t *)time_col->dev, n_rows, mask ? (const uint8_t *)mask->dev : nullptr, t_from,
                                     step, (uint32_t)n_buckets, g0, bits[0], (uint32_t *)ctx->idx_rows.p, d_hist, d_tot);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    uint32_t tot[256];
    HIPCHK(ctx, hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    size_t n_sel = 0;
    for (uint32_t t : tot) n_sel += t;
    // the triples (key, row id, value) -- or pairs, when no row ids are wanted: two sets when there is more than one pass
    const size_t arrk = (n_sel * 4 + 255) & ~(size_t)255, arr4 = want_rows ? arrk : 0;
    const size_t arrw = (n_sel * w + 255) & ~(size_t)255;
    const size_t set = arrk + arr4 + arrw;
    if (!compacted) {
        rc = ensure(ctx, ctx->sort_buf, (n_passes > 1 ? 2 : 1) * set + 256);   // (compacted: sized above, the buffer must not move any more)
        if (rc) return rc;
    }
    char *cur = (char *)ctx->sort_buf.p + cset, *oth = cur + set;
    auto K = [&](char *b) { return (uint32_t *)b; };
    auto I = [&](char *b) { return want_rows ? (uint32_t *)(b + arrk) : (uint32_t *)nullptr; };
    auto V = [&](char *b) { return (void *)(b + arrk + arr4); };
    e = launch_window_sort_first(ctx->stream, (int)w, keys_in, vals_in, n_in, gfirst, bits[0], d_hist, d_tot, K(cur), I(cur), V(cur), idx_in);
    const WindowSortGeom g1 = window_sort_geom(n_sel, ctx->n_cus);  // never more workgroups than g0: the histogram fits
    uint32_t shift = bits[0];
    for (int p = 1; p < n_passes && e == hipSuccess; p++) {
        e = launch_window_sort_pass(ctx->stream, (int)w, K(cur), I(cur), V(cur), n_sel, g1, shift, bits[p], d_hist, d_tot, K(oth), I(oth), V(oth));
        std::swap(cur, oth);
        shift += bits[p];
    }
    if (e == hipSuccess) e = launch_key_bounds(ctx->stream, K(cur), n_sel, (uint32_t)n_buckets, d_start);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    char *st = nullptr;
    rc = reduce_windows(ctx, V(cur), value_col->dtype, nullptr, n_sel, 0, d_start, n_buckets, flags, &st);
    if (rc) return rc;
    if (want_rows) e = launch_remap_rows(ctx->stream, (AggStateDev *)st, (uint32_t)n_buckets, I(cur), row_base);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    *states_dev = st;
    return KNOX_OK;
}

// the windows' states on the device (ordered time: reduced in place; out-of-order time: the sorted copy)
static int reduce_buckets_dev(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                              int64_t step, size_t n_buckets, uint64_t row_base, int flags, char **states_dev) {
    if (!ctx || !time_col || !value_col || step <= 0 || n_buckets == 0 || n_buckets > (1u << 26)) return KNOX_E_INVALID;
    if (time_col->dtype != KNOX_INT64 || value_col->dtype == KNOX_BYTES || time_col->n_rows != value_col->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != value_col->n_rows) return KNOX_E_INVALID;
    // every window edge must be representable
    if ((__int128)t_from + (__int128)step * (__int128)n_buckets > (__int128)INT64_MAX) return KNOX_E_RANGE;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, time_col); if (rc_) return rc_; }
    { const int rc_ = col_acquire(ctx, value_col); if (rc_) return rc_; }
    const size_t n_rows = value_col->n_rows;
    int rc = ensure(ctx, ctx->misc, 256);
    if (!rc) rc = ensure(ctx, ctx->idx_scratch, (n_buckets + 2) * 8 + 64);
    if (rc) return rc;
    // 1. order check (one streaming pass over the time column) unless the caller vouches for it
    //    -- or the column was checked before: the answer is a property of its content and is kept with it (col_sorted_cached:
    //    columns the library filled itself, until something writes into them again).  A time column is queried again and
    //    again; the check is 8 B per row, half of what an ordered window query reads.  flags bit4 (16): check anyway.
    uint32_t *d_flag = (uint32_t *)((char *)ctx->misc.p + 32);
    uint32_t unsorted = 0;
    const int known = (flags & 16) ? -1 : col_sorted_cached(time_col);
    const bool check = !(flags & 1) && known < 0;
    if (known == 0 && !(flags & 1)) unsorted = 1;
    if (check) {
        HIPCHK(ctx, launch_zero_bytes(ctx->stream, d_flag, 4));
        hipError_t e = launch_sorted_check(ctx->stream, (const int64_t *)time_col->dev, n_rows, d_flag);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        HIPCHK(ctx, hipMemcpyAsync(&unsorted, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    // 2. first row of every window edge (stays on the device)
    uint64_t *d_start = (uint64_t *)ctx->idx_scratch.p;
    hipError_t e = launch_bucket_bounds(ctx->stream, (const int64_t *)time_col->dev, n_rows, t_from, step, n_buckets, d_start);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    if (check) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        col_sorted_set(time_col, unsorted == 0);
    }
    // (Round 4: for <= 1024 windows over integer columns a path WITHOUT the sort was built -- every state field as an order-free fold,
    //  LDS atomics into a per-workgroup table, two streaming passes, the second one for the lowest row holding each window's min / max --
    //  and measured at 2^28 shuffled rows -> 1024 windows: 4.07 ms against the sort's 2.88 ms.  Eight 64-bit LDS atomics per selected
    //  row ran at well under one per clock and CU; dropped, in the history at b06218a.)
    ctx->win_path = unsorted ? KNOX_WINDOWS_SORTED : KNOX_WINDOWS_ORDERED;
    if (unsorted) return reduce_buckets_unordered(ctx, time_col, value_col, mask, t_from, step, n_buckets, row_base, flags, states_dev);
    // 3. reduce the windows' row ranges
    return reduce_windows(ctx, value_col->dev, value_col->dtype, mask ? (const uint32_t *)mask->dev : nullptr, n_rows, row_base, d_start, n_buckets,
                          flags, states_dev);
}

KNOX_API int knox_windows_last_path(const knox_ctx *ctx) { return ctx ? ctx->win_path : KNOX_E_INVALID; }

KNOX_API int knox_reduce_buckets(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                 int64_t step, size_t n_buckets, uint64_t row_base, int flags, knox_agg_state *states) {
    if (!states) return KNOX_E_INVALID;
    char *st = nullptr;
    int rc = reduce_buckets_dev(ctx, time_col, value_col, mask, t_from, step, n_buckets, row_base, flags & 0xffff, &st);
    if (rc) return rc;
    return download(ctx, states, st, n_buckets * sizeof(AggStateDev));
}

// The mergeable states themselves into the context's page-locked result buffer (a view, like knox_bits_indexes_view):
// for hosts that merge windows across GPUs.  144 B per window cross PCIe at DMA rate instead of through pageable memory.
KNOX_API int knox_reduce_buckets_view(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                      int64_t step, size_t n_buckets, uint64_t row_base, int flags, const knox_agg_state **states_view) {
    if (!states_view) return KNOX_E_INVALID;
    char *st = nullptr;
    int rc = reduce_buckets_dev(ctx, time_col, value_col, mask, t_from, step, n_buckets, row_base, flags & 0xffff, &st);
    if (rc) return rc;
    const size_t bytes = n_buckets * sizeof(AggStateDev);
    rc = ensure_host(ctx, ctx->h_view, ctx->h_view_cap, bytes);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_view, st, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *states_view = (const knox_agg_state *)ctx->h_view;
    return KNOX_OK;
}

// Reducer.Value() of every window, finalised on the device: 8 (value) + 1 (seen) bytes per window cross PCIe into the
// context's page-locked result buffer; the caller gets views (valid until the next *_view / *_values call on the ctx)
KNOX_API int knox_reduce_buckets_values(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                        int64_t step, size_t n_buckets, int func, int flags, const void **values_view,
                                        const uint8_t **seen_view) {
    if (!values_view || !seen_view || func < KNOX_R_SUM || func > KNOX_R_COUNT) return KNOX_E_INVALID;
    if (func == KNOX_R_VAR || func == KNOX_R_STD) flags |= 2;
    const bool sums = func == KNOX_R_SUM || func == KNOX_R_COUNT || func == KNOX_R_MEAN || func == KNOX_R_VAR || func == KNOX_R_STD;
    if (!ctx || !value_col || n_buckets == 0 || n_buckets > (1u << 26)) return KNOX_E_INVALID;
    const size_t w = (size_t)value_col->width;
    const size_t off_seen = (n_buckets * w + 63) & ~(size_t)63, bytes = off_seen + n_buckets;
    int rc = ensure(ctx, ctx->pack_counts, bytes + 64);
    if (!rc) rc = ensure_host(ctx, ctx->h_view, ctx->h_view_cap, bytes);
    if (rc) return rc;
    char *d = (char *)ctx->pack_counts.p;
    char *st = nullptr;
    ctx->win_values = d;
    ctx->win_seen = (uint8_t *)d + off_seen;
    ctx->win_func = func;
    ctx->win_values_done = false;
    rc = reduce_buckets_dev(ctx, time_col, value_col, mask, t_from, step, n_buckets, 0, (flags & 0xffff) | kWindowsNoRowIds | (sums ? kWindowsSumsOnly : 0), &st);
    ctx->win_values = nullptr;
    if (rc) return rc;
    if (!ctx->win_values_done) {
        hipError_t e = launch_window_values(ctx->stream, (int)w, (const AggStateDev *)st, (uint32_t)n_buckets, func, d, (uint8_t *)d + off_seen);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_view, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *values_view = ctx->h_view;
    *seen_view = (const uint8_t *)ctx->h_view + off_seen;
    return KNOX_OK;
}

KNOX_API int knox_reduce_buckets_sequential(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask,
                                            int64_t t_from, int64_t step, size_t n_buckets, int func, int flags, void *values_out,
                                            uint8_t *seen_out) {
    if (!ctx || !time_col || !value_col || !values_out || !seen_out || step <= 0 || n_buckets == 0 || n_buckets > (1u << 26)) return KNOX_E_INVALID;
    if (time_col->dtype != KNOX_INT64 || value_col->dtype == KNOX_BYTES || time_col->n_rows != value_col->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != value_col->n_rows) return KNOX_E_INVALID;
    if (func < 0 || func > KNOX_R_STD_JOIN) return KNOX_E_INVALID;
    if ((__int128)t_from + (__int128)step * (__int128)n_buckets > (__int128)INT64_MAX) return KNOX_E_RANGE;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, time_col); if (rc_) return rc_; }
    { const int rc_ = col_acquire(ctx, value_col); if (rc_) return rc_; }
    const size_t n_rows = value_col->n_rows;
    const size_t off_vals = ((n_buckets + 2) * 8 + 63) & ~(size_t)63;
    const size_t off_seen = off_vals + ((n_buckets * 8 + 63) & ~(size_t)63);
    int rc = ensure(ctx, ctx->misc, 256);
    if (!rc) rc = ensure(ctx, ctx->idx_scratch, off_seen + n_buckets + 64);
    if (rc) return rc;
    uint32_t *d_flag = (uint32_t *)((char *)ctx->misc.p + 32);
    const int known = (flags & 16) ? -1 : col_sorted_cached(time_col);  // as in reduce_buckets_dev
    if (known == 0 && !(flags & 1)) return KNOX_E_UNSUPPORTED;
    const bool check = !(flags & 1) && known < 0;
    if (check) {
        HIPCHK(ctx, launch_zero_bytes(ctx->stream, d_flag, 4));
        hipError_t e = launch_sorted_check(ctx->stream, (const int64_t *)time_col->dev, n_rows, d_flag);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    }
    char *base = (char *)ctx->idx_scratch.p;
    uint64_t *d_start = (uint64_t *)base;
    hipError_t e = launch_bucket_bounds(ctx->stream, (const int64_t *)time_col->dev, n_rows, t_from, step, n_buckets, d_start);
    // the window bounds never leave the device: the reducer lanes read them in place
    // one lane per window when there are enough windows to give every lane of the chip one, a whole wave per window (coalesced
    // fetches, the recurrence fed through v_readlane) below that; flags bit 2 / bit 3 force the one or the other (tests, A/B)
    const bool wave_per_window = (flags & 8) != 0 || ((flags & 4) == 0 && n_buckets <= (size_t)ctx->n_cus * 128);  // measured (tools/bench_windows.py, 2^26 / 2^28 rows): 16384 windows 1.7-1.8 x faster by wave, 65536 windows 1.25-1.3 x faster by lane
    if (e == hipSuccess)
        e = launch_bucket_sequential(ctx->stream, value_col->dtype, func, (const int64_t *)time_col->dev, value_col->dev,
                                     mask ? (const uint8_t *)mask->dev : nullptr, d_start, (uint32_t)n_buckets, wave_per_window ? 1 : 0,
                                     base + off_vals, (uint8_t *)(base + off_seen));
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    uint32_t unsorted = 0;
    if (check) {
        HIPCHK(ctx, hipMemcpyAsync(&unsorted, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        col_sorted_set(time_col, unsorted == 0);
        if (unsorted) return KNOX_E_UNSUPPORTED;
    }
    rc = download(ctx, values_out, base + off_vals, n_buckets * (size_t)value_col->width);
    if (!rc) rc = download(ctx, seen_out, base + off_seen, n_buckets);
    return rc;
}

KNOX_API int knox_reduce_sequential(knox_ctx *ctx, const knox_col *col, const knox_bits *mask, int func, void *value_out, int *seen,
                                    int64_t *row) {
    if (!ctx || !col || !value_out || col->dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (mask && mask->size != col->n_rows) return KNOX_E_INVALID;
    if (func < 0 || func > 8) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, col); if (rc_) return rc_; }
    int rc = ensure(ctx, ctx->misc, 256);
    if (rc) return rc;
    uint64_t *d = (uint64_t *)((char *)ctx->misc.p + 192);
    hipError_t e = launch_reduce_sequential(ctx->stream, col->dtype, func, col->dev, col->n_rows, mask ? (const uint8_t *)mask->dev : nullptr, d);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    uint64_t h[3] = {0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(value_out, &h[0], (size_t)col->width);
    if (seen) *seen = (int)h[1];
    if (row) *row = h[2] == ~0ull ? -1 : (int64_t)h[2];
    return KNOX_OK;
}

// ---- NativeBucket.Emit's gap filling over dense per-window values (pure host code) --------------
// bucket_native.go:161-218 walks the windows from trange.From; a window without a reducer is filled by
// Fill(mode, step, last.Time(), next.Time(), lastVal, nextVal) (fill.go:59-85), where `last` starts as the
// FIRST reducer and `next` is the next reducer or, past the last one, `last` again -- quirks kept: a gap before
// the first window is "filled" from the first reducer, linear fill outside the reducers divides 0 by 0.
template <typename T>
static size_t fill_windows(int mode, const T *vals, const uint8_t *seen, size_t n, int64_t t_from, int64_t step, int64_t ups, size_t limit,
                           T *out, uint8_t *flags) {
    memset(flags, 0, n);
    size_t first = 0;
    while (first < n && !seen[first]) first++;
    if (first == n) return 0;  // no reducers: Emit writes nothing
    auto unix_s = [&](size_t k) {  // time.Time.Unix(): seconds, floored
        const int64_t t = t_from + (int64_t)k * step;
        int64_t q = t / ups;
        if ((t % ups != 0) && ((t < 0) != (ups < 0))) q--;
        return q;
    };
    size_t last = first, next_seen = first, count = 0;
    for (size_t k = 0; k < n && count < limit; k++) {
        while (next_seen < n && (next_seen < k || !seen[next_seen])) next_seen++;  // reducers[idx]
        const size_t next = next_seen < n ? next_seen : last;
        if (next != k) {
            switch (mode) {
            case KNOX_FILL_NULL: out[k] = 0; flags[k] = 2; count++; break;
            case KNOX_FILL_LAST: out[k] = vals[last]; flags[k] = 1; count++; break;
            case KNOX_FILL_ZERO: out[k] = 0; flags[k] = 1; count++; break;
            case KNOX_FILL_LINEAR: {
                // linearFill (fill.go:56-61): the slope between (prevTime, prevVal) and (nextTime, nextVal), in seconds
                const double m = (double)(T)(vals[next] - vals[last]) / (double)(unix_s(next) - unix_s(last));
                const double x = (double)(unix_s(k) - unix_s(last));
                const double b = (double)vals[last];
                out[k] = (T)(m * x + b);
                flags[k] = 1;
                count++;
                break;
            }
            default: break;  // none
            }
            continue;
        }
        out[k] = vals[k];
        flags[k] = 1;
        count++;
        last = k;
    }
    return count;
}

KNOX_API int knox_fill_windows(int dtype, int mode, const void *values, const uint8_t *seen, size_t n_buckets, int64_t t_from, int64_t step,
                               int64_t units_per_second, size_t limit, void *out_values, uint8_t *out_flags, size_t *n_out) {
    if (!values || !seen || !out_values || !out_flags || step <= 0 || units_per_second <= 0) return KNOX_E_INVALID;
    if (mode < KNOX_FILL_NONE || mode > KNOX_FILL_ZERO) return KNOX_E_INVALID;
    if ((__int128)t_from + (__int128)step * (__int128)n_buckets > (__int128)INT64_MAX) return KNOX_E_RANGE;
    size_t c;
#define KNOX_FILL(T) c = fill_windows<T>(mode, (const T *)values, seen, n_buckets, t_from, step, units_per_second, limit, (T *)out_values, out_flags); break
    switch (dtype) {
    case KNOX_INT64: KNOX_FILL(int64_t);
    case KNOX_INT32: KNOX_FILL(int32_t);
    case KNOX_INT16: KNOX_FILL(int16_t);
    case KNOX_INT8: KNOX_FILL(int8_t);
    case KNOX_UINT64: KNOX_FILL(uint64_t);
    case KNOX_UINT32: KNOX_FILL(uint32_t);
    case KNOX_UINT16: KNOX_FILL(uint16_t);
    case KNOX_UINT8: KNOX_FILL(uint8_t);
    case KNOX_FLOAT64: KNOX_FILL(double);
    case KNOX_FLOAT32: KNOX_FILL(float);
    default: return KNOX_E_INVALID;
    }
#undef KNOX_FILL
    if (n_out) *n_out = c;
    return KNOX_OK;
}

