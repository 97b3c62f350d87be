// knox_window.hip -- time-window (NativeBucket) aggregation over ordered time columns: window bounds, work items,
// per-window reduce / finalize / moments, and the literal per-window reducers incl. the join reducers (parity mode).
// The sort that brings out-of-order time into window order is knox_sort.hip.
#include "knox_dev.h"
#include "knox_finalize.h"

namespace knox {

// ---------------------------------------------------------------------------------------------
// Time-window aggregation ("next" row f1): pkg/series/series.go:192-246 -> NativeBucket.Push
// (internal/reducer/bucket_native.go:104-159) keeps one reducer per window [from + k*step,
// from + (k+1)*step).  With a non-decreasing time column every window is ONE contiguous row range:
// the ranges come from a binary search per window edge, are cut into work items, each work item is
// reduced by one workgroup with the same exact machinery as the global aggregates, and the
// partials of a window are merged in row order.  Out-of-order timestamps (which the reference
// handles by inserting reducers, :123-141) are detected and refused: the caller keeps the host path.
// ---------------------------------------------------------------------------------------------
// One streaming pass over the time column, 8 B per row: four independent 16-byte loads per lane and loop iteration (the
// first version had one, and a shuffle between it and the next: 92 % of its wave cycles were waits, 5.5 TB/s; round 4
// profile profiles/r04_windows.txt).  A lane compares the two timestamps of each of its vectors with each other and the
// second one with the first timestamp of the NEXT vector (the neighbour lane's; the wave's last lane reads it).
__global__ void __launch_bounds__(kBlock) sorted_check_kernel(const int64_t *time, uint64_t n_rows, uint32_t *flag) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    bool bad = false;
    const uint64_t npairs = n_rows >> 1;  // 16-byte vectors of two timestamps
    const uint64_t per_iter = (uint64_t)gridDim.x * blockDim.x * U;
    const uint64_t wave_base = ((uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) * U;  // this wave's first vector of the iteration
    for (uint64_t base = wave_base; base < npairs; base += per_iter) {
        // (asked for with the iteration's timestamps, looked at behind them: once ANY wave has seen the clock step back the answer is
        //  known and the rest of the column need not be read -- a shuffled clock costs a few microseconds instead of a whole pass)
        const uint32_t known = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int64_t e0[U], e1[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; u++) {   // vector v = base + u*64 + lane: consecutive lanes, consecutive vectors
            const uint64_t v = base + (uint64_t)u * 64 + lane;
            in[u] = v < npairs;
            e0[u] = INT64_MIN, e1[u] = INT64_MIN;
            if (in[u]) {
                const u32x4 r = ld16<true>((const u32x4 *)time + v);
                e0[u] = (int64_t)(((uint64_t)r.y << 32) | r.x);
                e1[u] = (int64_t)(((uint64_t)r.w << 32) | r.z);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t v = base + (uint64_t)u * 64 + lane;
            int64_t nxt = (int64_t)__shfl_down((unsigned long long)e0[u], 1, 64);  // first timestamp of the next lane's pair
            if (lane == 63 || v + 1 >= npairs) nxt = (2 * v + 2 < n_rows) ? time[2 * v + 2] : INT64_MAX;
            if (in[u]) bad = bad || e0[u] > e1[u] || e1[u] > nxt;
        }
        if (known != 0 || __any(bad)) break;
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1u);
}

hipError_t launch_sorted_check(hipStream_t s, const int64_t *time, uint64_t n_rows, uint32_t *flag) {
    if (n_rows < 2) return hipSuccess;
    uint64_t g = ((n_rows >> 1) + kBlock * 4 - 1) / (kBlock * 4);
    if (g > 2048) g = 2048;
    if (g == 0) g = 1;
    KNOX_LAUNCH(sorted_check_kernel, (int)g, time, n_rows, flag);
}

__global__ void __launch_bounds__(kBlock) bucket_bounds_kernel(const int64_t *time, uint64_t n_rows, int64_t t_from, int64_t step,
                                                               uint64_t n_buckets, uint64_t *row_start) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_buckets) return;
    const int64_t edge = t_from + (int64_t)k * step;  // host checked: no overflow
    // lower_bound(time, edge), started from where a clock that ticks evenly would have the edge: a guess by linear interpolation
    // between the first and the last timestamp, a gallop from it until the edge is bracketed, the binary search inside the bracket
    // -- a few probes instead of log2(n) dependent ones for the clocks time series have (2^20 windows over 2^26 rows: 108 -> 30 us);
    // any non-decreasing clock gives the same answer (an unsorted one some answer: the caller discards it)
    uint64_t lo = 0, hi = n_rows;
    if (n_rows >= 64) {
        const int64_t t0 = time[0], tn = time[n_rows - 1];
        if (edge <= t0) {
            hi = 0;
        } else if (edge > tn) {
            lo = n_rows;
        } else {
            const double f = ((double)edge - (double)t0) / ((double)tn - (double)t0);
            uint64_t g = (uint64_t)(f * (double)(n_rows - 1));
            if (g >= n_rows) g = n_rows - 1;
            if (time[g] < edge) {                 // the answer lies to the right of g
                uint64_t prev = g, b = 1;
                while (g + b < n_rows && time[g + b] < edge) { prev = g + b; b <<= 1; }
                lo = prev + 1;
                hi = g + b < n_rows ? g + b : n_rows;
            } else {                              // at or to the left of g
                uint64_t prev = g, b = 1;
                while (g >= b && time[g - b] >= edge) { prev = g - b; b <<= 1; }
                hi = prev;
                lo = g >= b ? g - b + 1 : 0;
            }
        }
    }
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (time[mid] < edge) lo = mid + 1; else hi = mid;
    }
    row_start[k] = lo;
}

hipError_t launch_bucket_bounds(hipStream_t s, const int64_t *time, uint64_t n_rows, int64_t t_from, int64_t step, uint64_t n_buckets,
                                uint64_t *row_start) {
    KNOX_LAUNCH(bucket_bounds_kernel, (int)((n_buckets + 1 + kBlock - 1) / kBlock), time, n_rows, t_from, step, n_buckets, row_start);
}

template <int W, unsigned AGG, unsigned FLV>
__global__ void __launch_bounds__(kBlock) bucket_reduce_kernel(const ScanArgs a, const BucketItem *items, uint32_t n_items) {
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;
    constexpr uint64_t WAVE_ROWS = (uint64_t)RPL * kUnroll;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint8_t *mask8 = (const uint8_t *)a.mask;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint64_t rb = items[it].begin, re = items[it].end;
        Acc acc;
        acc.init();
        WaveMM wm;
        wm.init();
        FirstLast fl;
        fl.init();
        uint32_t cnt32 = 0;
        const uint64_t a0 = (rb + WAVE_ROWS - 1) / WAVE_ROWS, a1 = re / WAVE_ROWS;  // tile indices [a0, a1) of the aligned middle
        // ragged edges (or the whole item when it holds no aligned tile): one row per lane, exact per-row update.  A lane must meet its
        // rows in ASCENDING order -- min / max adopt a value only when it is strictly better, which is what makes the first occurrence
        // win --, so the head edge goes BEFORE the middle (it came after it until late in round 4: a window whose smallest value stood
        // in its first, unaligned rows and again further down reported the later row; found by the sort-free states agreeing with numpy
        // and not with this kernel)
        const uint64_t e0 = a0 < a1 ? a0 * WAVE_ROWS : re;  // head edge [rb, e0)
        const uint64_t e1 = a0 < a1 ? a1 * WAVE_ROWS : re;  // tail edge [e1, re)
        auto edge = [&](uint64_t lo, uint64_t hi) {
            for (uint64_t row = lo + threadIdx.x; row < hi; row += blockDim.x) {
                const uint32_t hit = mask8 ? (uint32_t)((mask8[row >> 3] >> (row & 7)) & 1u) : 1u;
                acc.count += hit;
                acc_update<W, (AGG | A_FIRSTLAST)>(acc, load_elem<W>(a.agg_col, row), hit, a.row_base + row, FLV);
            }
        };
        edge(rb, e0);
        // aligned middle: whole wave tiles, vector path (its first iteration takes the exact min / max path and reads the wave's bounds
        // from the accumulators the head edge left)
        for (uint64_t t = a0 + wave; t < a1; t += kWavesPerBlock) {
            const uint64_t row0 = t * WAVE_ROWS;
            const u32x4 *p = (const u32x4 *)a.agg_col + row0 / K + lane;
            u32x4 ra[kUnroll];
            uint32_t q[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                ra[u] = ld16<true>(p + u * 64);
                q[u] = a.mask ? a.mask[((row0 + (uint64_t)u * RPL) >> 5) + (lane / G)] : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                q[u] = (q[u] >> (K * (lane & (G - 1)))) & ((1u << K) - 1u);
                cnt32 += __builtin_popcount(q[u]);
            }
            agg_iter<W, AGG, FLV>(acc, wm, fl, ra, q, row0, a.row_base, lane);
        }
        edge(e1, re);
        acc.count += cnt32;
        if constexpr ((AGG & A_FIRSTLAST) != 0) fl.template finish<K>(acc, a.row_base, lane);
        block_reduce_store(acc, a.partials + it);
        __syncthreads();  // block_reduce_store's LDS is reused by the next item
    }
}

// many small windows: one WAVE per work item -- no workgroup barrier and no LDS round per item.  Round 4: the rows go
// through the same vector path as the large items (agg_iter: 16-byte loads, the wave-uniform min / max shortcut, one
// first / last update per iteration) instead of one exact per-row update per lane step (2.2 TB/s at 2^20 windows of 1024
// rows, profiles/r04_windows.txt): the wave walks [rb & ~31, re) in steps of WAVE_ROWS rows -- a start on a multiple of
// 32 rows keeps the 16-byte loads and the mask words lane aligned -- and rows outside [rb, re) are taken out of the
// selection bits (a wave-uniform test skips that arithmetic for interior steps); lanes whose rows lie at or beyond `re`
// do not load.
template <int W, unsigned AGG, unsigned FLV>
__global__ void __launch_bounds__(kBlock) bucket_reduce_wave_kernel(const ScanArgs a, const BucketItem *items, uint32_t n_items) {
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;
    constexpr uint64_t WAVE_ROWS = (uint64_t)RPL * kUnroll;
    constexpr uint32_t FULL = K == 32 ? 0xffffffffu : (1u << K) - 1u;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = gridDim.x * kWavesPerBlock;
    for (uint32_t it = wave; it < n_items; it += n_waves) {
        const uint64_t rb = items[it].begin, re = items[it].end;
        Acc acc;
        acc.init();
        WaveMM wm;
        wm.init();
        FirstLast fl;
        fl.init();
        uint32_t cnt32 = 0;
        for (uint64_t row0 = rb & ~(uint64_t)31; row0 < re; row0 += WAVE_ROWS) {
            u32x4 ra[kUnroll];
            uint32_t q[kUnroll];
            const bool edge = row0 < rb || row0 + WAVE_ROWS > re;  // wave uniform
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t r = row0 + (uint64_t)u * RPL + (uint64_t)lane * K;  // this lane's K rows of load u
                ra[u] = u32x4{0u, 0u, 0u, 0u};
                q[u] = 0u;
                if (r < re) {
                    ra[u] = ld16<true>((const u32x4 *)a.agg_col + r / K);
                    const uint32_t w = a.mask ? a.mask[r >> 5] : 0xffffffffu;
                    q[u] = (w >> (K * (lane & (G - 1)))) & FULL;
                }
            }
            if (edge) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) {
                    const uint64_t r = row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
                    const uint64_t below = rb > r ? rb - r : 0;            // rows of this lane before the window
                    const uint64_t left = re > r ? re - r : 0;             // rows of this lane before the window's end
                    const uint32_t keep_hi = left >= (uint64_t)K ? FULL : (1u << (uint32_t)left) - 1u;
                    const uint32_t drop_lo = below >= (uint64_t)K ? FULL : (1u << (uint32_t)below) - 1u;
                    q[u] &= keep_hi & ~drop_lo;
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) cnt32 += __builtin_popcount(q[u]);
            agg_iter<W, AGG, FLV>(acc, wm, fl, ra, q, row0, a.row_base, lane);
        }
        acc.count += cnt32;
        if constexpr ((AGG & A_FIRSTLAST) != 0) fl.template finish<K>(acc, a.row_base, lane);
        wave_reduce(acc);
        if (lane == 0) {
            AggPartial p;
            p.count = acc.count; p.isum = acc.isum; p.hi = acc.hi; p.lo = acc.lo; p.xlo = acc.xlo; p.xhi = acc.xhi;
            p.minkey = acc.minkey; p.minrow = acc.minrow; p.maxkey = acc.maxkey; p.maxrow = acc.maxrow;
            p.firstrow = acc.firstrow; p.lastrow = acc.lastrow1;
            a.partials[it] = p;
        }
    }
}

// Very many very small windows (64 ... a few thousand rows: 2^20 windows of a 2^26 ... 2^30-row column): a wave per window spends
// its time in the wave-wide reduction of the twelve accumulator fields (six shuffle steps through the LDS crossbar, ~1000
// instructions per window; 1.6 TB/s at 1024 rows per window, latency bound below that -- profiles/r04_windows.txt).  Here a
// window belongs to a GROUP OF 16 LANES, four windows per wave: every lane walks rows rb + sub, + 16, ... in ascending order
// with the exact per-row update (four rows loaded ahead), and the group's accumulators meet in four DPP steps inside the
// 16-lane row (lane ^ 1, lane ^ 2, mirror of 8, mirror of 16: no LDS, no out-of-range lane).  Acc::merge breaks ties by row id, so
// the butterfly's order does not show in the result.
template <int CTRL> __device__ __forceinline__ uint64_t dpp_mov64(uint64_t v) {
    return ((uint64_t)dpp_mov<CTRL>((uint32_t)(v >> 32)) << 32) | dpp_mov<CTRL>((uint32_t)v);
}
template <int CTRL> __device__ __forceinline__ void acc_merge_dpp(Acc &a) {
    Acc b;
    b.count = dpp_mov64<CTRL>(a.count);
    b.isum = dpp_mov64<CTRL>(a.isum);
    b.hi = __longlong_as_double((long long)dpp_mov64<CTRL>((uint64_t)__double_as_longlong(a.hi)));
    b.lo = __longlong_as_double((long long)dpp_mov64<CTRL>((uint64_t)__double_as_longlong(a.lo)));
    b.xlo = dpp_mov64<CTRL>(a.xlo);
    b.xhi = (int64_t)dpp_mov64<CTRL>((uint64_t)a.xhi);
    b.minkey = dpp_mov64<CTRL>(a.minkey);
    b.minrow = dpp_mov64<CTRL>(a.minrow);
    b.maxkey = dpp_mov64<CTRL>(a.maxkey);
    b.maxrow = dpp_mov64<CTRL>(a.maxrow);
    b.firstrow = dpp_mov64<CTRL>(a.firstrow);
    b.lastrow1 = dpp_mov64<CTRL>(a.lastrow1);
    a.merge(b);
}

template <int W, unsigned AGG, unsigned FLV>
__global__ void __launch_bounds__(kBlock) bucket_reduce_group_kernel(const ScanArgs a, const BucketItem *items, uint32_t n_items) {
    using U = typename WT<W>::U;
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = gridDim.x * kWavesPerBlock;
    const uint8_t *mask8 = (const uint8_t *)a.mask;
    const uint64_t *mask64 = (const uint64_t *)a.mask;
    const uint64_t last64 = ((a.n_rows + 63) >> 6) - 1;
    for (uint32_t base = wave * 4u; base < n_items; base += n_waves * 4u) {  // wave uniform: the DPP steps below see all 64 lanes
        const uint32_t it = base + (uint32_t)grp;
        uint64_t rb = 0, re = 0;
        if (it < n_items) { rb = items[it].begin; re = items[it].end; }
        Acc acc;
        acc.init();
        uint64_t row = rb + (uint64_t)sub;
        for (; row + 48 < re; row += 64) {
            U e[4];
            uint32_t h[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] = load_elem<W>(a.agg_col, row + 16u * j);
            // the lane's four selection bits sit 16 apart: bits s, s + 16, s + 32, s + 48 of the 64-bit word that holds row `row`
            // and, for s > 15, its successor -- two aligned 8-byte loads instead of four byte loads (a bitset is padded to whole
            // 64-bit words, knox_internal.h bits_padded)
            uint64_t m = ~0ull;
            if (mask64) {
                const uint64_t i = row >> 6;
                const uint32_t sh = (uint32_t)row & 63u;
                const uint64_t v0 = mask64[i];
                const uint64_t v1 = (sh > 15u && i < last64) ? mask64[i + 1] : 0ull;
                m = sh ? (v0 >> sh) | (v1 << (64u - sh)) : v0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) h[j] = (uint32_t)(m >> (16 * j)) & 1u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc.count += h[j];
                acc_update<W, AGG>(acc, e[j], h[j], a.row_base + row + 16u * j, FLV);
            }
        }
        for (; row < re; row += 16) {
            const uint32_t hit = mask8 ? (uint32_t)((mask8[row >> 3] >> (row & 7)) & 1u) : 1u;
            acc.count += hit;
            acc_update<W, AGG>(acc, load_elem<W>(a.agg_col, row), hit, a.row_base + row, FLV);
        }
        acc_merge_dpp<0xB1>(acc);   // quad_perm [1,0,3,2]
        acc_merge_dpp<0x4E>(acc);   // quad_perm [2,3,0,1]
        acc_merge_dpp<0x141>(acc);  // row_half_mirror
        acc_merge_dpp<0x140>(acc);  // row_mirror
        if (sub == 0 && it < n_items) {
            AggPartial p;
            p.count = acc.count; p.isum = acc.isum; p.hi = acc.hi; p.lo = acc.lo; p.xlo = acc.xlo; p.xhi = acc.xhi;
            p.minkey = acc.minkey; p.minrow = acc.minrow; p.maxkey = acc.maxkey; p.maxrow = acc.maxrow;
            p.firstrow = acc.firstrow; p.lastrow = acc.lastrow1;
            a.partials[it] = p;
        }
    }
}

// sums_only: the caller wants Sum / Count / Mean (and the mean the var / std pass centres on) only -- knox_reduce_buckets_values knows its
// reducer -- so the per-row update keeps two sums and the count; min / max / first / last stay at their initial "none"
template <int W> static hipError_t bucket_reduce_groups_w(hipStream_t s, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items, bool sums_only) {
    constexpr unsigned FF = A_FSUM | A_MINMAX | A_FIRSTLAST;
    if constexpr (W >= 4) {
        if (a.agg_flavor == FL_FLOAT) {
            if (sums_only) KNOX_LAUNCH((bucket_reduce_group_kernel<W, A_FSUM, FL_FLOAT>), grid, a, items, n_items);
            KNOX_LAUNCH((bucket_reduce_group_kernel<W, FF, FL_FLOAT>), grid, a, items, n_items);
        }
    }
    if (a.agg_flavor == FL_SIGNED) {
        if (sums_only) KNOX_LAUNCH((bucket_reduce_group_kernel<W, (A_ISUM | A_FSUM), FL_SIGNED>), grid, a, items, n_items);
        KNOX_LAUNCH((bucket_reduce_group_kernel<W, A_ALL, FL_SIGNED>), grid, a, items, n_items);
    }
    if (sums_only) KNOX_LAUNCH((bucket_reduce_group_kernel<W, (A_ISUM | A_FSUM), FL_UNSIGNED>), grid, a, items, n_items);
    KNOX_LAUNCH((bucket_reduce_group_kernel<W, A_ALL, FL_UNSIGNED>), grid, a, items, n_items);
}

hipError_t launch_bucket_reduce_groups(hipStream_t s, int width, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items, bool sums_only) {
    if (n_items == 0 || grid <= 0) return hipSuccess;
    switch (width) {
    case 8: return bucket_reduce_groups_w<8>(s, grid, a, items, n_items, sums_only);
    case 4: return bucket_reduce_groups_w<4>(s, grid, a, items, n_items, sums_only);
    case 2: return bucket_reduce_groups_w<2>(s, grid, a, items, n_items, sums_only);
    case 1: return bucket_reduce_groups_w<1>(s, grid, a, items, n_items, sums_only);
    default: return hipErrorInvalidValue;
    }
}

template <int W> static hipError_t bucket_reduce_waves_w(hipStream_t s, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items) {
    constexpr unsigned FF = A_FSUM | A_MINMAX | A_FIRSTLAST;
    if constexpr (W >= 4) {
        if (a.agg_flavor == FL_FLOAT) KNOX_LAUNCH((bucket_reduce_wave_kernel<W, FF, FL_FLOAT>), grid, a, items, n_items);
    }
    if (a.agg_flavor == FL_SIGNED) KNOX_LAUNCH((bucket_reduce_wave_kernel<W, A_ALL, FL_SIGNED>), grid, a, items, n_items);
    KNOX_LAUNCH((bucket_reduce_wave_kernel<W, A_ALL, FL_UNSIGNED>), grid, a, items, n_items);
}

hipError_t launch_bucket_reduce_waves(hipStream_t s, int width, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items) {
    if (n_items == 0 || grid <= 0) return hipSuccess;
    switch (width) {
    case 8: return bucket_reduce_waves_w<8>(s, grid, a, items, n_items);
    case 4: return bucket_reduce_waves_w<4>(s, grid, a, items, n_items);
    case 2: return bucket_reduce_waves_w<2>(s, grid, a, items, n_items);
    case 1: return bucket_reduce_waves_w<1>(s, grid, a, items, n_items);
    default: return hipErrorInvalidValue;
    }
}

template <int W> static hipError_t bucket_reduce_w(hipStream_t s, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items) {
    constexpr unsigned FF = A_FSUM | A_MINMAX | A_FIRSTLAST;
    if constexpr (W >= 4) {
        if (a.agg_flavor == FL_FLOAT) KNOX_LAUNCH((bucket_reduce_kernel<W, FF, FL_FLOAT>), grid, a, items, n_items);
    }
    if (a.agg_flavor == FL_SIGNED) KNOX_LAUNCH((bucket_reduce_kernel<W, A_ALL, FL_SIGNED>), grid, a, items, n_items);
    KNOX_LAUNCH((bucket_reduce_kernel<W, A_ALL, FL_UNSIGNED>), grid, a, items, n_items);
}

hipError_t launch_bucket_reduce(hipStream_t s, int width, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items) {
    if (n_items == 0 || grid <= 0) return hipSuccess;
    switch (width) {
    case 8: return bucket_reduce_w<8>(s, grid, a, items, n_items);
    case 4: return bucket_reduce_w<4>(s, grid, a, items, n_items);
    case 2: return bucket_reduce_w<2>(s, grid, a, items, n_items);
    case 1: return bucket_reduce_w<1>(s, grid, a, items, n_items);
    default: return hipErrorInvalidValue;
    }
}

// one thread per window: ordered merge of its work items' partials, fetch adopted values
// values != nullptr: Reducer.Value() of reducer `func` per window goes out as well (knox_finalize.h, the code window_values_kernel
// runs on the stored states) -- and with state_out == nullptr INSTEAD of the 144-byte state: a caller that only wants values
// (knox_reduce_buckets_values without a var / std pass) then skips a 144 B write and a 144 B read per window
template <int W>
__global__ void __launch_bounds__(kBlock) bucket_finalize_kernel(const AggPartial *partials, const uint32_t *item_start, uint32_t n_buckets,
                                                                 const void *agg_col, uint64_t row_base, int dtype, AggStateDev *state_out,
                                                                 int func, void *values, uint8_t *seen) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_buckets) return;
    Acc t;
    t.init();
    for (uint32_t i = item_start[k]; i < item_start[k + 1]; i++) {
        const AggPartial p = partials[i];
        Acc b;
        b.count = p.count; b.isum = p.isum; b.hi = p.hi; b.lo = p.lo; b.xlo = p.xlo; b.xhi = p.xhi;
        b.minkey = p.minkey; b.minrow = p.minrow; b.maxkey = p.maxkey; b.maxrow = p.maxrow;
        b.firstrow = p.firstrow; b.lastrow1 = p.lastrow;
        t.merge(b);
    }
    AggStateDev s;
    s.dtype = dtype;
    s.flags = 1u;
    s.count = (int64_t)t.count;
    s.isum = t.isum;
    s.fsum_hi = t.hi;
    s.fsum_lo = t.lo;
    if (dtype <= 8) int128_to_dd(t.xlo, t.xhi, s.fsum_hi, s.fsum_lo);
    s.min_row = t.minrow == kNoRow ? -1 : (int64_t)t.minrow;
    s.max_row = t.maxrow == kNoRow ? -1 : (int64_t)t.maxrow;
    s.first_row = t.firstrow == kNoRow ? -1 : (int64_t)t.firstrow;
    s.last_row = t.lastrow1 == 0 ? -1 : (int64_t)(t.lastrow1 - 1);
    s.min_bits = s.min_row >= 0 ? (uint64_t)load_elem<W>(agg_col, t.minrow - row_base) : 0;
    s.max_bits = s.max_row >= 0 ? (uint64_t)load_elem<W>(agg_col, t.maxrow - row_base) : 0;
    s.first_bits = s.first_row >= 0 ? (uint64_t)load_elem<W>(agg_col, t.firstrow - row_base) : 0;
    s.last_bits = s.last_row >= 0 ? (uint64_t)load_elem<W>(agg_col, t.lastrow1 - 1 - row_base) : 0;
    s.m_shift = 0.0; s.m_dev_hi = 0.0; s.m_dev_lo = 0.0; s.m_sq_hi = 0.0; s.m_sq_lo = 0.0;
    if (state_out != nullptr) state_out[k] = s;
    if (values != nullptr) {
        uint64_t bits = 0;
        int sn = 0;
        int64_t row = -1;
        const int rc = knox_fin::agg_value_bits(s, func, &bits, &sn, &row);
        // (WT<W>::U is the WORKING type -- 32 bits for 1- and 2-byte columns --, the values go out in the column's storage width)
        if constexpr (W == 8) ((uint64_t *)values)[k] = bits;
        else if constexpr (W == 4) ((uint32_t *)values)[k] = (uint32_t)bits;
        else if constexpr (W == 2) ((uint16_t *)values)[k] = (uint16_t)bits;
        else ((uint8_t *)values)[k] = (uint8_t)bits;
        seen[k] = (rc == 0 && sn) ? 1 : 0;
    }
}

hipError_t launch_bucket_finalize(hipStream_t s, int width, const AggPartial *partials, const uint32_t *item_start, uint32_t n_buckets,
                                  const void *agg_col, uint64_t row_base, int dtype, AggStateDev *state_out, int func, void *values, uint8_t *seen) {
    if (n_buckets == 0) return hipSuccess;
    const int grid = (int)((n_buckets + kBlock - 1) / kBlock);
    switch (width) {
    case 8: KNOX_LAUNCH((bucket_finalize_kernel<8>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out, func, values, seen);
    case 4: KNOX_LAUNCH((bucket_finalize_kernel<4>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out, func, values, seen);
    case 2: KNOX_LAUNCH((bucket_finalize_kernel<2>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out, func, values, seen);
    case 1: KNOX_LAUNCH((bucket_finalize_kernel<1>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out, func, values, seen);
    default: return hipErrorInvalidValue;
    }
}

// var/std per window: the same work items again, each centred on ITS window's mean (already in
// states[bucket]); 4 doubles per work item, merged per window in row order
template <int W, unsigned FLV>
__global__ void __launch_bounds__(kBlock) bucket_moments_kernel(const void *col, const uint32_t *mask, const BucketItem *items, uint32_t n_items,
                                                                const AggStateDev *states, double *partials) {
    using U = typename WT<W>::U;
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;
    constexpr uint64_t WAVE_ROWS = (uint64_t)RPL * kUnroll;
    __shared__ Mom sm[kBlock / 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint8_t *mask8 = (const uint8_t *)mask;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint64_t rb = items[it].begin, re = items[it].end;
        const double k0 = moments_shift(states + items[it].bucket, 0.0);
        Mom m;
        m.init();
        const uint64_t a0 = (rb + WAVE_ROWS - 1) / WAVE_ROWS, a1 = re / WAVE_ROWS;
        for (uint64_t t = a0 + wave; t < a1; t += kWavesPerBlock) {
            const uint64_t row0 = t * WAVE_ROWS;
            const u32x4 *p = (const u32x4 *)col + row0 / K + lane;
            u32x4 r[kUnroll];
            uint32_t q[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                r[u] = ld16<true>(p + u * 64);
                q[u] = mask ? mask[((row0 + (uint64_t)u * RPL) >> 5) + (lane / G)] : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t qq = (q[u] >> (K * (lane & (G - 1)))) & ((1u << K) - 1u);
#pragma unroll
                for (int j = 0; j < K; j++) m.add(to_f64_c<W, FLV>(elem<W>(r[u], j)), k0, ((qq >> j) & 1u) != 0);
            }
        }
        const uint64_t e0 = a0 < a1 ? a0 * WAVE_ROWS : re;
        const uint64_t e1 = a0 < a1 ? a1 * WAVE_ROWS : re;
        for (int part = 0; part < 2; part++) {
            const uint64_t lo = part == 0 ? rb : e1, hi = part == 0 ? e0 : re;
            for (uint64_t row = lo + threadIdx.x; row < hi; row += blockDim.x) {
                const bool hit = mask8 ? ((mask8[row >> 3] >> (row & 7)) & 1u) != 0 : true;
                m.add(to_f64_c<W, FLV>((U)load_elem<W>(col, row)), k0, hit);
            }
        }
        mom_block_reduce(m, sm);
        if (threadIdx.x == 0) {
            double *o = partials + 4 * (size_t)it;
            o[0] = m.qh; o[1] = m.ql; o[2] = m.dh; o[3] = m.dl;
        }
        __syncthreads();  // sm is reused by the next item
    }
}

// the 16-lanes-per-window form of the second pass (see bucket_reduce_group_kernel): the same rows in the same per-lane order,
// the four running double-doubles met in four DPP steps
template <int CTRL> __device__ __forceinline__ void mom_merge_dpp(Mom &m) {
    Mom o;
    o.qh = __longlong_as_double((long long)dpp_mov64<CTRL>((uint64_t)__double_as_longlong(m.qh)));
    o.ql = __longlong_as_double((long long)dpp_mov64<CTRL>((uint64_t)__double_as_longlong(m.ql)));
    o.dh = __longlong_as_double((long long)dpp_mov64<CTRL>((uint64_t)__double_as_longlong(m.dh)));
    o.dl = __longlong_as_double((long long)dpp_mov64<CTRL>((uint64_t)__double_as_longlong(m.dl)));
    m.merge(o);
}

template <int W, unsigned FLV>
__global__ void __launch_bounds__(kBlock) bucket_moments_group_kernel(const void *col, const uint32_t *mask, uint64_t n_rows, const BucketItem *items,
                                                                      uint32_t n_items, const AggStateDev *states, double *partials) {
    using U = typename WT<W>::U;
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = gridDim.x * kWavesPerBlock;
    const uint8_t *mask8 = (const uint8_t *)mask;
    const uint64_t *mask64 = (const uint64_t *)mask;
    const uint64_t last64 = ((n_rows + 63) >> 6) - 1;
    for (uint32_t base = wave * 4u; base < n_items; base += n_waves * 4u) {
        const uint32_t it = base + (uint32_t)grp;
        uint64_t rb = 0, re = 0;
        double k0 = 0.0;
        if (it < n_items) {
            rb = items[it].begin; re = items[it].end;
            k0 = moments_shift(states + items[it].bucket, 0.0);
        }
        Mom m;
        m.init();
        uint64_t row = rb + (uint64_t)sub;
        for (; row + 48 < re; row += 64) {
            U e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] = load_elem<W>(col, row + 16u * j);
            uint64_t mm = ~0ull;
            if (mask64) {
                const uint64_t i = row >> 6;
                const uint32_t sh = (uint32_t)row & 63u;
                const uint64_t v0 = mask64[i];
                const uint64_t v1 = (sh > 15u && i < last64) ? mask64[i + 1] : 0ull;
                mm = sh ? (v0 >> sh) | (v1 << (64u - sh)) : v0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) m.add(to_f64_c<W, FLV>(e[j]), k0, ((mm >> (16 * j)) & 1u) != 0);
        }
        for (; row < re; row += 16) {
            const bool hit = mask8 ? ((mask8[row >> 3] >> (row & 7)) & 1u) != 0 : true;
            m.add(to_f64_c<W, FLV>((U)load_elem<W>(col, row)), k0, hit);
        }
        mom_merge_dpp<0xB1>(m);
        mom_merge_dpp<0x4E>(m);
        mom_merge_dpp<0x141>(m);
        mom_merge_dpp<0x140>(m);
        if (sub == 0 && it < n_items) {
            double *o = partials + 4 * (size_t)it;
            o[0] = m.qh; o[1] = m.ql; o[2] = m.dh; o[3] = m.dl;
        }
    }
}

__global__ void __launch_bounds__(kBlock) bucket_moments_finalize_kernel(const double *partials, const uint32_t *item_start, uint32_t n_buckets,
                                                                         AggStateDev *states) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_buckets) return;
    Mom m;
    m.init();
    for (uint32_t i = item_start[k]; i < item_start[k + 1]; i++) {
        Mom o;
        o.qh = partials[4 * (size_t)i]; o.ql = partials[4 * (size_t)i + 1]; o.dh = partials[4 * (size_t)i + 2]; o.dl = partials[4 * (size_t)i + 3];
        m.merge(o);
    }
    AggStateDev *s = states + k;
    s->m_shift = moments_shift(s, 0.0);
    s->m_dev_hi = m.dh; s->m_dev_lo = m.dl; s->m_sq_hi = m.qh; s->m_sq_lo = m.ql;
    s->flags |= 2u;
}

template <int W>
static hipError_t bucket_moments_w(hipStream_t s, uint32_t flavor, int grid, const void *col, const uint32_t *mask, uint64_t group_rows,
                                   const BucketItem *items, uint32_t n_items, const AggStateDev *states, double *partials) {
    if (group_rows) {  // the 16-lanes-per-window form: grid counts workgroups of 16 windows
        if constexpr (W >= 4) {
            if (flavor == FL_FLOAT) KNOX_LAUNCH((bucket_moments_group_kernel<W, FL_FLOAT>), grid, col, mask, group_rows, items, n_items, states, partials);
        }
        if (flavor == FL_SIGNED) KNOX_LAUNCH((bucket_moments_group_kernel<W, FL_SIGNED>), grid, col, mask, group_rows, items, n_items, states, partials);
        KNOX_LAUNCH((bucket_moments_group_kernel<W, FL_UNSIGNED>), grid, col, mask, group_rows, items, n_items, states, partials);
    }
    if constexpr (W >= 4) {
        if (flavor == FL_FLOAT) KNOX_LAUNCH((bucket_moments_kernel<W, FL_FLOAT>), grid, col, mask, items, n_items, states, partials);
    }
    if (flavor == FL_SIGNED) KNOX_LAUNCH((bucket_moments_kernel<W, FL_SIGNED>), grid, col, mask, items, n_items, states, partials);
    KNOX_LAUNCH((bucket_moments_kernel<W, FL_UNSIGNED>), grid, col, mask, items, n_items, states, partials);
}

hipError_t launch_bucket_moments(hipStream_t s, int width, uint32_t flavor, int grid, const void *col, const uint32_t *mask, uint64_t group_rows,
                                 const BucketItem *items, uint32_t n_items, const uint32_t *item_start, uint32_t n_buckets,
                                 AggStateDev *states, double *partials) {
    if (n_buckets == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (n_items != 0 && grid > 0) {
        switch (width) {
        case 8: e = bucket_moments_w<8>(s, flavor, grid, col, mask, group_rows, items, n_items, states, partials); break;
        case 4: e = bucket_moments_w<4>(s, flavor, grid, col, mask, group_rows, items, n_items, states, partials); break;
        case 2: e = bucket_moments_w<2>(s, flavor, grid, col, mask, group_rows, items, n_items, states, partials); break;
        case 1: e = bucket_moments_w<1>(s, flavor, grid, col, mask, group_rows, items, n_items, states, partials); break;
        default: return hipErrorInvalidValue;
        }
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(bucket_moments_finalize_kernel, dim3((n_buckets + kBlock - 1) / kBlock), dim3(kBlock), 0, s, (const double *)partials,
                       item_start, n_buckets, states);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Literal reducers per time window: one lane owns one window [start[k], start[k+1]) and applies
// the reference's reducer statement by statement in row order -- plain reducers (reducer.go:
// 130-429) and the join reducers (reducer.go:432-731), whose `join` argument is "this row's raw
// timestamp equals the previous streamed row's" (pkg/series/series.go:200-203).  Equal raw
// timestamps always share a window, so the flag is window-local.  Bit-exact with the reference
// for every reducer (Welford included); throughput is one row per lane-step, which is what the
// mode is for: parity and small / many-window series, not the 1B-row scan.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct SeqJoinRed {
    bool seen = false, done = false;
    T v = 0, j = 0;
    uint64_t n = 0;
    double mean = 0, m2 = 0;

    __device__ inline void welford(int func, T x) {
        n++;
        const double d = (double)x - mean;
        mean += d / (double)n;
        if (func != 5 && func != 13) m2 += d * ((double)x - mean);
    }
    __device__ inline void push_join(int func, T x, bool join) {
        if (func == 10) { seen = true; v = join ? (T)(v + x) : x; return; }          // LastJoin :476-483
        if (!seen) { seen = true; if (func == 9) v = x; else j = x; return; }        // r.t.IsZero()
        switch (func) {
        case 9: if (join && !done) v = (T)(v + x); else done = true; break;          // FirstJoin :447-457
        case 12: if (join) j = (T)(j + x); else { if (v < j) v = j; j = x; } break;  // MaxJoin :504-518
        case 11: if (join) j = (T)(j + x); else { if (j < v) v = j; j = x; } break;  // MinJoin :541-555
        default: if (join) j = (T)(j + x); else { welford(func, j); j = x; } break;  // Mean/Var/StdJoin :578-731
        }
    }
    __device__ inline void push(int func, T x) {
        switch (func) {
        case 8: v = (T)(v + 1); break;
        case 0: v = (T)(v + x); break;
        case 1: if (!seen) v = x; break;
        case 2: v = x; break;
        case 4: if (!seen || v < x) v = x; break;
        case 3: if (!seen || v > x) v = x; break;
        default: welford(func, x); break;
        }
        seen = true;
    }
    // Reducer.Value(): the join reducers flush the pending group first
    __device__ inline T value(int func, bool *ok) {
        if (func >= 11 && seen) push_join(func, func == 11 ? j : (T)0, false);
        const bool mom = func == 5 || func == 6 || func == 7 || func >= 13;
        *ok = mom ? n > 0 : seen;
        switch (func) {
        case 5: case 13: return (T)mean;
        case 6: case 14: return n < 2 ? (T)__builtin_nan("") : (T)(m2 / (double)(n - 1));
        case 7: case 15: return n < 2 ? (T)__builtin_nan("") : (T)__builtin_sqrt(m2 / (double)(n - 1));
        default: return v;
        }
    }
};

template <typename T>
__global__ void bucket_seq_kernel(int func, const int64_t *__restrict__ time, const T *__restrict__ src, const uint8_t *__restrict__ bits,
                                  const uint64_t *__restrict__ start, uint32_t n_buckets, T *__restrict__ out_v, uint8_t *__restrict__ out_seen) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_buckets) return;
    SeqJoinRed<T> r;
    bool have_prev = false;
    int64_t prev = 0;
    const uint64_t e = start[k + 1];
    for (uint64_t i = start[k]; i < e; i++) {
        if (bits != nullptr && !((bits[i >> 3] >> (i & 7)) & 1)) continue;
        const T x = src[i];
        if (func >= 9) {
            const int64_t t = time[i];
            r.push_join(func, x, have_prev && t == prev);
            prev = t;
            have_prev = true;
        } else
            r.push(func, x);
    }
    bool ok;
    out_v[k] = r.value(func, &ok);
    out_seen[k] = ok ? 1 : 0;
}

// FEW windows (the 19 GB/s corner of round 3: `Mean` over 1024 windows keeps 1024 lanes = 16 waves of the whole chip
// busy, each lane fetching its own cache line per row): one WAVE per window.  The recurrences are serial per window by
// definition (Welford: mean += (x - mean) / n), so the wave does not split the ARITHMETIC -- every lane runs the same
// recurrence on the same broadcast value, which costs nothing extra on a SIMD machine and keeps the results bit for bit
// those of the one-lane kernel -- it splits the MEMORY side: 64 consecutive rows (value, selection bit, raw timestamp for
// the join reducers) are fetched by the 64 lanes in one coalesced access, the next 64 are in flight while the current ones
// are consumed, and the selected ones are handed to the recurrence in row order through v_readlane (the ballot of the
// selection bits is walked with s_ff1: unselected rows cost nothing).
template <typename T> __device__ __forceinline__ T readlane_any(T v, int j) {
    if constexpr (sizeof(T) == 8) {
        uint64_t u;
        __builtin_memcpy(&u, &v, 8);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), j);
        u = ((uint64_t)hi << 32) | lo;
        T o;
        __builtin_memcpy(&o, &u, 8);
        return o;
    } else {
        uint32_t u = 0;
        __builtin_memcpy(&u, &v, sizeof(T));
        u = (uint32_t)__builtin_amdgcn_readlane((int)u, j);
        T o;
        __builtin_memcpy(&o, &u, sizeof(T));
        return o;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) bucket_seq_wave_kernel(int func, const int64_t *__restrict__ time, const T *__restrict__ src,
                                                                 const uint8_t *__restrict__ bits, const uint64_t *__restrict__ start,
                                                                 uint32_t n_buckets, T *__restrict__ out_v, uint8_t *__restrict__ out_seen) {
    const int lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);  // wave uniform
    if (k >= n_buckets) return;
    SeqJoinRed<T> r;
    bool have_prev = false;
    int64_t prev = 0;
    const uint64_t b = start[k], e = start[k + 1];
    const bool join = func >= 9;
    // the chunk in flight: this lane's row of the NEXT 64
    bool sel_n = false;
    T x_n = 0;
    int64_t t_n = 0;
    auto fetch = [&](uint64_t base) {
        const uint64_t i = base + (uint64_t)lane;
        sel_n = i < e && (bits == nullptr || ((bits[i >> 3] >> (i & 7)) & 1));
        x_n = 0;
        t_n = 0;
        if (sel_n) {
            x_n = src[i];
            if (join) t_n = time[i];
        }
    };
    if (b < e) fetch(b);
    for (uint64_t base = b; base < e; base += 64) {
        const T x = x_n;
        const int64_t t = t_n;
        uint64_t m = __ballot(sel_n);
        if (base + 64 < e) fetch(base + 64);
        while (m) {  // wave uniform: every lane walks the same selected rows in row order
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const T xj = readlane_any<T>(x, j);
            if (join) {
                const int64_t tj = readlane_any<int64_t>(t, j);
                r.push_join(func, xj, have_prev && tj == prev);
                prev = tj;
                have_prev = true;
            } else
                r.push(func, xj);
        }
    }
    bool ok;
    const T v = r.value(func, &ok);
    if (lane == 0) {
        out_v[k] = v;
        out_seen[k] = ok ? 1 : 0;
    }
}

// lanes per window: one (many windows: every lane of the chip has a window of its own) or a whole wave (few windows)
hipError_t launch_bucket_sequential(hipStream_t s, int dtype, int func, const int64_t *time, const void *col, const uint8_t *mask,
                                    const uint64_t *start, uint32_t n_buckets, int wave_per_window, void *out_v, uint8_t *out_seen) {
    const int tpb = 64;
    const int grid = (int)((n_buckets + tpb - 1) / tpb);
    const int wgrid = (int)((n_buckets + kWavesPerBlock - 1) / kWavesPerBlock);
#define KNOX_BSEQ(T)                                                                                                                        \
    if (wave_per_window)                                                                                                                    \
        hipLaunchKernelGGL((bucket_seq_wave_kernel<T>), dim3(wgrid), dim3(kBlock), 0, s, func, time, (const T *)col, mask, start, n_buckets, \
                           (T *)out_v, out_seen);                                                                                           \
    else                                                                                                                                    \
        hipLaunchKernelGGL((bucket_seq_kernel<T>), dim3(grid), dim3(tpb), 0, s, func, time, (const T *)col, mask, start, n_buckets,         \
                           (T *)out_v, out_seen);                                                                                           \
    break
    switch (dtype) {
    case 1: KNOX_BSEQ(int64_t);
    case 2: KNOX_BSEQ(int32_t);
    case 3: KNOX_BSEQ(int16_t);
    case 4: KNOX_BSEQ(int8_t);
    case 5: KNOX_BSEQ(uint64_t);
    case 6: KNOX_BSEQ(uint32_t);
    case 7: KNOX_BSEQ(uint16_t);
    case 8: KNOX_BSEQ(uint8_t);
    case 9: KNOX_BSEQ(double);
    case 10: KNOX_BSEQ(float);
    default: return hipErrorInvalidValue;
    }
#undef KNOX_BSEQ
    return hipGetLastError();
}


}  // namespace knox
