// knox_agg.hip -- everything around the streaming scan that produces or merges aggregate state: masked reduce over an
// existing selection, the tail / small-n kernel, the finalisers (ordered merge of workgroup partials, per-pack counts),
// the second pass of var / std, and the literal one-lane reducers (parity mode).
#include "knox_dev.h"

namespace knox {

int tile_rows(int width) { return 64 * (16 / width) * kUnroll * kWavesPerBlock; }

// ---------------------------------------------------------------------------------------------
// masked reduce kernel: selection comes from an existing bitset (or all rows)
// ---------------------------------------------------------------------------------------------
template <int W, unsigned AGG, unsigned FLV>
__global__ void __launch_bounds__(kBlock) reduce_kernel(const ScanArgs a) {
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;
    constexpr int WAVE_ROWS = RPL * kUnroll;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Acc acc;
    acc.init();
    WaveMM wm;
    wm.init();
    FirstLast fl;
    fl.init();
    uint32_t cnt32 = 0;
    const uint32_t iters = a.iters;
    for (uint64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint64_t wrow0 = (tile * kWavesPerBlock + (uint64_t)wave) * iters * WAVE_ROWS;
#pragma unroll 1
        for (uint32_t it = 0; it < iters; it++) {
        const uint64_t row0 = wrow0 + (uint64_t)it * WAVE_ROWS;
        const uint64_t vec0 = row0 / K + lane;
        u32x4 ra[kUnroll];
        uint32_t q[kUnroll];
        const u32x4 *p = (const u32x4 *)a.agg_col + vec0;
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
        }  // it
    }
    acc.count = cnt32;
    if constexpr ((AGG & A_FIRSTLAST) != 0) fl.template finish<K>(acc, a.row_base, lane);
    block_reduce_store(acc, a.partials + blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// tail / small-n kernel: rows [n_tiles*TILE, n_rows), any n, byte-granular stores so nothing is
// written past byte (n+7)/8 - 1 and the bits >= n of the last byte stay 0 (poison-tail rule,
// internal/cmp/tests/tests.go:24-48).  One workgroup; thread t owns output bytes t, t+256, ...
// ---------------------------------------------------------------------------------------------

template <int W>
__global__ void __launch_bounds__(kBlock) scan_tail_kernel(const ScanArgs a, int n_leaves, int tail_slot, int reduce_mode) {
    using U = typename WT<W>::U;
    const uint64_t start = a.tail_first_row;
    Acc acc;
    acc.init();
    if (start < a.n_rows) {
        const uint64_t nrows = a.n_rows - start;
        const uint64_t nbytes = (nrows + 7) >> 3;
        const uint8_t *mask8 = (const uint8_t *)a.mask;
        for (uint64_t b = threadIdx.x; b < nbytes; b += blockDim.x) {
            uint32_t out = 0;
            uint32_t mbyte = 0xffu;
            if (reduce_mode && mask8) mbyte = mask8[(start >> 3) + b];
#pragma unroll 1
            for (int k = 0; k < 8; k++) {
                const uint64_t row = start + 8 * b + k;
                if (row >= a.n_rows) break;
                uint32_t hit;
                if (reduce_mode) {
                    hit = (mbyte >> k) & 1u;
                } else {
                    uint32_t idx = 0;
                    bool skipped = false;
                    if (a.pack_skip != nullptr && a.pack_shift != 255u) skipped = a.pack_skip[row >> a.pack_shift] != 0;
                    for (int l = 0; l < n_leaves; l++) {
                        U x = load_elem<W>(a.leaf[l].col, row);
                        if constexpr (W >= 4) x = a.leaf[l].is_float ? fkey<W>(x) : x;
                        const uint32_t h = (((U)(x - (U)a.leaf[l].lo)) & WT<W>::MASK) <= (U)a.leaf[l].diff ? 1u : 0u;
                        idx |= h << l;
                    }
                    hit = skipped ? 0u : (uint32_t)((a.tt >> idx) & 1ull);
                }
                out |= hit << k;
                acc.count += hit;
                if (a.agg_col != nullptr) acc_update<W, A_ALL>(acc, load_elem<W>(a.agg_col, row), hit, a.row_base + row, a.agg_flavor);
            }
            if (a.bits != nullptr && !reduce_mode) ((uint8_t *)a.bits)[(start >> 3) + b] = (uint8_t)out;
        }
    }
    block_reduce_store(acc, a.partials + tail_slot);
}

// ---------------------------------------------------------------------------------------------
// finalize: ordered merge of the workgroup partials, fetch adopted values
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void finalize_body(const AggPartial *partials, int n_partials, const void *agg_col, uint64_t row_base, int dtype,
                                              AggStateDev *state_out, int64_t *count_out) {
    __shared__ Acc sm[kBlock / 64];
    Acc acc;
    acc.init();
    for (int i = threadIdx.x; i < n_partials; i += blockDim.x) {
        const AggPartial p = partials[i];
        Acc b;
        b.count = p.count; b.isum = p.isum; b.hi = p.hi; b.lo = p.lo; b.xlo = p.xlo; b.xhi = p.xhi;
        b.minkey = p.minkey; b.minrow = p.minrow; b.maxkey = p.maxkey; b.maxrow = p.maxrow;
        b.firstrow = p.firstrow; b.lastrow1 = p.lastrow;
        acc.merge(b);
    }
    wave_reduce(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Acc t = sm[0];
        for (int w = 1; w < kBlock / 64; w++) t.merge(sm[w]);
        if (count_out) *count_out = (int64_t)t.count;
        if (state_out) {
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
            s.min_bits = (agg_col && s.min_row >= 0) ? (uint64_t)load_elem<W>(agg_col, t.minrow - row_base) : 0;
            s.max_bits = (agg_col && s.max_row >= 0) ? (uint64_t)load_elem<W>(agg_col, t.maxrow - row_base) : 0;
            s.first_bits = (agg_col && s.first_row >= 0) ? (uint64_t)load_elem<W>(agg_col, t.firstrow - row_base) : 0;
            s.last_bits = (agg_col && s.last_row >= 0) ? (uint64_t)load_elem<W>(agg_col, t.lastrow1 - 1 - row_base) : 0;
            s.m_shift = 0.0; s.m_dev_hi = 0.0; s.m_dev_lo = 0.0; s.m_sq_hi = 0.0; s.m_sq_lo = 0.0;
            *state_out = s;
        }
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) finalize_kernel(const AggPartial *partials, int n_partials, const void *agg_col,
                                                          uint64_t row_base, int dtype, AggStateDev *state_out,
                                                          int64_t *count_out) {
    finalize_body<W>(partials, n_partials, agg_col, row_base, dtype, state_out, count_out);
}

// one wave per pack: the pack's count from the scan kernel's per-wave-tile counts ([workgroup][wave][round], tile t was
// round t / grid of workgroup t % grid)
__device__ __forceinline__ void pack_counts_body(uint32_t block, const uint32_t *tile_counts, uint32_t grid, uint32_t rounds, uint32_t tiles_per_pack,
                                                 uint64_t n_packs, int64_t *counts) {
    const int lane = threadIdx.x & 63;
    const uint64_t p = (uint64_t)block * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= n_packs) return;
    uint64_t c = 0;
    for (uint32_t i = lane; i < tiles_per_pack * kWavesPerBlock; i += 64) {
        const uint64_t tile = p * tiles_per_pack + i / kWavesPerBlock;
        const uint32_t wave = i % kWavesPerBlock;
        c += tile_counts[((tile % grid) * kWavesPerBlock + wave) * (uint64_t)rounds + tile / grid];
    }
    for (int off = 32; off >= 1; off >>= 1) c += (uint64_t)__shfl_down((unsigned long long)c, off, 64);
    if (lane == 0) counts[p] = (int64_t)c;
}

// A scan's whole follow-up work as ONE launch: workgroup 0 merges the workgroup partials into the aggregate state, the
// others turn the per-tile counts into per-pack counts.  (What a step costs beyond the scan kernel is the dispatch gap
// per dependent launch, ~6.5 us each, not the duration of these small kernels.)
template <int W>
__global__ void __launch_bounds__(kBlock) finalize_packs_kernel(const AggPartial *partials, int n_partials, const void *agg_col, uint64_t row_base,
                                                                int dtype, AggStateDev *state_out, int64_t *count_out,
                                                                const uint32_t *tile_counts, uint32_t grid, uint32_t rounds,
                                                                uint32_t tiles_per_pack, uint64_t n_packs, int64_t *pack_counts) {
    if (blockIdx.x == 0) finalize_body<W>(partials, n_partials, agg_col, row_base, dtype, state_out, count_out);
    else pack_counts_body(blockIdx.x - 1, tile_counts, grid, rounds, tiles_per_pack, n_packs, pack_counts);
}

// per-pack counts from the scan kernel's per-wave-tile counts ([workgroup][wave][round], tile t was
// round t / grid of workgroup t % grid): one wave per pack
__global__ void __launch_bounds__(kBlock) pack_counts_kernel(const uint32_t *tile_counts, uint32_t grid, uint32_t rounds, uint32_t tiles_per_pack,
                                                             uint64_t n_packs, int64_t *counts) {
    pack_counts_body(blockIdx.x, tile_counts, grid, rounds, tiles_per_pack, n_packs, counts);
}

hipError_t launch_pack_counts(hipStream_t s, const uint32_t *tile_counts, uint32_t grid, uint32_t rounds, uint32_t tiles_per_pack,
                              uint64_t n_packs, int64_t *counts) {
    if (n_packs == 0) return hipSuccess;
    KNOX_LAUNCH(pack_counts_kernel, (int)((n_packs + kBlock / 64 - 1) / (kBlock / 64)), tile_counts, grid, rounds, tiles_per_pack, n_packs, counts);
}

std::atomic<int> g_scan_occupancy_mismatches{0};

// the scan kernel's instantiations live in knox_scan.hip, one object per width
template <int W> hipError_t scan_nl(hipStream_t s, int nl, unsigned agg, bool sep, int grid, const ScanArgs &a);

hipError_t launch_scan(hipStream_t s, int width, int n_leaves, unsigned agg, bool agg_separate, int grid,
                       const ScanArgs &a) {
    if (grid <= 0) return hipSuccess;
    switch (width) {
    case 8: return scan_nl<8>(s, n_leaves, agg, agg_separate, grid, a);
    case 4: return scan_nl<4>(s, n_leaves, agg, agg_separate, grid, a);
    case 2: return scan_nl<2>(s, n_leaves, agg, agg_separate, grid, a);
    case 1: return scan_nl<1>(s, n_leaves, agg, agg_separate, grid, a);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_tail(hipStream_t s, int width, int n_leaves, unsigned agg, int tail_slot, const ScanArgs &a) {
    (void)agg;
    switch (width) {
    case 8: KNOX_LAUNCH((scan_tail_kernel<8>), 1, a, n_leaves, tail_slot, 0);
    case 4: KNOX_LAUNCH((scan_tail_kernel<4>), 1, a, n_leaves, tail_slot, 0);
    case 2: KNOX_LAUNCH((scan_tail_kernel<2>), 1, a, n_leaves, tail_slot, 0);
    case 1: KNOX_LAUNCH((scan_tail_kernel<1>), 1, a, n_leaves, tail_slot, 0);
    default: return hipErrorInvalidValue;
    }
}

template <int W> static hipError_t reduce_w(hipStream_t s, unsigned agg, int grid, const ScanArgs &a) {
    constexpr unsigned FF = A_FSUM | A_MINMAX | A_FIRSTLAST;
    hipError_t e = hipSuccess;
    if (grid > 0) {
        bool done = false;
        if constexpr (W >= 4) {
            if (a.agg_flavor == FL_FLOAT) {
                hipLaunchKernelGGL((reduce_kernel<W, FF, FL_FLOAT>), dim3(grid), dim3(kBlock), 0, s, a);
                done = true;
            }
        }
        if (!done) {
            if ((agg & ~A_ISUM) == 0) hipLaunchKernelGGL((reduce_kernel<W, A_ISUM, FL_SIGNED>), dim3(grid), dim3(kBlock), 0, s, a);
            else if (a.agg_flavor == FL_SIGNED) hipLaunchKernelGGL((reduce_kernel<W, A_ALL, FL_SIGNED>), dim3(grid), dim3(kBlock), 0, s, a);
            else hipLaunchKernelGGL((reduce_kernel<W, A_ALL, FL_UNSIGNED>), dim3(grid), dim3(kBlock), 0, s, a);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((scan_tail_kernel<W>), dim3(1), dim3(kBlock), 0, s, a, 0, grid, 1);
    return hipGetLastError();
}

hipError_t launch_reduce(hipStream_t s, int width, unsigned agg, int grid, const ScanArgs &a) {
    switch (width) {
    case 8: return reduce_w<8>(s, agg, grid, a);
    case 4: return reduce_w<4>(s, agg, grid, a);
    case 2: return reduce_w<2>(s, agg, grid, a);
    case 1: return reduce_w<1>(s, agg, grid, a);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize_packs(hipStream_t s, int width, const AggPartial *partials, int n_partials, const void *agg_col, uint64_t row_base,
                                 int dtype, AggStateDev *state_out, int64_t *count_out, const uint32_t *tile_counts, uint32_t grid,
                                 uint32_t rounds, uint32_t tiles_per_pack, uint64_t n_packs, int64_t *pack_counts) {
    const int g = 1 + (int)((n_packs + kBlock / 64 - 1) / (kBlock / 64));
    switch (width) {
    case 8: KNOX_LAUNCH((finalize_packs_kernel<8>), g, partials, n_partials, agg_col, row_base, dtype, state_out, count_out, tile_counts, grid, rounds, tiles_per_pack, n_packs, pack_counts);
    case 4: KNOX_LAUNCH((finalize_packs_kernel<4>), g, partials, n_partials, agg_col, row_base, dtype, state_out, count_out, tile_counts, grid, rounds, tiles_per_pack, n_packs, pack_counts);
    case 2: KNOX_LAUNCH((finalize_packs_kernel<2>), g, partials, n_partials, agg_col, row_base, dtype, state_out, count_out, tile_counts, grid, rounds, tiles_per_pack, n_packs, pack_counts);
    case 1: KNOX_LAUNCH((finalize_packs_kernel<1>), g, partials, n_partials, agg_col, row_base, dtype, state_out, count_out, tile_counts, grid, rounds, tiles_per_pack, n_packs, pack_counts);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize(hipStream_t s, int width, const AggPartial *partials, int n_partials, const void *agg_col,
                           uint64_t row_base, int dtype, AggStateDev *state_out, int64_t *count_out) {
    switch (width) {
    case 8: KNOX_LAUNCH((finalize_kernel<8>), 1, partials, n_partials, agg_col, row_base, dtype, state_out, count_out);
    case 4: KNOX_LAUNCH((finalize_kernel<4>), 1, partials, n_partials, agg_col, row_base, dtype, state_out, count_out);
    case 2: KNOX_LAUNCH((finalize_kernel<2>), 1, partials, n_partials, agg_col, row_base, dtype, state_out, count_out);
    case 1: KNOX_LAUNCH((finalize_kernel<1>), 1, partials, n_partials, agg_col, row_base, dtype, state_out, count_out);
    default: return hipErrorInvalidValue;
    }
}


template <int W, unsigned FLV>
__global__ void __launch_bounds__(kBlock) moments_kernel(const void *col, const uint32_t *mask, uint64_t n_tiles, uint64_t n_rows,
                                                         const AggStateDev *st, double shift, double *partials) {
    using U = typename WT<W>::U;
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;
    constexpr int WAVE_ROWS = RPL * kUnroll;
    __shared__ Mom sm[kBlock / 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const double k0 = moments_shift(st, shift);
    Mom m;
    m.init();
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t row0 = (tile * kWavesPerBlock + (uint64_t)wave) * WAVE_ROWS;
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
    // ragged tail: rows the full tiles do not cover, by workgroup 0
    if (blockIdx.x == 0) {
        const uint8_t *mask8 = (const uint8_t *)mask;
        for (uint64_t row = n_tiles * (uint64_t)(kWavesPerBlock * WAVE_ROWS) + threadIdx.x; row < n_rows; row += kBlock) {
            const bool hit = mask8 ? ((mask8[row >> 3] >> (row & 7)) & 1u) != 0 : true;
            m.add(to_f64_c<W, FLV>((U)load_elem<W>(col, row)), k0, hit);
        }
    }
    mom_block_reduce(m, sm);
    if (threadIdx.x == 0) {
        double *o = partials + 4 * (size_t)blockIdx.x;
        o[0] = m.qh; o[1] = m.ql; o[2] = m.dh; o[3] = m.dl;
    }
}

__global__ void __launch_bounds__(kBlock) moments_finalize_kernel(const double *partials, int n_partials, const AggStateDev *st, double shift,
                                                                  double *out5, uint32_t *flags_out) {
    __shared__ Mom sm[kBlock / 64];
    Mom m;
    m.init();
    for (int i = threadIdx.x; i < n_partials; i += blockDim.x) {
        Mom o;
        o.qh = partials[4 * i]; o.ql = partials[4 * i + 1]; o.dh = partials[4 * i + 2]; o.dl = partials[4 * i + 3];
        m.merge(o);
    }
    mom_block_reduce(m, sm);
    if (threadIdx.x == 0) {
        out5[0] = moments_shift(st, shift);
        out5[1] = m.dh; out5[2] = m.dl; out5[3] = m.qh; out5[4] = m.ql;
        if (flags_out) *flags_out |= 2u;
    }
}

template <int W>
static hipError_t moments_w(hipStream_t s, uint32_t flavor, int grid, const void *col, const uint32_t *mask, uint64_t n_tiles,
                            uint64_t n_rows, const AggStateDev *st, double shift, double *partials) {
    if constexpr (W >= 4) {
        if (flavor == FL_FLOAT) KNOX_LAUNCH((moments_kernel<W, FL_FLOAT>), grid, col, mask, n_tiles, n_rows, st, shift, partials);
    }
    if (flavor == FL_SIGNED) KNOX_LAUNCH((moments_kernel<W, FL_SIGNED>), grid, col, mask, n_tiles, n_rows, st, shift, partials);
    KNOX_LAUNCH((moments_kernel<W, FL_UNSIGNED>), grid, col, mask, n_tiles, n_rows, st, shift, partials);
}

hipError_t launch_moments(hipStream_t s, int width, uint32_t flavor, int grid, const void *col, const uint32_t *mask, uint64_t n_rows,
                          const AggStateDev *state_in, double shift, double *partials, double *out5, uint32_t *flags_out) {
    const uint64_t n_tiles = n_rows / (uint64_t)tile_rows(width);
    grid = (int)(n_tiles < (uint64_t)grid ? (n_tiles ? n_tiles : 1) : (uint64_t)grid);
    hipError_t e;
    switch (width) {
    case 8: e = moments_w<8>(s, flavor, grid, col, mask, n_tiles, n_rows, state_in, shift, partials); break;
    case 4: e = moments_w<4>(s, flavor, grid, col, mask, n_tiles, n_rows, state_in, shift, partials); break;
    case 2: e = moments_w<2>(s, flavor, grid, col, mask, n_tiles, n_rows, state_in, shift, partials); break;
    case 1: e = moments_w<1>(s, flavor, grid, col, mask, n_tiles, n_rows, state_in, shift, partials); break;
    default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)partials, grid, state_in, shift, out5, flags_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// literal sequential reducers (one lane, row order): internal/reducer/reducer.go:138-429
// bit-for-bit the Go fold incl. Welford mean/var -- parity-test mode only, O(n) latency bound.
// out[0] = value bits (T in the low bytes), out[1] = seen, out[2] = row (as uint64, ~0 = none)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void reduce_seq_kernel(int func, const T *src, uint64_t n, const uint8_t *bits, uint64_t *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool seen = false;
    uint64_t row = ~0ull, cnt = 0;
    T v = 0;
    double mean = 0, m2 = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (bits != nullptr && !((bits[i >> 3] >> (i & 7)) & 1)) continue;
        const T x = src[i];
        switch (func) {
        case 8: if (!seen) row = i; v = (T)(v + 1); break;                       // count :138-148
        case 0: if (!seen) row = i; v = (T)(v + x); break;                       // sum   :168-178
        case 1: if (!seen) { row = i; v = x; } break;                            // first :198-208
        case 2: row = i; v = x; break;                                           // last  :228-236
        case 4: if (!seen || v < x) { row = i; v = x; } break;                   // max   :256-266
        case 3: if (!seen || v > x) { row = i; v = x; } break;                   // min   :286-296
        case 5: { if (cnt == 0) row = i; cnt++; double d = (double)x - mean; mean += d / (double)cnt; break; }
        case 6: case 7: {
            if (cnt == 0) row = i;
            cnt++;
            double d = (double)x - mean;
            mean += d / (double)cnt;
            m2 += d * ((double)x - mean);
            break;
        }
        default: break;
        }
        seen = true;
    }
    T r = v;
    if (func == 5) r = (T)mean;
    if (func == 6) r = (cnt < 2) ? (T)__builtin_nan("") : (T)(m2 / (double)(cnt - 1));
    if (func == 7) r = (cnt < 2) ? (T)__builtin_nan("") : (T)__builtin_sqrt(m2 / (double)(cnt - 1));
    uint64_t raw = 0;
    __builtin_memcpy(&raw, &r, sizeof(T));
    out[0] = raw;
    out[1] = seen ? 1 : 0;
    out[2] = row;
}

hipError_t launch_reduce_sequential(hipStream_t s, int dtype, int func, const void *col, uint64_t n_rows, const uint8_t *mask,
                                    uint64_t *out3) {
#define KNOX_SEQ(T) hipLaunchKernelGGL((reduce_seq_kernel<T>), dim3(1), dim3(64), 0, s, func, (const T *)col, n_rows, mask, out3); break
    switch (dtype) {
    case 1: KNOX_SEQ(int64_t);
    case 2: KNOX_SEQ(int32_t);
    case 3: KNOX_SEQ(int16_t);
    case 4: KNOX_SEQ(int8_t);
    case 5: KNOX_SEQ(uint64_t);
    case 6: KNOX_SEQ(uint32_t);
    case 7: KNOX_SEQ(uint16_t);
    case 8: KNOX_SEQ(uint8_t);
    case 9: KNOX_SEQ(double);
    case 10: KNOX_SEQ(float);
    default: return hipErrorInvalidValue;
    }
#undef KNOX_SEQ
    return hipGetLastError();
}

}  // namespace knox
