// knox_bitset.hip -- bitset combinators, popcounts, Indexes (bitset -> row ids), Int128 / Int256 compares and the numeric
// IN / NOT IN matcher: kernels that read or write selection bitsets without touching the streaming scan.
#include "knox_dev.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace knox {

// ---------------------------------------------------------------------------------------------
// Int128 / Int256 compares over the stride layout (internal/cmp/int128.go:14-471, int256.go:14-270;
// num.Int128Stride{X0 []int64, X1 []uint64}, Int256Stride{X0..X3}): one 8-byte limb column per
// limb, most significant first.  The order is lexicographic with a signed top limb
// (pkg/num/int128.go:541-555, int256.go:701-713): the host flips the sign bit of a[0] / b[0], the
// kernel flips it on the loaded top limb, and every limb compares unsigned.  A lane owns 2 adjacent
// rows per 16-byte load of every limb column (the W = 8 layout), words are assembled with the same
// DPP butterfly as the scan kernel.  result = hits & mask (bits pre-zeroed per the Matcher
// contract); rows past the last full tile are written byte-granular (poison-tail rule).
// ---------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ bool wide_hit(uint32_t op, const uint64_t (&x)[L], const uint64_t (&a)[L], const uint64_t (&b)[L]) {
    bool lt = false, eq = true;  // row <=> a
#pragma unroll
    for (int l = 0; l < L; l++) {
        lt = lt || (eq && x[l] < a[l]);
        eq = eq && x[l] == a[l];
    }
    switch (op) {
    case 1: return eq;            // FilterModeEqual
    case 2: return !eq;           // NotEqual
    case 3: return !lt && !eq;    // Gt
    case 4: return !lt;           // Ge
    case 5: return lt;            // Lt
    case 6: return lt || eq;      // Le
    default: {                    // Range: a <= x && x <= b
        bool gt = false, e2 = true;
#pragma unroll
        for (int l = 0; l < L; l++) {
            gt = gt || (e2 && x[l] > b[l]);
            e2 = e2 && x[l] == b[l];
        }
        return !lt && !gt;
    }
    }
}

template <int L>
__global__ void __launch_bounds__(kBlock) wide_cmp_kernel(const WideArgs w) {
    constexpr int RPL = 128;               // rows per wave-load (64 lanes x 2)
    constexpr int WAVE_ROWS = RPL * kUnroll;
    constexpr uint64_t SIGN = 1ull << 63;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint64_t a[L], b[L];
#pragma unroll
    for (int l = 0; l < L; l++) { a[l] = w.a[l]; b[l] = w.b[l]; }
    uint32_t cnt = 0;
    for (uint64_t tile = blockIdx.x; tile < w.n_tiles; tile += gridDim.x) {
        const uint64_t row0 = (tile * kWavesPerBlock + (uint64_t)wave) * WAVE_ROWS;
        u32x4 r[L][kUnroll];
#pragma unroll
        for (int l = 0; l < L; l++) {
            const u32x4 *p = (const u32x4 *)w.limb[l] + row0 / 2 + lane;
#pragma unroll
            for (int u = 0; u < kUnroll; u++) r[l][u] = ld16<true>(p + u * 64);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            uint32_t q = 0;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                uint64_t x[L];
#pragma unroll
                for (int l = 0; l < L; l++) x[l] = elem<8>(r[l][u], j);
                x[0] ^= SIGN;
                q |= (wide_hit<L>(w.op, x, a, b) ? 1u : 0u) << j;
            }
            uint32_t word = pack_word<2>(q, lane);  // complete in every lane of the 16-lane group
            const uint64_t wi = ((row0 + (uint64_t)u * RPL) >> 5) + (lane >> 4);
            if ((lane & 15) == 0) {
                if (w.mask) word &= w.mask[wi];
                w.bits[wi] = word;
                cnt += __builtin_popcount(word);
            }
        }
    }
    // ragged tail by workgroup 0: thread t owns output bytes t, t + 256, ...
    if (blockIdx.x == 0) {
        const uint64_t start = w.n_tiles * (uint64_t)(kWavesPerBlock * WAVE_ROWS);
        if (start < w.n_rows) {
            const uint64_t nbytes = (w.n_rows - start + 7) >> 3;
            const uint8_t *mask8 = (const uint8_t *)w.mask;
            for (uint64_t bi = threadIdx.x; bi < nbytes; bi += blockDim.x) {
                uint32_t out = 0;
                for (int k = 0; k < 8; k++) {
                    const uint64_t row = start + 8 * bi + k;
                    if (row >= w.n_rows) break;
                    uint64_t x[L];
#pragma unroll
                    for (int l = 0; l < L; l++) x[l] = w.limb[l][row];
                    x[0] ^= SIGN;
                    out |= (wide_hit<L>(w.op, x, a, b) ? 1u : 0u) << k;
                }
                if (mask8) out &= mask8[(start >> 3) + bi];
                ((uint8_t *)w.bits)[(start >> 3) + bi] = (uint8_t)out;
                cnt += __builtin_popcount(out);
            }
        }
    }
    // count: wave reduce, one atomic per wave
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (lane == 0 && cnt && w.count) atomicAdd(w.count, (unsigned long long)cnt);
}

// Int128 / Int256 IN / NOT IN (i128InSetMatcher / i128NotInSetMatcher.MatchVector, match_i128.go:318-430 and
// the i256 twins): membership in a sorted set; keys[k * L + l] = limb l IN THE ORDER THE CALLER PASSES THE LIMB COLUMNS (round 4: least
// significant first, no sign flip -- membership needs no value order; was: most significant first, top limb
// already sign-flipped) of set value k.  One lane per row, bits.Set semantics (OR), mask-selected rows only.
template <int L>
__global__ void __launch_bounds__(kBlock) wide_inset_kernel(int negate, const uint64_t *l0, const uint64_t *l1, const uint64_t *l2,
                                                            const uint64_t *l3, uint64_t n_rows, const uint64_t *keys, uint32_t n_keys,
                                                            uint8_t *bits, const uint8_t *mask) {
    const uint64_t *limb[4] = {l0, l1, l2, l3};
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = (n_rows + 63) >> 6;
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row = (g << 6) + lane;
        bool m = false;
        if (row < n_rows) {
            uint64_t x[L];
#pragma unroll
            for (int l = 0; l < L; l++) x[l] = limb[l][row];
            uint32_t lo = 0, hi = n_keys;
            bool found = false;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint64_t *k = keys + (uint64_t)mid * L;
                int c = 0;  // key <=> x
#pragma unroll
                for (int l = 0; l < L; l++)
                    if (c == 0 && k[l] != x[l]) c = k[l] < x[l] ? -1 : 1;
                if (c == 0) { found = true; break; }
                if (c < 0) lo = mid + 1; else hi = mid;
            }
            m = negate ? !found : found;
        }
        const uint64_t ballot = __ballot(m);
        if (lane < 8) {
            const uint64_t r0 = (g << 6) + 8 * (uint64_t)lane;
            if (r0 < n_rows) {
                uint8_t byte = (uint8_t)(ballot >> (8 * lane));
                const uint64_t bi = r0 >> 3;
                if (mask != nullptr) byte &= mask[bi];
                if (byte) bits[bi] |= byte;
            }
        }
    }
}

// Round 4: the set's first compared limb in LDS (<= 4096 members): a branch-free binary search there, the other limbs of the members
// that share it compared from the L2; four groups of 64 rows per wave iteration.  Membership needs no value order, so the set is
// sorted -- and the limbs are compared -- LEAST significant limb first: that limb tells members apart where the top limbs of
// small 128- / 256-bit values are all zero (the callers pass the limb columns in that order: knox_abi.cpp knox_match_wide_set).
template <int L>
__global__ void __launch_bounds__(kBlock) wide_inset_prefix_kernel(int negate, const uint64_t *l0, const uint64_t *l1, const uint64_t *l2,
                                                                   const uint64_t *l3, uint64_t n_rows, const uint64_t *keys, uint32_t n_keys,
                                                                   uint8_t *bits, const uint8_t *mask) {
    constexpr int kG = 4;
    __shared__ uint64_t spre[4096];
    const uint64_t *limb[4] = {l0, l1, l2, l3};
    uint32_t pow2 = 1;
    while (pow2 < n_keys) pow2 <<= 1;
    for (uint32_t i = threadIdx.x; i < pow2; i += kBlock) spre[i] = i < n_keys ? keys[(uint64_t)i * L] : ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = (n_rows + 63) >> 6;
    for (uint64_t g0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kG; g0 < ngroups; g0 += nwaves_total * kG) {
        uint64_t x[kG][L];
#pragma unroll
        for (int k = 0; k < kG; k++) {
            const uint64_t row = ((g0 + k) << 6) + lane;
#pragma unroll
            for (int l = 0; l < L; l++) x[k][l] = row < n_rows ? limb[l][row] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < kG; k++) {
            if (g0 + k >= ngroups) break;  // wave uniform
            const uint64_t row = ((g0 + k) << 6) + lane;
            bool m = false;
            if (row < n_rows) {
                uint32_t pos = 0;          // first member whose first limb is >= the row's
                for (uint32_t step = pow2 >> 1; step >= 1; step >>= 1) pos += spre[pos + step - 1] < x[k][0] ? step : 0u;
                bool found = false;
                while (pos < n_keys && spre[pos] == x[k][0]) {
                    const uint64_t *kk = keys + (uint64_t)pos * L;
                    bool eq = true;
#pragma unroll
                    for (int l = 1; l < L; l++) eq = eq && kk[l] == x[k][l];
                    if (eq) { found = true; break; }
                    pos++;
                }
                m = negate ? !found : found;
            }
            const uint64_t ballot = __ballot(m);
            if (lane < 8) {
                const uint64_t r0 = ((g0 + k) << 6) + 8 * (uint64_t)lane;
                if (r0 < n_rows) {
                    uint8_t byte = (uint8_t)(ballot >> (8 * lane));
                    const uint64_t bi = r0 >> 3;
                    if (mask != nullptr) byte &= mask[bi];
                    if (byte) bits[bi] |= byte;
                }
            }
        }
    }
}

hipError_t launch_wide_inset(hipStream_t s, int n_limbs, bool negate, const uint64_t *const *limbs, uint64_t n_rows, const uint64_t *keys,
                             uint32_t n_keys, uint8_t *bits, const uint8_t *mask) {
    if (n_rows == 0) return hipSuccess;
    uint64_t g = (n_rows + kBlock - 1) / kBlock;
    if (g > 2048 * 4) g = 2048 * 4;
    const char *form = getenv("KNOX_WIDE_IN_FORM");
    if (n_keys <= 4096 && !(form && !strcmp(form, "general"))) {
        uint64_t gp = (n_rows + (uint64_t)kBlock * 4 - 1) / ((uint64_t)kBlock * 4);
        if (gp > 2048 * 4) gp = 2048 * 4;
        if (n_limbs == 2) KNOX_LAUNCH(wide_inset_prefix_kernel<2>, (int)gp, negate ? 1 : 0, limbs[0], limbs[1], nullptr, nullptr, n_rows, keys, n_keys, bits, mask);
        if (n_limbs == 4) KNOX_LAUNCH(wide_inset_prefix_kernel<4>, (int)gp, negate ? 1 : 0, limbs[0], limbs[1], limbs[2], limbs[3], n_rows, keys, n_keys, bits, mask);
        return hipErrorInvalidValue;
    }
    if (n_limbs == 2) KNOX_LAUNCH(wide_inset_kernel<2>, (int)g, negate ? 1 : 0, limbs[0], limbs[1], nullptr, nullptr, n_rows, keys, n_keys, bits, mask);
    if (n_limbs == 4) KNOX_LAUNCH(wide_inset_kernel<4>, (int)g, negate ? 1 : 0, limbs[0], limbs[1], limbs[2], limbs[3], n_rows, keys, n_keys, bits, mask);
    return hipErrorInvalidValue;
}

// numInSetMatcher.MatchRangeVectors over materialized min / max statistics vectors (operator/filter/match_num.go:693-727): one
// row per pack; minU64, maxU64 = uint64(min), uint64(max) -- Go's conversion sign-extends a signed T -- swapped when
// minU64 > maxU64 ("source could contain negative integers"); the pack MAY match when the set holds a member in
// [minU64, maxU64] (xroar ContainsRange).  keys = the set's members converted the same way, sorted ascending, unique.
// bits are overwritten: hit, and-ed with the mask when there is one (the reference tests masked rows only and Sets).
template <int W, bool SIGNED>
__global__ void __launch_bounds__(kBlock) range_set_kernel(const void *mins, const void *maxs, uint64_t n_rows, const uint64_t *keys, uint32_t n_keys,
                                                           uint64_t *bits, const uint64_t *mask) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = (n_rows + 63) >> 6;
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row = (g << 6) + lane;
        bool m = false;
        if (row < n_rows) {
            uint64_t lo_v, hi_v;
            if constexpr (SIGNED) {
                lo_v = sext_u64<W>(load_elem<W>(mins, row));
                hi_v = sext_u64<W>(load_elem<W>(maxs, row));
            } else {
                lo_v = (uint64_t)load_elem<W>(mins, row);
                hi_v = (uint64_t)load_elem<W>(maxs, row);
            }
            if (lo_v > hi_v) { const uint64_t t = lo_v; lo_v = hi_v; hi_v = t; }
            uint32_t lo = 0, hi = n_keys;  // first key >= lo_v
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (keys[mid] < lo_v) lo = mid + 1; else hi = mid;
            }
            m = lo < n_keys && keys[lo] <= hi_v;
        }
        uint64_t word = __ballot(m);
        if (mask != nullptr) word &= mask[g];
        if (lane == 0) bits[g] = word;
    }
}

hipError_t launch_range_set(hipStream_t s, int width, bool is_signed, const void *mins, const void *maxs, uint64_t n_rows, const uint64_t *keys,
                            uint32_t n_keys, uint64_t *bits, const uint64_t *mask) {
    if (n_rows == 0) return hipSuccess;
    const int g = (int)std::min<uint64_t>((n_rows + kBlock - 1) / kBlock, 4096);
#define KNOX_RS(W)                                                                                                \
    if (is_signed) KNOX_LAUNCH((range_set_kernel<W, true>), g, mins, maxs, n_rows, keys, n_keys, bits, mask);    \
    KNOX_LAUNCH((range_set_kernel<W, false>), g, mins, maxs, n_rows, keys, n_keys, bits, mask)
    switch (width) {
    case 8: KNOX_RS(8);
    case 4: KNOX_RS(4);
    case 2: KNOX_RS(2);
    case 1: KNOX_RS(1);
    default: return hipErrorInvalidValue;
    }
#undef KNOX_RS
}

hipError_t launch_wide_cmp(hipStream_t s, int n_limbs, int grid, const WideArgs &w) {
    if (grid < 1) grid = 1;
    if (n_limbs == 2) KNOX_LAUNCH((wide_cmp_kernel<2>), grid, w);
    if (n_limbs == 4) KNOX_LAUNCH((wide_cmp_kernel<4>), grid, w);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// bitset combinators (internal/bitset/generic/bitset.go:13-330) on 64-bit words.
// The last word is masked to `size` bits (bytemask semantics extended to the padded word).
// flags[0] |= any set bit, flags[1] |= any valid bit clear (all == !flags[1]).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t valid_mask(uint64_t word, uint64_t nwords, uint64_t size_bits) {
    if (word + 1 < nwords || (size_bits & 63) == 0) return ~0ull;
    return (~0ull) >> (64 - (size_bits & 63));
}

template <int OP>
__global__ void __launch_bounds__(kBlock) bitop_kernel(uint64_t *dst, const uint64_t *src, uint64_t nwords,
                                                       uint64_t size_bits, uint32_t *flags) {
    uint64_t any = 0, notall = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t vm = valid_mask(i, nwords, size_bits);
        uint64_t d = dst[i];
        uint64_t sv = 0;
        if constexpr (OP == BIT_AND || OP == BIT_ANDNOT || OP == BIT_OR || OP == BIT_XOR || OP == BIT_COPY) sv = src[i];
        if constexpr (OP == BIT_AND) d &= sv;
        if constexpr (OP == BIT_ANDNOT) d &= ~sv;
        if constexpr (OP == BIT_OR) d |= sv;
        if constexpr (OP == BIT_XOR) d ^= sv;
        if constexpr (OP == BIT_NEG) d = ~d;
        if constexpr (OP == BIT_COPY) d = sv;
        if constexpr (OP == BIT_ONE) d = ~0ull;
        d &= vm;
        dst[i] = d;
        any |= d;
        notall |= (~d) & vm;
    }
    if (flags != nullptr) {
        // one look per WORKGROUP (same-address traffic serializes in L2): only the first few workgroups that find a flag
        // clear pay for the read-modify-write
        const int wany = __syncthreads_or(any != 0);
        const int wnot = __syncthreads_or(notall != 0);
        if (threadIdx.x == 0) {
            if (wany && __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flags[0], 1u);
            if (wnot && __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flags[1], 1u);
        }
    }
}

hipError_t launch_bitop(hipStream_t s, int op, uint64_t *dst, const uint64_t *src, uint64_t size_bits, uint32_t *flags) {
    const uint64_t nwords = (size_bits + 63) >> 6;
    if (nwords == 0) return hipSuccess;
    uint64_t g = (nwords + kBlock - 1) / kBlock;
    if (g > 1024) g = 1024;
    const int grid = (int)g;
    switch (op) {
    case BIT_AND: KNOX_LAUNCH((bitop_kernel<BIT_AND>), grid, dst, src, nwords, size_bits, flags);
    case BIT_ANDNOT: KNOX_LAUNCH((bitop_kernel<BIT_ANDNOT>), grid, dst, src, nwords, size_bits, flags);
    case BIT_OR: KNOX_LAUNCH((bitop_kernel<BIT_OR>), grid, dst, src, nwords, size_bits, flags);
    case BIT_XOR: KNOX_LAUNCH((bitop_kernel<BIT_XOR>), grid, dst, src, nwords, size_bits, flags);
    case BIT_NEG: KNOX_LAUNCH((bitop_kernel<BIT_NEG>), grid, dst, src, nwords, size_bits, flags);
    case BIT_COPY: KNOX_LAUNCH((bitop_kernel<BIT_COPY>), grid, dst, src, nwords, size_bits, flags);
    case BIT_ONE: KNOX_LAUNCH((bitop_kernel<BIT_ONE>), grid, dst, src, nwords, size_bits, flags);
    default: return hipErrorInvalidValue;
    }
}

// PopCount generic/bitset.go:332-353
__global__ void __launch_bounds__(kBlock) popcount_kernel(const uint64_t *bits, uint64_t nwords, uint64_t size_bits,
                                                          unsigned long long *out) {
    uint64_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x)
        c += __builtin_popcountll(bits[i] & valid_mask(i, nwords, size_bits));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += shfl_down_u64(c, off);
    __shared__ uint64_t sm[kBlock / 64];  // one atomic per workgroup: same-address atomics serialize in L2
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += sm[w];
        if (t) atomicAdd(out, (unsigned long long)t);
    }
}

hipError_t launch_popcount(hipStream_t s, const uint64_t *bits, uint64_t size_bits, unsigned long long *count_out) {
    const uint64_t nwords = (size_bits + 63) >> 6;
    if (nwords == 0) return hipSuccess;
    uint64_t g = (nwords + kBlock - 1) / kBlock;
    if (g > 2048) g = 2048;
    KNOX_LAUNCH(popcount_kernel, (int)g, bits, nwords, size_bits, count_out);
}

// per-segment popcount: one workgroup per segment (seg_bits multiple of 64)
// skip (nullable): segments whose bits are known to be zero (the pruned packs of a scan) -- their count is 0 without a read: a
// bloom-pruned point lookup counted the whole 128 MiB bitset of 2^30 rows for its five surviving packs (42 of its ~100 us)
__global__ void __launch_bounds__(kBlock) popcount_seg_kernel(const uint64_t *bits, uint64_t nwords, uint64_t size_bits,
                                                              uint64_t seg_words, int64_t *counts, uint64_t n_seg, const uint8_t *skip,
                                                              unsigned long long *total) {
    __shared__ uint64_t sm[kBlock / 64];
    for (uint64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        if (skip != nullptr && skip[seg]) {   // workgroup uniform
            if (threadIdx.x == 0) counts[seg] = 0;
            continue;
        }
        const uint64_t w0 = seg * seg_words;
        uint64_t w1 = w0 + seg_words;
        if (w1 > nwords) w1 = nwords;
        uint64_t c = 0;
        if (((w0 | w1) & 1) == 0 && w1 < nwords) {
            // whole 16-byte vectors, no partial last word in this segment: 16 B per lane, 4 loads in flight
            const u32x4 *v = (const u32x4 *)(bits + w0);
            const uint64_t nvec = (w1 - w0) >> 1;
#pragma unroll 4
            for (uint64_t k = threadIdx.x; k < nvec; k += blockDim.x) {
                const u32x4 x = v[k];
                c += __builtin_popcount(x.x) + __builtin_popcount(x.y) + __builtin_popcount(x.z) + __builtin_popcount(x.w);
            }
        } else {
            for (uint64_t i = w0 + threadIdx.x; i < w1; i += blockDim.x) c += __builtin_popcountll(bits[i] & valid_mask(i, nwords, size_bits));
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += shfl_down_u64(c, off);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint64_t c4 = sm[0] + sm[1] + sm[2] + sm[3];
            counts[seg] = (int64_t)c4;
            if (total != nullptr && c4) atomicAdd(total, (unsigned long long)c4);   // total (nullable): the bitset's popcount with the same pass
        }
        __syncthreads();
    }
}

hipError_t launch_popcount_segments(hipStream_t s, const uint64_t *bits, uint64_t size_bits, uint64_t seg_bits,
                                    int64_t *counts, uint64_t n_seg, const uint8_t *skip, unsigned long long *total) {
    if (n_seg == 0) return hipSuccess;
    const uint64_t nwords = (size_bits + 63) >> 6;
    uint64_t g = n_seg > 65535 ? 65535 : n_seg;
    KNOX_LAUNCH(popcount_seg_kernel, (int)g, bits, nwords, size_bits, seg_bits >> 6, counts, n_seg, skip, total);
}

// bytes of device memory set to zero by one streaming kernel (16-byte nontemporal stores; ptr 8-byte aligned): the runtime's fill
// kernel behind hipMemsetAsync writes ~2 TB/s, and the multi-pass plans clear a 128 MiB scratch bitset per leaf
__global__ void __launch_bounds__(kBlock) zero_bytes_kernel(uint8_t *p, uint64_t bytes) {
    const uint64_t head = ((16 - ((uintptr_t)p & 15)) & 15) < bytes ? ((16 - ((uintptr_t)p & 15)) & 15) : bytes;
    const uint64_t nvec = (bytes - head) >> 4;
    const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x, nth = (uint64_t)gridDim.x * kBlock;
    if (tid < head) p[tid] = 0;
    const u32x4 z = {0u, 0u, 0u, 0u};
    u32x4 *v = (u32x4 *)(p + head);
    for (uint64_t i = tid; i < nvec; i += nth) __builtin_nontemporal_store(z, v + i);
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + tid < bytes) p[tail0 + tid] = 0;
}

hipError_t launch_zero_bytes(hipStream_t s, void *p, uint64_t bytes) {
    if (bytes == 0) return hipSuccess;
    uint64_t g = (bytes / 16 + kBlock - 1) / kBlock;
    if (g > 256 * 16) g = 256 * 16;
    if (g == 0) g = 1;
    KNOX_LAUNCH(zero_bytes_kernel, (int)g, (uint8_t *)p, bytes);
}

// ---------------------------------------------------------------------------------------------
// Indexes (generic/bitset.go:355-396): bitset -> ascending uint32 row ids.
// chunk = 16384 bits = 256 words per workgroup; counts -> exclusive scan -> ordered write.
// ---------------------------------------------------------------------------------------------
constexpr int kIdxWords = kBlock;  // one 64-bit word per thread: 16384 bits per workgroup

__global__ void __launch_bounds__(kBlock) idx_count_kernel(const uint64_t *bits, uint64_t nwords, uint64_t size_bits,
                                                           uint32_t *block_counts) {
    __shared__ uint32_t sm[kBlock / 64];
    const uint64_t i = (uint64_t)blockIdx.x * kIdxWords + threadIdx.x;
    uint32_t c = 0;
    if (i < nwords) c = __builtin_popcountll(bits[i] & valid_mask(i, nwords, size_bits));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// inclusive scan of one value per thread over a 64-lane wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// single workgroup exclusive scan in place, 4 entries per thread per round (4096 per round);
// total written to counts[n] and *n_out
__global__ void __launch_bounds__(1024) idx_scan_kernel(uint32_t *counts, uint64_t n, unsigned long long *n_out) {
    __shared__ uint32_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;  // identical in every thread
    for (uint64_t base = 0; base < n; base += 4096) {
        const uint64_t i = base + 4 * (uint64_t)threadIdx.x;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + k < n ? counts[i + k] : 0;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        const uint32_t incl = wave_incl_scan(mine, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t t = wsum[w];
            before += w < wave ? t : 0;
            total += t;
        }
        uint32_t o = carry + before + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k < n) counts[i + k] = o;
            o += v[k];
        }
        carry += total;
        __syncthreads();  // wsum is rewritten next round
    }
    if (threadIdx.x == 0) {
        counts[n] = carry;
        *n_out = carry;
    }
}

__global__ void __launch_bounds__(kBlock) idx_write_kernel(const uint64_t *bits, uint64_t nwords, uint64_t size_bits,
                                                           const uint32_t *block_offsets, uint32_t *rows_out, uint64_t capacity) {
    __shared__ uint32_t wsum[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * kIdxWords + threadIdx.x;
    uint64_t w = 0;
    if (i < nwords) w = bits[i] & valid_mask(i, nwords, size_bits);
    const uint32_t c = __builtin_popcountll(w);
    const uint32_t incl = wave_incl_scan(c, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int k = 0; k < kBlock / 64; k++) before += k < wave ? wsum[k] : 0;
    if (c) {
        uint64_t o = (uint64_t)block_offsets[blockIdx.x] + before + incl - c;
        const uint32_t row0 = (uint32_t)(i << 6);
        while (w) {
            const int k = __builtin_ctzll(w);
            if (o < capacity) rows_out[o] = row0 + k;
            o++;
            w &= w - 1;
        }
    }
}

// exclusive scan of n 32-bit counts in place by one workgroup, the total to *n_out (also used by the simple8 decoder)
hipError_t launch_idx_scan(hipStream_t s, uint32_t *counts, uint64_t n, unsigned long long *n_out) {
    hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, s, counts, n, n_out);
    return hipGetLastError();
}

hipError_t launch_indexes(hipStream_t s, const uint64_t *bits, uint64_t size_bits, uint32_t *scratch, uint32_t *rows_out,
                          uint64_t capacity, unsigned long long *n_out) {
    const uint64_t nwords = (size_bits + 63) >> 6;
    const uint64_t nblocks = (nwords + kIdxWords - 1) / kIdxWords;
    if (nblocks == 0) return hipMemsetAsync(n_out, 0, sizeof(*n_out), s);
    hipLaunchKernelGGL(idx_count_kernel, dim3((unsigned)nblocks), dim3(kBlock), 0, s, bits, nwords, size_bits, scratch);
    hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, s, scratch, nblocks, n_out);
    hipLaunchKernelGGL(idx_write_kernel, dim3((unsigned)nblocks), dim3(kBlock), 0, s, bits, nwords, size_bits, scratch, rows_out,
                       capacity);
    return hipGetLastError();
}

// The bitset ranges of PRUNED packs cleared in one launch ("pruned = no row matches" is part of a scan's contract).  The runtime's
// fill kernel behind hipMemsetAsync writes ~2 TB/s (profiles/r04_*: __amd_rocclr_fillBufferAligned), and a pruned scan issued up to
// sixteen of them: zeroing the 128 MiB bitset of 2^30 rows took 60 of the ~100 us a bloom-pruned point lookup cost, four times
// its scan kernel.  One workgroup per 16 KiB slice of a pack's range; slices of surviving packs return at once.
__global__ void __launch_bounds__(kBlock) zero_pruned_kernel(uint8_t *bits, const uint8_t *skip, uint32_t n_packs, uint64_t pack_bytes,
                                                             uint64_t bits_bytes, uint32_t slices_per_pack) {
    const uint32_t p = blockIdx.x / slices_per_pack, sl = blockIdx.x % slices_per_pack;
    if (p >= n_packs || !skip[p]) return;
    const uint64_t slice = (pack_bytes + slices_per_pack - 1) / slices_per_pack;
    uint64_t b0 = (uint64_t)p * pack_bytes + (uint64_t)sl * slice;
    uint64_t b1 = b0 + slice;
    const uint64_t pend = (uint64_t)(p + 1) * pack_bytes;
    if (b1 > pend) b1 = pend;
    if (b1 > bits_bytes) b1 = bits_bytes;
    if (b0 >= b1) return;
    // pack_bytes is a multiple of 8 (pack_rows of 64), so b0 is: 16-byte stores in the aligned middle, 8-byte ones around it, single
    // bytes only behind the last whole word of the bitset (nothing behind its (n + 7) / 8 bytes is touched)
    const uint64_t w1 = b1 & ~7ull;
    uint64_t a0 = (b0 + 15) & ~15ull, a1 = w1 & ~15ull;
    if (a0 > a1) a0 = a1 = b0;
    for (uint64_t b = b0 + 8ull * threadIdx.x; b < a0 && b < w1; b += 8ull * kBlock) *(uint64_t *)(bits + b) = 0ull;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (uint64_t b = a0 + 16ull * threadIdx.x; b < a1; b += 16ull * kBlock) __builtin_nontemporal_store(z, (u32x4 *)(bits + b));
    for (uint64_t b = (a1 > b0 ? a1 : b0) + 8ull * threadIdx.x; b < w1; b += 8ull * kBlock) *(uint64_t *)(bits + b) = 0ull;
    for (uint64_t b = (w1 > b0 ? w1 : b0) + threadIdx.x; b < b1; b += kBlock) bits[b] = 0;
}

hipError_t launch_zero_pruned(hipStream_t s, uint8_t *bits, const uint8_t *d_skip, uint32_t n_packs, uint64_t pack_rows, uint64_t n_rows) {
    if (n_packs == 0) return hipSuccess;
    const uint64_t pack_bytes = pack_rows >> 3, bits_bytes = (n_rows + 7) >> 3;
    uint32_t slices = (uint32_t)((pack_bytes + 16383) / 16384);
    if (slices == 0) slices = 1;
    while ((uint64_t)n_packs * slices > 0x7fffffffull && slices > 1) slices >>= 1;
    hipLaunchKernelGGL(zero_pruned_kernel, dim3(n_packs * slices), dim3(kBlock), 0, s, bits, d_skip, n_packs, pack_bytes, bits_bytes, slices);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// IN / NOT IN set matchers (operator/filter/match_num.go:672-690,789-808, match_float.go:57-73):
// one lane = one row; the value's canonical order key (floats: -0 -> +0, NaN never a member) is
// looked up by binary search in the sorted key set; wave ballot -> 8 result bytes OR-ed into bits.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) inset_kernel(uint32_t flavor, int negate, const void *col, uint64_t n_rows,
                                                       const uint64_t *keys, uint32_t n_keys, uint8_t *bits, const uint8_t *mask) {
    using U = typename WT<W>::U;
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = (n_rows + 63) >> 6;
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row = (g << 6) + lane;
        bool m = false;
        if (row < n_rows) {
            const U e = load_elem<W>(col, row);
            const U mag = e & (WT<W>::MASK ^ WT<W>::SIGN);
            uint64_t key;
            bool isnan = false;
            if (flavor == FL_FLOAT) {
                U expmask;
                if constexpr (W == 8) expmask = 0x7ff0000000000000ull; else expmask = 0x7f800000u;
                isnan = mag > expmask;
                key = (mag == 0) ? (uint64_t)WT<W>::SIGN : (uint64_t)fkey<W>(e);
            } else {
                key = (flavor == FL_SIGNED) ? (uint64_t)(e ^ WT<W>::SIGN) : (uint64_t)e;
            }
            bool found = false;
            if (!isnan) {
                uint32_t lo = 0, hi = n_keys;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint64_t k = keys[mid];
                    if (k < key) lo = mid + 1; else hi = mid;
                }
                found = lo < n_keys && keys[lo] == key;
            }
            m = negate ? !found : found;
        }
        const uint64_t ballot = __ballot(m);
        if (lane < 8) {
            const uint64_t r0 = (g << 6) + 8 * (uint64_t)lane;
            if (r0 < n_rows) {
                uint8_t byte = (uint8_t)(ballot >> (8 * lane));
                const uint64_t bi = r0 >> 3;
                if (mask != nullptr) byte &= mask[bi];
                if (byte) bits[bi] |= byte;
            }
        }
    }
}

// Round 4: the same matcher at streaming shape.  The kernel above (one row per lane, a binary search through global memory, result
// BYTES read-modified-written) ran at 335 G rows/s for ONE key and 23 G rows/s for 64 Ki keys (tools/bench_inset.py, 2^28 int64
// rows; a range scan of the same column: 900).  Here a wave takes whole tiles as the scans do -- four 16-byte loads per lane in
// flight, K rows per load -- and the lookup depends on the set:
//   1- / 2-byte columns    a 65 536-bit membership bitmap in LDS (built per workgroup from the keys): one bit test per row, any set size
//   up to 8 keys           the keys in turn from scalar registers against all of the lane's rows (one compare per row and key)
//   up to 4096 keys        the sorted keys in LDS under radix buckets (top bits of key - smallest member): a row searches its bucket only
//   larger sets            the host's radix buckets over the whole set: bounds and members through the L2, two or three dependent loads
//                          (without a bucket table from the caller: the row-per-lane kernel)
// Result bits leave as 32-bit words (the DPP word assembly of the scans), OR-ed into the caller's bits as before.  The rows behind
// the last whole tile go through the kernel above.
constexpr uint32_t kInsetRegKeys = 8, kInsetBranchFree = 256, kInsetLdsKeys = 4096;

// TIER (the host picks it from the set's size, one instantiation each so that no lookup pays for another's registers and branches):
// 0 = bitmap (1- / 2-byte columns) or scalar keys (<= 8), 1 = branch-free binary search in LDS (<= 256), 2 = radix buckets in LDS
// (<= 4096), 3 = pivots in LDS + the L2
template <int W, int TIER>
__global__ void __launch_bounds__(kBlock) inset_tile_kernel(uint32_t flavor, int negate, const void *col, uint64_t n_tiles, const uint64_t *keys,
                                                            uint32_t n_keys, const uint32_t *gstart, uint32_t g_buckets, uint32_t g_shift,
                                                            uint32_t *bits32, const uint32_t *mask32) {
    using U = typename WT<W>::U;
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;
    constexpr uint64_t WAVE_ROWS = (uint64_t)RPL * kUnroll;
    __shared__ uint64_t skeys[(W >= 4 && TIER >= 1) ? kInsetLdsKeys : 1];
    __shared__ uint16_t bstart[(W >= 4 && TIER == 2) ? kInsetLdsKeys + 2 : 1];   // radix buckets over the sorted keys (see the LDS tier below)
    __shared__ uint32_t sbm[W <= 2 ? 2048 : 1];
    const int lane = threadIdx.x & 63;
    uint32_t pow2 = 1, b_shift = 0;
    uint64_t k_min = 0;
    while (pow2 < n_keys) pow2 <<= 1;
    if constexpr (W <= 2) {
        for (int i = threadIdx.x; i < 2048; i += kBlock) sbm[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_keys; i += kBlock) {
            const uint64_t k = keys[i];
            if (k < 65536) atomicOr(&sbm[k >> 5], 1u << (k & 31));
        }
        __syncthreads();
    } else if constexpr (TIER == 1 || TIER == 2) {
        for (uint32_t i = threadIdx.x; i < pow2; i += kBlock) skeys[i] = i < n_keys ? keys[i] : ~0ull;
        __syncthreads();
      if constexpr (TIER == 2) {
        // buckets by the top bits of (key - smallest member): bstart[b] = first member of bucket b.  A row reads its bucket's two
        // bounds and searches between them -- one or two members when the set is spread evenly, the whole binary search at
        // worst -- instead of log2(n) steps over the whole set (the LDS tier was bound by exactly those reads)
        k_min = skeys[0];
        const uint64_t range = skeys[n_keys - 1] - k_min;
        b_shift = 0;
        while ((range >> b_shift) >= (uint64_t)pow2) b_shift++;
        for (uint32_t b = threadIdx.x; b <= pow2; b += kBlock) {
            const uint64_t first = ((uint64_t)b << b_shift);         // the bucket's first offset; b == pow2: behind the last member
            uint32_t lo = 0, hi = n_keys;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (skeys[mid] - k_min < first) lo = mid + 1; else hi = mid;
            }
            bstart[b] = (uint16_t)(b == pow2 ? n_keys : lo);
        }
        __syncthreads();
      }
    } else if constexpr (TIER == 3) {
        k_min = keys[0];
    }
    const uint64_t wave0 = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * kWavesPerBlock;
    for (uint64_t t = wave0; t < n_tiles; t += n_waves) {
        const uint64_t row0 = t * WAVE_ROWS;
        const u32x4 *p = (const u32x4 *)col + row0 / K + lane;
        u32x4 r[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) r[u] = ld16<true>(p + u * 64);
        uint32_t q[kUnroll] = {0, 0, 0, 0};
        // the rows' canonical order keys (floats: -0 -> +0, NaN never a member), as the kernel above forms them
        auto key_of = [&](U e, bool &isnan) -> uint64_t {
            isnan = false;
            if (flavor == FL_FLOAT) {
                const U mag = e & (WT<W>::MASK ^ WT<W>::SIGN);
                U expmask;
                if constexpr (W == 8) expmask = 0x7ff0000000000000ull; else expmask = (U)0x7f800000u;
                isnan = mag > expmask;
                return (mag == 0) ? (uint64_t)WT<W>::SIGN : (uint64_t)fkey<W>(e);
            }
            return (flavor == FL_SIGNED) ? (uint64_t)(U)(e ^ WT<W>::SIGN) : (uint64_t)e;
        };
        if constexpr (W <= 2) {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    bool nan_;
                    const uint32_t k = (uint32_t)key_of(elem<W>(r[u], j), nan_);
                    q[u] |= ((sbm[k >> 5] >> (k & 31)) & 1u) << j;
                }
            }
        } else if constexpr (TIER == 0) {
            U ek[kUnroll][K];
            uint32_t nanq[kUnroll] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    bool nan_;
                    ek[u][j] = (U)key_of(elem<W>(r[u], j), nan_);
                    nanq[u] |= (nan_ ? 1u : 0u) << j;
                }
            }
            for (uint32_t kk = 0; kk < n_keys; kk++) {
                const uint64_t kv64 = keys[kk];                       // uniform address: a scalar load
                const bool fits = kv64 <= (uint64_t)WT<W>::MASK;      // a key beyond the column type's range matches nothing
                const U kv = (U)kv64;
#pragma unroll
                for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                    for (int j = 0; j < K; j++) q[u] |= ((fits && ek[u][j] == kv) ? 1u : 0u) << j;
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) q[u] &= ~nanq[u];
        } else {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    bool nan_;
                    const uint64_t key = key_of(elem<W>(r[u], j), nan_);
                    bool found;
                    if constexpr (TIER == 1) {                               // few steps: the branch-free search costs less than the bucket's two bounds and a divergent loop (measured: 16 keys 0.54 against 0.97 ms, 64 keys 0.66 against 0.83; 1024 keys 1.30 against 0.93)
                        uint32_t pos = 0;
                        for (uint32_t step = pow2 >> 1; step >= 1; step >>= 1) pos += skeys[pos + step - 1] < key ? step : 0u;
                        found = pos < n_keys && skeys[pos] == key;
                    } else if constexpr (TIER == 2) {
                        const uint64_t off = key - k_min;                    // below the smallest member: wraps far above every bucket
                        const uint64_t b = off >> b_shift;
                        found = false;
                        if (b < (uint64_t)pow2) {
                            uint32_t lo = bstart[b], hi = bstart[b + 1];
                            while (lo < hi) {
                                const uint32_t mid = (lo + hi) >> 1;
                                const uint64_t km = skeys[mid];
                                if (km == key) { found = true; break; }
                                if (km < key) lo = mid + 1; else hi = mid;
                            }
                        }
                    } else {                                                 // the host's radix buckets over the whole set, bounds and members through the L2
                        const uint64_t b = (key - k_min) >> g_shift;
                        found = false;
                        if (b < (uint64_t)g_buckets) {
                            uint32_t lo = gstart[b], hi = gstart[b + 1];
                            while (lo < hi) {
                                const uint32_t mid = (lo + hi) >> 1;
                                const uint64_t km = keys[mid];
                                if (km == key) { found = true; break; }
                                if (km < key) lo = mid + 1; else hi = mid;
                            }
                        }
                    }
                    q[u] |= ((found && !nan_) ? 1u : 0u) << j;
                }
            }
        }
        constexpr uint32_t FULL = K == 32 ? 0xffffffffu : (1u << K) - 1u;
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const uint32_t hit = negate ? (~q[u] & FULL) : q[u];
            uint32_t w = pack_word<K>(hit, lane);
            if ((lane & (G - 1)) == 0) {
                const uint64_t idx = ((row0 + (uint64_t)u * RPL) >> 5) + (uint64_t)(lane / G);
                if (mask32 != nullptr) w &= mask32[idx];
                if (w) bits32[idx] |= w;
            }
        }
    }
}

// gstart (nullable): the host's radix buckets over a large set -- gstart[b] = first member with (member - keys[0]) >> g_shift >= b,
// g_buckets + 1 entries
hipError_t launch_inset(hipStream_t s, int width, uint32_t flavor, bool negate, const void *col, uint64_t n_rows,
                        const uint64_t *keys, uint32_t n_keys, const uint32_t *gstart, uint32_t g_buckets, uint32_t g_shift, uint8_t *bits,
                        const uint8_t *mask) {
    if (n_rows == 0) return hipSuccess;
    // whole tiles through the streaming kernel (KNOX_INSET_FORM=rows pins the row-per-lane kernel: measurements, tests); the rest, and
    // columns or bitsets that are not 16- / 4-byte aligned, through the row-per-lane kernel
    const char *form = getenv("KNOX_INSET_FORM");
    const uint64_t wave_rows = (uint64_t)64 * (16 / (uint64_t)width) * kUnroll;
    uint64_t n_tiles = n_rows / wave_rows;
    if ((form && !strcmp(form, "rows")) || ((uintptr_t)col & 15) || ((uintptr_t)bits & 3) || ((uintptr_t)mask & 3)) n_tiles = 0;
    if (width >= 4 && n_keys > kInsetLdsKeys && gstart == nullptr) n_tiles = 0;
    if (n_tiles) {
        uint64_t g = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
        if (g > 256 * 8) g = 256 * 8;
        const int tier = width <= 2 || n_keys <= kInsetRegKeys ? 0 : n_keys <= kInsetBranchFree ? 1 : n_keys <= kInsetLdsKeys ? 2 : 3;
#define KNOX_INSET_T(W, T)                                                                                                                      \
    hipLaunchKernelGGL((inset_tile_kernel<W, T>), dim3((unsigned)g), dim3(kBlock), 0, s, flavor, negate ? 1 : 0, col, n_tiles, keys, n_keys, \
                       gstart, g_buckets, g_shift, (uint32_t *)bits, (const uint32_t *)mask)
#define KNOX_INSET_W(W)                                                                        \
    do {                                                                                       \
        if (tier == 0) KNOX_INSET_T(W, 0);                                                     \
        else if (tier == 1) KNOX_INSET_T(W, 1);                                                \
        else if (tier == 2) KNOX_INSET_T(W, 2);                                                \
        else KNOX_INSET_T(W, 3);                                                               \
    } while (0)
        switch (width) {
        case 8: KNOX_INSET_W(8); break;
        case 4: KNOX_INSET_W(4); break;
        case 2: KNOX_INSET_T(2, 0); break;
        case 1: KNOX_INSET_T(1, 0); break;
        default: return hipErrorInvalidValue;
        }
#undef KNOX_INSET_W
#undef KNOX_INSET_T
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const uint64_t done = n_tiles * wave_rows;   // a multiple of 1024 rows: byte aligned in the bitset
    if (done == n_rows) return hipSuccess;
    const uint64_t left = n_rows - done;
    const void *tcol = (const uint8_t *)col + done * (uint64_t)width;
    uint8_t *tbits = bits + (done >> 3);
    const uint8_t *tmask = mask ? mask + (done >> 3) : nullptr;
    uint64_t g = (left + kBlock - 1) / kBlock;
    if (g > 2048 * 4) g = 2048 * 4;
    switch (width) {
    case 8: KNOX_LAUNCH((inset_kernel<8>), (int)g, flavor, negate ? 1 : 0, tcol, left, keys, n_keys, tbits, tmask);
    case 4: KNOX_LAUNCH((inset_kernel<4>), (int)g, flavor, negate ? 1 : 0, tcol, left, keys, n_keys, tbits, tmask);
    case 2: KNOX_LAUNCH((inset_kernel<2>), (int)g, flavor, negate ? 1 : 0, tcol, left, keys, n_keys, tbits, tmask);
    case 1: KNOX_LAUNCH((inset_kernel<1>), (int)g, flavor, negate ? 1 : 0, tcol, left, keys, n_keys, tbits, tmask);
    default: return hipErrorInvalidValue;
    }
}


}  // namespace knox
