// knox_dev.h -- device-side helpers shared by the gfx950 kernel translation units (knox_scan.hip, knox_agg.hip,
// knox_window.hip, knox_bitset.hip, knox_bytes.hip, knox_bloom.hip, knox_decode.hip).  Device code only: include from
// .hip sources after knox_kernels.h.  Nothing here is visible outside libknoxgpu.so.
//
// What runs here replaces, for decoded column vectors resident in HBM:
//   internal/cmp/number.go:13-243, float.go:13-242      predicate -> LSB-first bitset + count
//   internal/bitset/generic/bitset.go:13-435            and/andnot/or/xor/neg(+flags)/popcount/indexes
//   internal/reducer/reducer.go:138-349                 count/sum/min/max/first/last/mean state
//   pkg/stringx/match.go:20-84                          bytes EQ / NE
//   internal/filter/bloom/bloom.go:136-150              bloom probe
//
// Design (DESIGN.md has the long form):
//   * HBM-bound integer work: no MFMA, no LDS staging of values.  Every lane issues kUnroll
//     16-byte loads per column per iteration (1 KiB per wave instruction, fully coalesced),
//     a 256-thread workgroup streams 16 KiB contiguous per column per iteration.
//   * All compare modes of all types are ONE integer range test in "key space":
//         hit = U(key(x) - lo) <= diff
//     ints use their raw bits (two's-complement differences are order independent, this is
//     the reference's own Between form number.go:211-243); floats map through the IEEE
//     order key (sign-flip trick) so NaN falls outside every range and -0 == +0 is handled
//     by the host's choice of bounds.  NE / empty / full are negations folded into a truth
//     table over the leaf hits, which also evaluates any AND/OR tree of <= 4 leaves in
//     registers (filter.Match, match_core.go:14-215, without scratch bitsets).
//   * A lane owns K = 16/sizeof(T) ADJACENT rows, so a wave64 ballot would interleave bits.
//     Instead each lane builds its K-bit result and the 32-bit bitset word is assembled with
//     a DPP xor-butterfly OR over the 32/K lanes that share the word (quad_perm, quad_perm,
//     row_half_mirror, row_mirror): VALU only, no LDS, bit i of word w == row 32*w + i.
//   * Aggregates are per-lane registers over the grid-stride loop, then wave shuffle ->
//     LDS -> one partial per workgroup -> ordered merge in a finalize kernel.  No float
//     atomics anywhere: results are bitwise reproducible for a given grid.
//   * float sums are double-double (error-free TwoSum) so the rounded result is the exact
//     sum rounded once; integer sums wrap mod 2^64 exactly like Go.
#pragma once
#include "knox_kernels.h"
#include "knox_finalize.h"
#include <utility>

namespace knox {

// ---------------------------------------------------------------------------------------------
// width traits
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte global_load_dwordx4

template <int W> struct WT;
template <> struct WT<8> { using U = uint64_t; static constexpr int K = 2;  static constexpr U MASK = ~0ull;   static constexpr U SIGN = 1ull << 63; };
template <> struct WT<4> { using U = uint32_t; static constexpr int K = 4;  static constexpr U MASK = ~0u;     static constexpr U SIGN = 1u << 31; };
template <> struct WT<2> { using U = uint32_t; static constexpr int K = 8;  static constexpr U MASK = 0xffffu; static constexpr U SIGN = 1u << 15; };
template <> struct WT<1> { using U = uint32_t; static constexpr int K = 16; static constexpr U MASK = 0xffu;   static constexpr U SIGN = 1u << 7; };


typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

template <int W> __device__ __forceinline__ typename WT<W>::U elem(const u32x4 &r, int j) {
    const uint32_t d[4] = {r.x, r.y, r.z, r.w};
    if constexpr (W == 8) {
        // the loaded register pair as it is (composing (hi << 32) | lo makes the compiler split 64-bit
        // arithmetic into a zero-extended low part plus a shifted high part)
#ifdef KNOX_ELEM_COMPOSE  // diagnostics: the old form, for A/B runs
        return ((uint64_t)d[2 * j + 1] << 32) | d[2 * j];
#else
        return __builtin_bit_cast(u64x2, r)[j];
#endif
    } else if constexpr (W == 4) {
        return d[j];
    } else if constexpr (W == 2) {
        return (d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
    } else {
        return (d[j >> 2] >> (8 * (j & 3))) & 0xffu;
    }
}

// IEEE order key: monotone map of float bit patterns onto unsigned integers
// (-NaN < -Inf < ... < -0 < +0 < ... < +Inf < +NaN).
template <int W> __device__ __forceinline__ typename WT<W>::U fkey(typename WT<W>::U u) {
    if constexpr (W == 8) {
        return u ^ ((uint64_t)((int64_t)u >> 63) | (1ull << 63));
    } else if constexpr (W == 4) {
        return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
    } else {
        return u;
    }
}

template <bool NT> __device__ __forceinline__ u32x4 ld16(const u32x4 *p) {
    if constexpr (NT) {
        return __builtin_nontemporal_load(p);
    } else {
        return *p;
    }
}

// Diagnostics builds only (make diag): the scan kernel's column loads with an explicit gfx950 cache
// policy (sc0 / sc1 / nt bits), to separate kernel effects from HBM placement effects.
// The product build (KNOX_LOAD_POLICY undefined) uses the nontemporal builtin above.
#ifdef KNOX_LOAD_POLICY
#if KNOX_LOAD_POLICY == 1
#define KNOX_LD_BITS ""
#elif KNOX_LOAD_POLICY == 2
#define KNOX_LD_BITS " sc0 sc1"
#elif KNOX_LOAD_POLICY == 3
#define KNOX_LD_BITS " sc0 sc1 nt"
#elif KNOX_LOAD_POLICY == 4
#define KNOX_LD_BITS " sc1"
#elif KNOX_LOAD_POLICY == 5
#define KNOX_LD_BITS " sc1 nt"
#elif KNOX_LOAD_POLICY == 6
#define KNOX_LD_BITS " sc0"
#else
#define KNOX_LD_BITS " nt"
#endif
__device__ __forceinline__ u32x4 ld16_policy(const u32x4 *p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" KNOX_LD_BITS : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void ld_fence(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
#endif

// ---------------------------------------------------------------------------------------------
// bitset word assembly: lane holds q (K bits, rows K*lane .. K*lane+K-1 of the wave-load);
// the G = 32/K lanes sharing a 32-bit word OR their shifted pieces with a DPP butterfly.
// Needs all 64 lanes active (true in the streaming kernels).
// ---------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int K> __device__ __forceinline__ uint32_t pack_word(uint32_t q, int lane) {
    constexpr int G = 32 / K;
    uint32_t w = q << (K * (lane & (G - 1)));
    if constexpr (G >= 2) w |= dpp_mov<0xB1>(w);   // quad_perm [1,0,3,2]  lane ^ 1
    if constexpr (G >= 4) w |= dpp_mov<0x4E>(w);   // quad_perm [2,3,0,1]  lane ^ 2
    if constexpr (G >= 8) w |= dpp_mov<0x141>(w);  // row_half_mirror      lane -> 7 - lane (of 8)
    if constexpr (G >= 16) w |= dpp_mov<0x140>(w); // row_mirror           lane -> 15 - lane (of 16)
    return w;
}

// ---------------------------------------------------------------------------------------------
// aggregate accumulator
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dd_add(double &hi, double &lo, double h2, double l2) {
    // error-free TwoSum (Knuth); a non-finite running sum keeps its value, not NaN residue
    double s = hi + h2;
    double bb = s - hi;
    double e = (hi - (s - bb)) + (h2 - bb);
    if (!(__builtin_fabs(s) <= 1.7976931348623157e308)) e = 0.0;
    lo = lo + l2 + e;
    hi = s;
}

using knox_fin::int128_to_dd;   // (knox_finalize.h: shared with the sort-free window sums of knox_fold.hip)

struct Acc {
    uint64_t count, isum;
    double hi, lo;        // float columns: double-double sum of the selected values
    uint64_t xlo;         // integer columns: the EXACT 128-bit sum of the selected values (xhi:xlo, two's complement) --
    int64_t xhi;          // what Mean / Var need instead of a float64 running sum (finalize turns it into hi + lo)
    uint64_t minkey, minrow, maxkey, maxrow, firstrow, lastrow1;  // lastrow1 = row + 1, 0 = none
    __device__ __forceinline__ void add128(uint64_t v, int64_t vhi) {
        const uint64_t t = xlo + v;
        xhi += vhi + (int64_t)(t < v);
        xlo = t;
    }
    __device__ __forceinline__ void init() {
        count = 0; isum = 0; hi = 0.0; lo = 0.0; xlo = 0; xhi = 0;
        minkey = 0; minrow = kNoRow; maxkey = 0; maxrow = kNoRow; firstrow = kNoRow; lastrow1 = 0;
    }
    __device__ __forceinline__ void merge(const Acc &b) {
        count += b.count;
        isum += b.isum;
        dd_add(hi, lo, b.hi, b.lo);
        add128(b.xlo, b.xhi);
        if (b.minrow != kNoRow && (minrow == kNoRow || b.minkey < minkey || (b.minkey == minkey && b.minrow < minrow))) {
            minkey = b.minkey; minrow = b.minrow;
        }
        if (b.maxrow != kNoRow && (maxrow == kNoRow || b.maxkey > maxkey || (b.maxkey == maxkey && b.maxrow < maxrow))) {
            maxkey = b.maxkey; maxrow = b.maxrow;
        }
        firstrow = b.firstrow < firstrow ? b.firstrow : firstrow;
        lastrow1 = b.lastrow1 > lastrow1 ? b.lastrow1 : lastrow1;
    }
};

// the value sign-extended to 64 bit (two's complement)
template <int W> __device__ __forceinline__ uint64_t sext_u64(typename WT<W>::U e) {
    if constexpr (W == 8) return (uint64_t)e;
    else if constexpr (W == 4) return (uint64_t)(int64_t)(int32_t)e;
    else if constexpr (W == 2) return (uint64_t)(int64_t)(int16_t)e;
    else return (uint64_t)(int64_t)(int8_t)e;
}

// value -> float64 exactly as Go's float64(v) conversion of the column type
template <int W> __device__ __forceinline__ double to_f64(typename WT<W>::U e, uint32_t flavor) {
    if (flavor == FL_FLOAT) {
        if constexpr (W == 8) return __longlong_as_double((long long)e);
        else return (double)__uint_as_float((uint32_t)e);
    } else if (flavor == FL_SIGNED) {
        if constexpr (W == 8) return (double)(int64_t)e;
        else if constexpr (W == 4) return (double)(int32_t)e;
        else if constexpr (W == 2) return (double)(int16_t)e;
        else return (double)(int8_t)e;
    } else {
        if constexpr (W == 8) return (double)(uint64_t)e;
        else return (double)(uint32_t)e;
    }
}

// One selected-or-not row into the accumulator.  Rows arrive in increasing order per lane.
template <int W, unsigned AGG>
__device__ __forceinline__ void acc_update(Acc &acc, typename WT<W>::U e, uint32_t hit, uint64_t row, uint32_t flavor) {
    using U = typename WT<W>::U;
    if constexpr ((AGG & A_ISUM) != 0) {
        acc.isum += hit ? (uint64_t)e : 0ull;
    }
    if constexpr ((AGG & A_FSUM) != 0) {
        if (flavor == FL_FLOAT) {
            double x = to_f64<W>(e, flavor);
            x = hit ? x : 0.0;
            dd_add(acc.hi, acc.lo, x, 0.0);
        } else if (hit) {
            const uint64_t v = flavor == FL_SIGNED ? sext_u64<W>(e) : (uint64_t)e;
            acc.add128(v, flavor == FL_SIGNED ? (int64_t)v >> 63 : 0);
        }
    }
    if constexpr ((AGG & A_MINMAX) != 0) {
        // MinReducer / MaxReducer (reducer.go:256-296): strict compare, first occurrence wins;
        // a NaN is never adopted over a number (finalize handles "first value is NaN").
        const bool isf = flavor == FL_FLOAT;
        const U mag = e & (WT<W>::MASK ^ WT<W>::SIGN);
        U fk = fkey<W>(e);
        if (mag == 0) fk = WT<W>::SIGN;  // -0 and +0 compare equal: one key
        U expmask;
        if constexpr (W == 8) expmask = 0x7ff0000000000000ull; else expmask = 0x7f800000u;
        const bool isnan = isf && (mag > expmask);
        const U xs = (flavor == FL_SIGNED) ? WT<W>::SIGN : (U)0;
        const uint64_t okey = isf ? (uint64_t)fk : (uint64_t)(e ^ xs);
        const bool cand = hit && !isnan;
        if (cand && (acc.minrow == kNoRow || okey < acc.minkey)) { acc.minkey = okey; acc.minrow = row; }
        if (cand && (acc.maxrow == kNoRow || okey > acc.maxkey)) { acc.maxkey = okey; acc.maxrow = row; }
    }
    if constexpr ((AGG & A_FIRSTLAST) != 0) {
        if (hit) {
            if (acc.firstrow == kNoRow) acc.firstrow = row;
            acc.lastrow1 = row + 1;
        }
    }
}

// Wave-uniform bounds for the min/max fast path: a selected value can only become the wave's
// min (max) if its order key is below (above) the best key any lane of the wave holds.
struct WaveMM {
    uint64_t smin, smax;
    __device__ __forceinline__ void init() { smin = ~0ull; smax = 0ull; }
};

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// ---- compile-time value flavour helpers (FLV = FL_SIGNED / FL_UNSIGNED / FL_FLOAT) -------------
template <int W, unsigned FLV> __device__ __forceinline__ double to_f64_c(typename WT<W>::U e) {
    if constexpr (FLV == FL_FLOAT) {
        if constexpr (W == 8) return __longlong_as_double((long long)e);
        else return (double)__uint_as_float((uint32_t)e);
    } else if constexpr (FLV == FL_SIGNED) {
        if constexpr (W == 8) return (double)(int64_t)e;
        else if constexpr (W == 4) return (double)(int32_t)e;
        else if constexpr (W == 2) return (double)(int16_t)e;
        else return (double)(int8_t)e;
    } else {
        if constexpr (W == 8) return (double)(uint64_t)e;
        else return (double)(uint32_t)e;
    }
}
// raw order key (floats: a -0 / NaN key may trigger the exact path spuriously, never miss it)
template <int W, unsigned FLV> __device__ __forceinline__ uint64_t okey_raw(typename WT<W>::U e) {
    if constexpr (FLV == FL_FLOAT) return (uint64_t)fkey<W>(e);
    else if constexpr (FLV == FL_SIGNED) return (uint64_t)(e ^ WT<W>::SIGN);
    else return (uint64_t)e;
}

// first / last selected row of a lane, tracked per wave iteration: the lane's result bits of
// the kUnroll loads packed into one word + the iteration's first row; decoded once at the end.
struct FirstLast {
    uint64_t first_row0, last_row0, first_qq, last_qq;
    __device__ __forceinline__ void init() { first_row0 = 0; last_row0 = 0; first_qq = 0; last_qq = 0; }
    __device__ __forceinline__ void update(uint64_t row0, uint64_t qq) {
        if (qq != 0) {
            if (first_qq == 0) { first_qq = qq; first_row0 = row0; }
            last_qq = qq;
            last_row0 = row0;
        }
    }
    template <int K> __device__ __forceinline__ void finish(Acc &acc, uint64_t row_base, int lane) const {
        constexpr int RPL = 64 * K;
        if (first_qq != 0) {
            const int b = __builtin_ctzll(first_qq);
            const uint64_t r = row_base + first_row0 + (uint64_t)(b / K) * RPL + (uint64_t)lane * K + (uint64_t)(b % K);
            acc.firstrow = r < acc.firstrow ? r : acc.firstrow;
        }
        if (last_qq != 0) {
            const int b = 63 - __builtin_clzll(last_qq);
            const uint64_t r1 = row_base + last_row0 + (uint64_t)(b / K) * RPL + (uint64_t)lane * K + (uint64_t)(b % K) + 1;
            acc.lastrow1 = r1 > acc.lastrow1 ? r1 : acc.lastrow1;
        }
    }
};

// 4-byte columns: the same contract as agg_iter below, organised around ONE all-ones / zero mask per value
// (v_bfe_i32 of the selection bit) instead of a compare + select per use -- these instantiations are VALU bound:
//   xb = e & m                         the selected raw value or 0: feeds the wrapping sum, the exact sum and the maximum
//   sums (integers)                    Z = sum of xb (zero-extended, 64 bit); signed: exact sum = Z - 2^32 * #negatives
//   sums (float32)                     as in agg_iter: sixteen values add exactly in float64 when their exponents span <= 24
//   min / max pre-filter               v_min / v_max of bfi(m, e, neutral) in the column's OWN order (i32 / u32 / f32 --
//                                      v_min_f32 / v_max_f32 drop NaNs, which are never candidates); converted to order
//                                      keys once per iteration and compared with the wave's best keys
// (m & a) | (~m & b) as ONE instruction: written out, the compiler rebuilds the selection from the result bit with a
// shift + compare + select per use (3 instructions)
__device__ __forceinline__ uint32_t bfi_b32(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "s"(b));
    return d;
}

template <unsigned AGG, unsigned FLV>
__device__ __forceinline__ void agg_iter_w4(Acc &acc, WaveMM &wm, FirstLast &fl, const u32x4 (&r)[kUnroll],
                                            const uint32_t (&q)[kUnroll], uint64_t row0, uint64_t row_base, int lane) {
    constexpr int W = 4, K = 4, RPL = 256;
    static_assert(kUnroll * K <= 16, "the float exactness bound assumes at most 16 values per iteration");
    constexpr bool ISF = FLV == FL_FLOAT;
    constexpr bool ZSUM = (AGG & A_ISUM) != 0 || ((AGG & A_FSUM) != 0 && !ISF);
    uint64_t z = 0;
    uint32_t neg = 0;
    uint32_t amax = 0, amin = 0xffffffffu;
    double it = 0.0;
    int32_t smin = 0x7fffffff, smax = (int32_t)0x80000000;
    uint32_t umin = 0xffffffffu, umax = 0u;
    float fmn = __uint_as_float(0x7f800000u), fmx = __uint_as_float(0xff800000u);
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t e = (uint32_t)elem<W>(r[u], j);
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)q[u], j, 1);  // all ones when selected
            const uint32_t xb = e & m;
            if constexpr (ZSUM) z += xb;
            if constexpr ((AGG & A_FSUM) != 0 && FLV == FL_SIGNED) neg += xb >> 31;
            if constexpr ((AGG & A_FSUM) != 0 && ISF) {
                // smallest non-zero magnitude, as 2*|x| - 2 in one v_lshl_add (the sign bit leaves on the left; zeros and
                // unselected values wrap to the top and never win the minimum).  The largest magnitude comes from the
                // min / max pre-filter below when there is one, else it is tracked here.
                const uint32_t w2 = (xb << 1) + 0xfffffffeu;
                amin = w2 < amin ? w2 : amin;
                if constexpr ((AGG & A_MINMAX) == 0) {
                    const uint32_t mag = xb & 0x7fffffffu;
                    amax = mag > amax ? mag : amax;
                }
                it += (double)__uint_as_float(xb);
            }
            if constexpr ((AGG & A_MINMAX) != 0) {
                if constexpr (FLV == FL_SIGNED) {
                    const int32_t lo = (int32_t)bfi_b32(m, e, 0x7fffffffu), hi = (int32_t)bfi_b32(m, e, 0x80000000u);
                    smin = lo < smin ? lo : smin;
                    smax = hi > smax ? hi : smax;
                } else if constexpr (FLV == FL_UNSIGNED) {
                    const uint32_t lo = bfi_b32(m, e, 0xffffffffu);
                    umin = lo < umin ? lo : umin;
                    umax = xb > umax ? xb : umax;
                } else {
                    fmn = fminf(fmn, __uint_as_float(bfi_b32(m, e, 0x7f800000u)));
                    fmx = fmaxf(fmx, __uint_as_float(bfi_b32(m, e, 0xff800000u)));
                }
            }
        }
    }
    if constexpr ((AGG & A_ISUM) != 0) acc.isum += z;
    if constexpr ((AGG & A_FSUM) != 0 && !ISF) {
        if constexpr (FLV == FL_UNSIGNED) {
            acc.add128(z, 0);
        } else {
            const uint64_t sv = z - ((uint64_t)neg << 32);  // sign-extended sum: |.| < 2^36
            acc.add128(sv, (int64_t)sv >> 63);
        }
    }
    if constexpr ((AGG & A_FSUM) != 0 && ISF) {
        // zeros are ignored by the span test; Inf takes the per-value path (a NaN makes the sum NaN on either path).  With
        // a min / max pre-filter the largest magnitude is the larger of |min| and |max| of the selected values
        if constexpr ((AGG & A_MINMAX) != 0) {
            const uint32_t a0 = __float_as_uint(fmn) & 0x7fffffffu, a1 = __float_as_uint(fmx) & 0x7fffffffu;
            amax = a0 > a1 ? a0 : a1;
        }
        const uint32_t mmin = (amin >> 1) + 1u;  // the smallest non-zero magnitude (amin = 2*mag - 2)
        const bool ok = amin >= 0xfffffffeu || (amax < 0x7f800000u && ((amax >> 23) - (mmin >> 23)) <= 24u);
        if (__all(ok)) {
            const double s = acc.hi + it;
            const double bb = s - acc.hi;
            acc.lo += (acc.hi - (s - bb)) + (it - bb);
            acc.hi = s;
        } else {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    double x = to_f64_c<W, FLV>(elem<W>(r[u], j));
                    x = ((q[u] >> j) & 1u) ? x : 0.0;
                    const double s = acc.hi + x;
                    const double bb = s - acc.hi;
                    acc.lo += (acc.hi - (s - bb)) + (x - bb);
                    acc.hi = s;
                }
            }
        }
    }
    if constexpr ((AGG & A_MINMAX) != 0) {
        uint32_t imin, imax;
        if constexpr (FLV == FL_SIGNED) {
            imin = (uint32_t)smin ^ 0x80000000u;
            imax = (uint32_t)smax ^ 0x80000000u;
        } else if constexpr (FLV == FL_UNSIGNED) {
            imin = umin;
            imax = umax;
        } else {
            imin = (uint32_t)fkey<W>(__float_as_uint(fmn));
            imax = (uint32_t)fkey<W>(__float_as_uint(fmx));
        }
        // (a lane that selected nothing holds the neutral elements, which compare "better" than the wave's initial bounds:
        //  without the `any` term every iteration of a scan that selects nothing -- 0 % selectivity -- took the exact path)
        const bool any = (q[0] | q[1] | q[2] | q[3]) != 0;
        static_assert(kUnroll == 4, "the `any` term above names four loads");
        if (__any(any && ((uint64_t)imin < wm.smin || (uint64_t)imax > wm.smax))) {
            // the rare exact path walks the four loads in a rolled loop (code size) over COPIES of them: indexed in place, the loop
            // variable made the compiler keep the whole r[NL][4] block of the three- and four-leaf instantiations in scratch memory,
            // twelve / sixteen 16-byte scratch stores in EVERY iteration (knox_scan_w*.res: ScratchSize 208 / 272); the copies are
            // made -- and, if at all, spilled -- only here
            u32x4 rr[kUnroll];
            uint32_t qr[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) { rr[u] = r[u]; qr[u] = q[u]; }
#pragma unroll 1
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t rbase = row_base + row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
#pragma unroll
                for (int j = 0; j < K; j++) acc_update<W, A_MINMAX>(acc, elem<W>(rr[u], j), (qr[u] >> j) & 1u, rbase + j, FLV);
            }
            wm.smin = uniform_u64(wave_min_u64(acc.minrow != kNoRow ? acc.minkey : ~0ull));
            wm.smax = uniform_u64(wave_max_u64(acc.maxrow != kNoRow ? acc.maxkey : 0ull));
        }
    }
    if constexpr ((AGG & A_FIRSTLAST) != 0) {
        uint64_t qq = 0;
#pragma unroll
        for (int u = 0; u < kUnroll; u++) qq |= (uint64_t)q[u] << (u * K);
        fl.update(row0, qq);
    }
}

// One wave iteration (kUnroll 16-byte loads of this lane, selection bits q[u]) into the
// accumulator.  Same results as one acc_update per row, organised so the common case is cheap:
//   * sums: error-free TwoSum per selected value, no branch (a non-finite running sum keeps an
//     IEEE-correct `hi`; its `lo` is ignored by every consumer);
//   * min/max: one compare pair per value against the WAVE-wide best keys; only when some lane
//     beats them (O(log n) times per wave) does the exact per-value update run;
//   * first/last: FirstLast::update once per iteration.
template <int W, unsigned AGG, unsigned FLV>
__device__ __forceinline__ void agg_iter(Acc &acc, WaveMM &wm, FirstLast &fl, const u32x4 (&r)[kUnroll],
                                         const uint32_t (&q)[kUnroll], uint64_t row0, uint64_t row_base, int lane) {
    constexpr int K = WT<W>::K;
    constexpr int RPL = 64 * K;
    if constexpr (W == 4) {
        agg_iter_w4<AGG, FLV>(acc, wm, fl, r, q, row0, row_base, lane);
        return;
    }
    uint64_t it_z = 0;  // this iteration's sum of the selected raw (zero-extended) values
    if constexpr ((AGG & A_ISUM) != 0) {
        if constexpr (W <= 2) {  // <= 64 values below 2^16: a 32-bit partial sum cannot overflow
            uint32_t t = 0;
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) t += ((q[u] >> j) & 1u) ? (uint32_t)elem<W>(r[u], j) : 0u;
            }
            it_z = t;
        } else {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) it_z += ((q[u] >> j) & 1u) ? (uint64_t)elem<W>(r[u], j) : 0ull;
            }
        }
        acc.isum += it_z;
    }
    if constexpr ((AGG & A_FSUM) != 0 && FLV == FL_FLOAT) {
        bool done = false;
        if constexpr (W == 4) {
            // float32: the 16 selected values of one iteration add up EXACTLY in float64 when their exponents span at
            // most 24 binades (24-bit significands, 4 carry bits: every partial sum needs <= 52 bits), which is what
            // real columns look like -- then one error-free TwoSum per iteration replaces sixteen.  Zeros are ignored
            // by the test (min over mag - 1 wraps them to the top); Inf / NaN take the per-value path below.
            static_assert(kUnroll * K <= 16, "the exactness bound below assumes at most 16 values per iteration");
            uint32_t amax = 0, amin = 0xffffffffu;
            double it = 0.0;
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const uint32_t xb = ((q[u] >> j) & 1u) ? (uint32_t)elem<W>(r[u], j) : 0u;
                    const uint32_t mag = xb & 0x7fffffffu;
                    amax = mag > amax ? mag : amax;
                    amin = (mag - 1u) < amin ? (mag - 1u) : amin;
                    it += (double)__uint_as_float(xb);
                }
            }
            const bool ok = amax < 0x7f800000u && (amin == 0xffffffffu || ((amax >> 23) - ((amin + 1u) >> 23)) <= 24u);
            if (__all(ok)) {
                const double s = acc.hi + it;
                const double bb = s - acc.hi;
                acc.lo += (acc.hi - (s - bb)) + (it - bb);
                acc.hi = s;
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    double x = to_f64_c<W, FLV>(elem<W>(r[u], j));
                    x = ((q[u] >> j) & 1u) ? x : 0.0;
                    const double s = acc.hi + x;
                    const double bb = s - acc.hi;
                    acc.lo += (acc.hi - (s - bb)) + (x - bb);
                    acc.hi = s;
                }
            }
        }
    }
    if constexpr ((AGG & A_FSUM) != 0 && FLV != FL_FLOAT) {
        // integers: the exact sum.  8-byte values carry into the high word one by one; narrower values cannot
        // overflow 64 bits within one iteration (<= 64 values), so the iteration's sum is added once
        if constexpr (W == 8) {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const uint64_t v = ((q[u] >> j) & 1u) ? (uint64_t)elem<W>(r[u], j) : 0ull;
                    acc.add128(v, FLV == FL_SIGNED ? (int64_t)v >> 63 : 0);
                }
            }
        } else if constexpr (FLV == FL_UNSIGNED && (AGG & A_ISUM) != 0) {
            acc.add128(it_z, 0);  // unsigned: the exact sum is the raw sum, already formed above
        } else {
            uint64_t it = 0;
            if constexpr (W <= 2) {  // as above: 64 values of at most 16 bits fit an int32 partial sum
                int32_t t = 0;
#pragma unroll
                for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        const int32_t v = FLV == FL_SIGNED ? (int32_t)(int64_t)sext_u64<W>(elem<W>(r[u], j)) : (int32_t)elem<W>(r[u], j);
                        t += ((q[u] >> j) & 1u) ? v : 0;
                    }
                }
                it = (uint64_t)(int64_t)t;
            } else {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        const uint64_t v = FLV == FL_SIGNED ? sext_u64<W>(elem<W>(r[u], j)) : (uint64_t)elem<W>(r[u], j);
                        it += ((q[u] >> j) & 1u) ? v : 0ull;
                    }
                }
            }
            acc.add128(it, FLV == FL_SIGNED ? (int64_t)it >> 63 : 0);
        }
    }
    if constexpr ((AGG & A_MINMAX) != 0) {
        bool trig = false;
        if constexpr (W < 8) {
            // keys fit 32 bits: the iteration's min / max over the selected values (unselected: the neutral element)
            // with v_min / v_max, then ONE compare pair -- no per-value 64-bit compares and mask arithmetic
            uint32_t imin = 0xffffffffu, imax = 0u;
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const uint32_t key = (uint32_t)okey_raw<W, FLV>(elem<W>(r[u], j));
                    const bool sel = ((q[u] >> j) & 1u) != 0;
                    const uint32_t kmin = sel ? key : 0xffffffffu, kmax = sel ? key : 0u;
                    imin = kmin < imin ? kmin : imin;
                    imax = kmax > imax ? kmax : imax;
                }
            }
            trig = (q[0] | q[1] | q[2] | q[3]) != 0 && ((uint64_t)imin < wm.smin || (uint64_t)imax > wm.smax);  // nothing selected: nothing to adopt
        } else {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const uint64_t key = okey_raw<W, FLV>(elem<W>(r[u], j));
                    trig = trig || (((q[u] >> j) & 1u) && (key < wm.smin || key > wm.smax));
                }
            }
        }
        if (__any(trig)) {
            // the rare exact path walks the four loads in a rolled loop (code size) over COPIES of them: indexed in place, the loop
            // variable made the compiler keep the whole r[NL][4] block of the three- and four-leaf instantiations in scratch memory,
            // twelve / sixteen 16-byte scratch stores in EVERY iteration (knox_scan_w*.res: ScratchSize 208 / 272); the copies are
            // made -- and, if at all, spilled -- only here
            u32x4 rr[kUnroll];
            uint32_t qr[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) { rr[u] = r[u]; qr[u] = q[u]; }
#pragma unroll 1
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t rbase = row_base + row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
#pragma unroll
                for (int j = 0; j < K; j++) acc_update<W, A_MINMAX>(acc, elem<W>(rr[u], j), (qr[u] >> j) & 1u, rbase + j, FLV);
            }
            wm.smin = uniform_u64(wave_min_u64(acc.minrow != kNoRow ? acc.minkey : ~0ull));
            wm.smax = uniform_u64(wave_max_u64(acc.maxrow != kNoRow ? acc.maxkey : 0ull));
        }
    }
    if constexpr ((AGG & A_FIRSTLAST) != 0) {
        uint64_t qq = 0;
#pragma unroll
        for (int u = 0; u < kUnroll; u++) qq |= (uint64_t)q[u] << (u * K);
        fl.update(row0, qq);
    }
}

__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v, int off) {
    return (uint64_t)__shfl_down((unsigned long long)v, off, 64);
}
__device__ __forceinline__ double shfl_down_f64(double v, int off) { return __shfl_down(v, off, 64); }

__device__ __forceinline__ void wave_reduce(Acc &a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Acc b;
        b.count = shfl_down_u64(a.count, off);
        b.isum = shfl_down_u64(a.isum, off);
        b.hi = shfl_down_f64(a.hi, off);
        b.lo = shfl_down_f64(a.lo, off);
        b.xlo = shfl_down_u64(a.xlo, off);
        b.xhi = (int64_t)shfl_down_u64((uint64_t)a.xhi, off);
        b.minkey = shfl_down_u64(a.minkey, off);
        b.minrow = shfl_down_u64(a.minrow, off);
        b.maxkey = shfl_down_u64(a.maxkey, off);
        b.maxrow = shfl_down_u64(a.maxrow, off);
        b.firstrow = shfl_down_u64(a.firstrow, off);
        b.lastrow1 = shfl_down_u64(a.lastrow1, off);
        a.merge(b);
    }
}

// wave shuffle -> LDS -> thread 0 writes the workgroup partial
__device__ __forceinline__ void block_reduce_store(Acc &acc, AggPartial *out) {
    __shared__ Acc sm[kBlock / 64];
    wave_reduce(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Acc t = sm[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) t.merge(sm[w]);
        AggPartial p;
        p.count = t.count; p.isum = t.isum; p.hi = t.hi; p.lo = t.lo; p.xlo = t.xlo; p.xhi = t.xhi;
        p.minkey = t.minkey; p.minrow = t.minrow; p.maxkey = t.maxkey; p.maxrow = t.maxrow;
        p.firstrow = t.firstrow; p.lastrow = t.lastrow1;
        *out = p;
    }
}

// ---------------------------------------------------------------------------------------------
// streaming scan kernel
// ---------------------------------------------------------------------------------------------
// ANYF: at least one leaf column is a float (its values go through the IEEE order key);
// integer-only instantiations carry no key code at all.
// ---------------------------------------------------------------------------------------------
// 1- and 2-byte columns, one leaf: SWAR evaluation of a whole 32-bit word of packed values (4 x u8 or
// 2 x u16) -- element-at-a-time extraction makes these widths VALU bound (62 % / 77 % of HBM peak).
//   d   = x - lo per element:            ((x | H) - (lo & L)) ^ ((x ^ ~lo) & H)      (no borrow crosses
//   hit = d <= diff per element (top bit): (D & ~d) | (~(D ^ d) & ((D | H) - (d & L)))   an element)
// (Hacker's Delight 2-18; both checked exhaustively over all byte pairs / 2e5 random half-word pairs.)
// hit / negated / constant is applied on the top bits, which then feed the bitset gather and, as 0/1
// element masks, a v_dot4_u32_u8 / v_dot2_u32_u16 for the wrapping integer sum.
// ---------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int W> struct Swar {
    static constexpr uint32_t H = W == 1 ? 0x80808080u : 0x80008000u;
    static constexpr uint32_t L = ~H;
    uint32_t A, NLO, D4, C1, M1H, M0H;
    __device__ __forceinline__ void init(uint32_t lo, uint32_t diff, bool m1, bool m0) {
        constexpr uint32_t vm = W == 1 ? 0xffu : 0xffffu, rep = W == 1 ? 0x01010101u : 0x00010001u;
        const uint32_t lo4 = (lo & vm) * rep;
        A = lo4 & L;
        NLO = ~lo4;
        D4 = (diff > vm ? vm : diff) * rep;
        C1 = D4 | H;
        M1H = m1 ? H : 0u;
        M0H = m0 ? H : 0u;
    }
    // top bit of every element = final hit, every other bit 0
    __device__ __forceinline__ uint32_t hits(uint32_t x) const {
        const uint32_t t = (x | H) - A;
        const uint32_t dd = t ^ ((x ^ NLO) & H);
        const uint32_t t2 = C1 - (dd & L);
        const uint32_t res = (D4 & ~dd) | (~(D4 ^ dd) & t2);
        return (res & M1H) | (~res & M0H);
    }
};

// one 16-byte load of a 1- / 2-byte column: K result bits in row order, masked sum added to sum32
template <int W, bool SUM> __device__ __forceinline__ uint32_t swar_load(const Swar<W> &sw, const u32x4 &r, uint32_t &sum32) {
    const uint32_t d[4] = {r.x, r.y, r.z, r.w};
    uint32_t qq = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t g = sw.hits(d[w]);
        if constexpr (W == 1) {
            const uint32_t m = g >> 7;  // 0 / 1 per byte
            if constexpr (SUM) sum32 = __builtin_amdgcn_udot4(d[w], m, sum32, false);
            qq |= ((m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xfu) << (4 * w);
        } else {
            const uint32_t m = g >> 15;  // 0 / 1 per half word
            if constexpr (SUM) sum32 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d[w]), __builtin_bit_cast(u16x2, m), sum32, false);
            qq |= ((m | (m >> 15)) & 0x3u) << (2 * w);
        }
    }
    return qq;
}

// The same contract as swar_load with the wave's compare unit instead of SWAR arithmetic (measured against it with
// -DKNOX_NARROW_SWAR): per element ONE subtract that wraps at the element width (2-byte: v_pk_sub_u16 does both halves of
// a dword at once; 1-byte: v_sub_u32_sdwa picks the byte, subtracts and keeps the low 8 bits), ONE compare against the
// range width straight into an SGPR pair (SDWA selects the half word) and ONE add-with-carry that shifts the lane's bit
// of that mask into the row-order word, highest element first: qq = 2 qq + hit.  2.5 (2-byte) / 3 (1-byte) VALU
// operations per row where the SWAR form takes 8 / 5.5; the selected values' sum takes the 0/1 element masks
// v_dot2 / v_dot4 want from the result bits with one 24-bit multiply (the partial products do not collide) and one AND.
// lo_rep: the range's low end, in both halves of the dword (2-byte) / in the low byte (1-byte); vdiff: the range width,
// clamped to the element's range.
template <int W, bool SUM>
__device__ __forceinline__ uint32_t narrow_load(const u32x4 &r, uint32_t lo_rep, uint32_t vdiff, uint32_t m1, uint32_t m0, uint32_t &sum32) {
    static_assert(W == 1 || W == 2, "1- and 2-byte columns");
    const uint32_t d[4] = {r.x, r.y, r.z, r.w};
    uint32_t qq = 0;
#pragma unroll
    for (int w = 3; w >= 0; w--) {
        if constexpr (W == 2) {
            uint32_t t;
            uint64_t m;
            asm("v_pk_sub_u16 %0, %1, %2" : "=v"(t) : "v"(d[w]), "v"(lo_rep));
            asm("v_cmp_le_u16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_0" : "=s"(m) : "v"(t), "v"(vdiff));
            asm("v_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(qq), "+s"(m));
            asm("v_cmp_le_u16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(m) : "v"(t), "v"(vdiff));
            asm("v_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(qq), "+s"(m));
        } else {
#define KNOX_BYTE_STEP(SEL)                                                                                                                        \
    {                                                                                                                                              \
        uint32_t t;                                                                                                                                \
        uint64_t m;                                                                                                                                \
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:" SEL " src1_sel:DWORD" : "=v"(t) : "v"(d[w]), "v"(lo_rep)); \
        asm("v_cmp_le_u32_e64 %0, %1, %2" : "=s"(m) : "v"(t), "v"(vdiff));                                                                        \
        asm("v_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(qq), "+s"(m));                                                                          \
    }
            KNOX_BYTE_STEP("BYTE_3") KNOX_BYTE_STEP("BYTE_2") KNOX_BYTE_STEP("BYTE_1") KNOX_BYTE_STEP("BYTE_0")
#undef KNOX_BYTE_STEP
        }
    }
    qq = (qq & m1) | (~qq & m0);  // hit / negated / constant (m1, m0: all ones or zero over the K result bits)
    if constexpr (SUM) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if constexpr (W == 1) {
                // bits b0..b3 of the nibble -> bytes 0x01 at 0, 8, 16, 24: nib * (1 + 2^7 + 2^14 + 2^21), all 16 partial products distinct
                const uint32_t m4 = __umul24((qq >> (4 * w)) & 0xfu, 0x00204081u) & 0x01010101u;
                sum32 = __builtin_amdgcn_udot4(d[w], m4, sum32, false);
            } else {
                const uint32_t m2 = __umul24((qq >> (2 * w)) & 0x3u, 0x00008001u) & 0x00010001u;
                sum32 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d[w]), __builtin_bit_cast(u16x2, m2), sum32, false);
            }
        }
    }
    return qq;
}

// One wave iteration of a 1- / 2-byte column with ANY aggregate class, whole 32-bit words at a time (the per-value
// form above costs ~14 VALU operations per value: 24 % / 42 % of the HBM peak for int8 / int16 with all aggregates).
// Per word of 4 (2) values:
//   hit bits        Swar::hits (top bit per element), row-order result bits as in swar_load
//   raw masked sum  v_dot4_u32_u8 / v_dot2_u32_u16 of the word with the 0/1 element masks (the wrapping integer sum; for
//                   unsigned columns also the exact sum)
//   signed exact    sum of sign-extended values = raw sum - 2^(8W) * #(selected values with the sign bit set): one
//                   AND + popcount per word
//   min / max       order key = word ^ sign bits; unselected elements -> 0 through the element-wide mask
//                   (g - m) | g; the max runs on the keys, the min on their complement, both as packed 16-bit
//                   maxima (v_pk_max_u16; bytes are widened two at a time with v_perm_b32); after the iteration ONE
//                   compare pair against the wave-wide best keys decides whether the exact per-value update (with
//                   row ids, O(log n) times per wave) has to run -- the same contract as agg_iter
// Results are identical to agg_iter (tests/test_gpu_scan.py: all types x all aggregates against the oracle).
template <int W, unsigned AGG, unsigned FLV>
__device__ __forceinline__ void agg_iter_swar(Acc &acc, WaveMM &wm, FirstLast &fl, const Swar<W> &sw, const u32x4 (&r)[kUnroll],
                                              uint32_t (&q)[kUnroll], uint32_t &cnt32, uint64_t row0, uint64_t row_base, int lane) {
    static_assert(W <= 2 && FLV != FL_FLOAT, "SWAR aggregates: 1- and 2-byte integer columns");
    constexpr int K = WT<W>::K;
    constexpr int RPL = 64 * K;
    constexpr uint32_t H = Swar<W>::H;
    constexpr uint32_t XS = FLV == FL_SIGNED ? H : 0u;
    uint32_t sum32 = 0, neg = 0;
    u16x2 pmax = {0, 0}, pminc = {0, 0};
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        const uint32_t d[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
        uint32_t qq = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t g = sw.hits(d[w]);
            const uint32_t m = g >> (8 * W - 1);
            if constexpr (W == 1) {
                if constexpr ((AGG & (A_ISUM | A_FSUM)) != 0) sum32 = __builtin_amdgcn_udot4(d[w], m, sum32, false);
                qq |= ((m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xfu) << (4 * w);
            } else {
                if constexpr ((AGG & (A_ISUM | A_FSUM)) != 0)
                    sum32 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d[w]), __builtin_bit_cast(u16x2, m), sum32, false);
                qq |= ((m | (m >> 15)) & 0x3u) << (2 * w);
            }
            if constexpr ((AGG & A_FSUM) != 0 && FLV == FL_SIGNED) neg += (uint32_t)__builtin_popcount(d[w] & g);
            if constexpr ((AGG & A_MINMAX) != 0) {
                // element-wide mask of the selected elements: 2-byte elements by one full-rate 24-bit multiply (m <= 0x00010001)
                const uint32_t mfull = W == 2 ? __umul24(m, 0xffffu) : ((g - m) | g);
                const uint32_t key = d[w] ^ XS;
                const uint32_t kmax = key & mfull, kminc = kmax ^ mfull;  // = ~key & mfull
                if constexpr (W == 2) {
                    pmax = __builtin_elementwise_max(pmax, __builtin_bit_cast(u16x2, kmax));
                    pminc = __builtin_elementwise_max(pminc, __builtin_bit_cast(u16x2, kminc));
                } else {
                    // bytes 0,1 and 2,3 zero-extended to 16-bit lanes: selector 0x0c = constant 0x00
                    pmax = __builtin_elementwise_max(pmax, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kmax, kmax, 0x0c010c00u)));
                    pmax = __builtin_elementwise_max(pmax, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kmax, kmax, 0x0c030c02u)));
                    pminc = __builtin_elementwise_max(pminc, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kminc, kminc, 0x0c010c00u)));
                    pminc = __builtin_elementwise_max(pminc, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kminc, kminc, 0x0c030c02u)));
                }
            }
        }
        q[u] = qq;
        cnt32 += (uint32_t)__builtin_popcount(qq);
    }
    if constexpr ((AGG & A_ISUM) != 0) acc.isum += sum32;  // the raw (zero-extended) values: SumReducer mod 2^64 is formed by finalize
    if constexpr ((AGG & A_FSUM) != 0) {
        if constexpr (FLV == FL_UNSIGNED) {
            acc.add128(sum32, 0);
        } else {
            const int64_t it = (int64_t)sum32 - ((int64_t)neg << (8 * W));
            acc.add128((uint64_t)it, it >> 63);
        }
    }
    if constexpr ((AGG & A_MINMAX) != 0) {
        constexpr uint32_t EMAX = W == 1 ? 0xffu : 0xffffu;
        const uint32_t imax = pmax.x > pmax.y ? pmax.x : pmax.y;
        const uint32_t iminc = pminc.x > pminc.y ? pminc.x : pminc.y;
        const uint32_t imin = EMAX - iminc;  // nothing selected: iminc = 0 -> EMAX, the neutral element (a selected EMAX is no news either)
        // okey_raw of these types IS the W-byte key zero-extended: comparable with the wave-wide 64-bit bounds; a lane
        // that selected nothing has nothing to adopt (its neutral elements would "beat" the initial bounds)
        const bool any = (q[0] | q[1] | q[2] | q[3]) != 0;
        if (__any(any && ((uint64_t)imin < wm.smin || (uint64_t)imax > wm.smax))) {
            // the rare exact path walks the four loads in a rolled loop (code size) over COPIES of them: indexed in place, the loop
            // variable made the compiler keep the whole r[NL][4] block of the three- and four-leaf instantiations in scratch memory,
            // twelve / sixteen 16-byte scratch stores in EVERY iteration (knox_scan_w*.res: ScratchSize 208 / 272); the copies are
            // made -- and, if at all, spilled -- only here
            u32x4 rr[kUnroll];
            uint32_t qr[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) { rr[u] = r[u]; qr[u] = q[u]; }
#pragma unroll 1
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t rbase = row_base + row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
#pragma unroll
                for (int j = 0; j < K; j++) acc_update<W, A_MINMAX>(acc, elem<W>(rr[u], j), (qr[u] >> j) & 1u, rbase + j, FLV);
            }
            wm.smin = uniform_u64(wave_min_u64(acc.minrow != kNoRow ? acc.minkey : ~0ull));
            wm.smax = uniform_u64(wave_max_u64(acc.maxrow != kNoRow ? acc.maxkey : 0ull));
        }
    }
    if constexpr ((AGG & A_FIRSTLAST) != 0) {
        uint64_t qq = 0;
#pragma unroll
        for (int u = 0; u < kUnroll; u++) qq |= (uint64_t)q[u] << (u * K);
        fl.update(row0, qq);
    }
}

// agg_iter_swar with narrow_load's predicate (round 3): the same sums / min-max pre-filter / first-last bookkeeping, fed by the
// result bits instead of SWAR hit masks.  Results are identical (tests/test_gpu_scan.py: all types x all aggregates).
template <int W, unsigned AGG, unsigned FLV>
__device__ __forceinline__ void agg_iter_narrow(Acc &acc, WaveMM &wm, FirstLast &fl, uint32_t nlo, uint32_t ndiff, uint32_t m1, uint32_t m0, const u32x4 (&r)[kUnroll],
                                              uint32_t (&q)[kUnroll], uint32_t &cnt32, uint64_t row0, uint64_t row_base, int lane) {
    static_assert(W <= 2 && FLV != FL_FLOAT, "SWAR aggregates: 1- and 2-byte integer columns");
    constexpr int K = WT<W>::K;
    constexpr int RPL = 64 * K;
    constexpr uint32_t H = Swar<W>::H;
    constexpr uint32_t XS = FLV == FL_SIGNED ? H : 0u;
    uint32_t sum32 = 0;
    u16x2 pmax = {0, 0}, pminc = {0, 0};
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
        const uint32_t d[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
        // the predicate through the compare unit (narrow_load), then the 0/1 element masks the sums and the min / max
        // pre-filter want from the result bits: one 24-bit multiply + AND per word
        uint32_t unused = 0;
        const uint32_t qq = narrow_load<W, false>(r[u], nlo, ndiff, m1, m0, unused);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t m;
            if constexpr (W == 1) m = __umul24((qq >> (4 * w)) & 0xfu, 0x00204081u) & 0x01010101u;
            else m = __umul24((qq >> (2 * w)) & 0x3u, 0x00008001u) & 0x00010001u;
            const uint32_t g = m << (8 * W - 1);
            // signed columns: the signed dot product IS the exact sum of the selected values (64 values of <= 16 bits fit an
            // int32), and the wrapping sum is that sum mod 2^(8W) -- no count of negative values (round 4)
            if constexpr ((AGG & (A_ISUM | A_FSUM)) != 0) {
                if constexpr (W == 1 && FLV == FL_SIGNED) sum32 = (uint32_t)__builtin_amdgcn_sdot4((int)d[w], (int)m, (int)sum32, false);
                else if constexpr (W == 1) sum32 = __builtin_amdgcn_udot4(d[w], m, sum32, false);
                else if constexpr (FLV == FL_SIGNED) sum32 = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, d[w]), __builtin_bit_cast(s16x2, m), (int)sum32, false);
                else sum32 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d[w]), __builtin_bit_cast(u16x2, m), sum32, false);
            }
            if constexpr ((AGG & A_MINMAX) != 0) {
                // element-wide mask of the selected elements: 2-byte elements by one full-rate 24-bit multiply (m <= 0x00010001)
                const uint32_t mfull = W == 2 ? __umul24(m, 0xffffu) : ((g - m) | g);
                const uint32_t key = d[w] ^ XS;
                const uint32_t kmax = key & mfull, kminc = kmax ^ mfull;  // = ~key & mfull
                if constexpr (W == 2) {
                    pmax = __builtin_elementwise_max(pmax, __builtin_bit_cast(u16x2, kmax));
                    pminc = __builtin_elementwise_max(pminc, __builtin_bit_cast(u16x2, kminc));
                } else {
                    // bytes 0,1 and 2,3 zero-extended to 16-bit lanes: selector 0x0c = constant 0x00
                    pmax = __builtin_elementwise_max(pmax, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kmax, kmax, 0x0c010c00u)));
                    pmax = __builtin_elementwise_max(pmax, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kmax, kmax, 0x0c030c02u)));
                    pminc = __builtin_elementwise_max(pminc, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kminc, kminc, 0x0c010c00u)));
                    pminc = __builtin_elementwise_max(pminc, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(kminc, kminc, 0x0c030c02u)));
                }
            }
        }
        q[u] = qq;
        cnt32 += (uint32_t)__builtin_popcount(qq);
    }
    // the wrapping sum: finalize keeps the low 8W bits (knox_finalize.h: `isum & m`), where the zero-extended and the signed sum agree
    if constexpr ((AGG & A_ISUM) != 0) acc.isum += FLV == FL_SIGNED ? (uint64_t)(int64_t)(int32_t)sum32 : (uint64_t)sum32;
    if constexpr ((AGG & A_FSUM) != 0) {
        if constexpr (FLV == FL_UNSIGNED) {
            acc.add128(sum32, 0);
        } else {
            const int64_t it = (int64_t)(int32_t)sum32;
            acc.add128((uint64_t)it, it >> 63);
        }
    }
    if constexpr ((AGG & A_MINMAX) != 0) {
        constexpr uint32_t EMAX = W == 1 ? 0xffu : 0xffffu;
        const uint32_t imax = pmax.x > pmax.y ? pmax.x : pmax.y;
        const uint32_t iminc = pminc.x > pminc.y ? pminc.x : pminc.y;
        const uint32_t imin = EMAX - iminc;  // nothing selected: iminc = 0 -> EMAX, the neutral element (a selected EMAX is no news either)
        // okey_raw of these types IS the W-byte key zero-extended: comparable with the wave-wide 64-bit bounds; a lane
        // that selected nothing has nothing to adopt (its neutral elements would "beat" the initial bounds)
        const bool any = (q[0] | q[1] | q[2] | q[3]) != 0;
        if (__any(any && ((uint64_t)imin < wm.smin || (uint64_t)imax > wm.smax))) {
            // the rare exact path walks the four loads in a rolled loop (code size) over COPIES of them: indexed in place, the loop
            // variable made the compiler keep the whole r[NL][4] block of the three- and four-leaf instantiations in scratch memory,
            // twelve / sixteen 16-byte scratch stores in EVERY iteration (knox_scan_w*.res: ScratchSize 208 / 272); the copies are
            // made -- and, if at all, spilled -- only here
            u32x4 rr[kUnroll];
            uint32_t qr[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) { rr[u] = r[u]; qr[u] = q[u]; }
#pragma unroll 1
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t rbase = row_base + row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
#pragma unroll
                for (int j = 0; j < K; j++) acc_update<W, A_MINMAX>(acc, elem<W>(rr[u], j), (qr[u] >> j) & 1u, rbase + j, FLV);
            }
            wm.smin = uniform_u64(wave_min_u64(acc.minrow != kNoRow ? acc.minkey : ~0ull));
            wm.smax = uniform_u64(wave_max_u64(acc.maxrow != kNoRow ? acc.maxkey : 0ull));
        }
    }
    if constexpr ((AGG & A_FIRSTLAST) != 0) {
        uint64_t qq = 0;
#pragma unroll
        for (int u = 0; u < kUnroll; u++) qq |= (uint64_t)q[u] << (u * K);
        fl.update(row0, qq);
    }
}


// one element of a column by row (tail kernels, finalisers, gathers)
template <int W> __device__ __forceinline__ typename WT<W>::U load_elem(const void *col, uint64_t row) {
    if constexpr (W == 8) return ((const uint64_t *)col)[row];
    else if constexpr (W == 4) return ((const uint32_t *)col)[row];
    else if constexpr (W == 2) return ((const uint16_t *)col)[row];
    else return ((const uint8_t *)col)[row];
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
#define KNOX_LAUNCH(KERN, GRID, ...)                                                           \
    do {                                                                                       \
        hipLaunchKernelGGL(KERN, dim3(GRID), dim3(kBlock), 0, s, __VA_ARGS__);                 \
        return hipGetLastError();                                                              \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Second pass for VarReducer / StdReducer (internal/reducer/reducer.go:352-429).  The reference
// runs Welford's recurrence row by row; a parallel scan uses the corrected two-pass form instead:
//     K = mean estimate of pass 1,  D = sum (v - K),  Q = sum (v - K)^2,
//     m2 = Q - D*D/n,  var = m2 / (n - 1)
// v - K is kept exact as a double-double (TwoSum), its square to ~2^-104, both sums double-double,
// so Q carries no cancellation (all terms >= 0) and the host's final m2 is correctly rounded in
// practice.  Any non-finite selected value makes Q NaN -- as it makes the reference's m2 NaN.
// ---------------------------------------------------------------------------------------------
struct Mom {
    double qh, ql, dh, dl;
    __device__ __forceinline__ void init() { qh = ql = dh = dl = 0.0; }
    __device__ __forceinline__ void add(double v, double k0, bool hit) {
        double a = v - k0;
        const double bb = a - v;
        double b = (v - (a - bb)) + (-k0 - bb);
        if (!hit) { a = 0.0; b = 0.0; }
        const double p = a * a;
        const double e = __builtin_fma(a, a, -p) + 2.0 * a * b;
        dd_add(qh, ql, p, e);
        dd_add(dh, dl, a, b);
    }
    __device__ __forceinline__ void merge(const Mom &o) {
        dd_add(qh, ql, o.qh, o.ql);
        dd_add(dh, dl, o.dh, o.dl);
    }
};

__device__ __forceinline__ void mom_block_reduce(Mom &m, Mom *sm) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Mom o;
        o.qh = shfl_down_f64(m.qh, off); o.ql = shfl_down_f64(m.ql, off);
        o.dh = shfl_down_f64(m.dh, off); o.dl = shfl_down_f64(m.dl, off);
        m.merge(o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < kBlock / 64; w++) m.merge(sm[w]);
}

__device__ __forceinline__ double moments_shift(const AggStateDev *st, double shift) {
    if (st == nullptr) return shift;
    return st->count > 0 ? st->fsum_hi / (double)st->count : 0.0;
}

}  // namespace knox
