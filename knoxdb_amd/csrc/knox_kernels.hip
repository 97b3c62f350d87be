// knox_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the KnoxDB pack-scan path.
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
#include "knox_kernels.h"
#include <utility>

#define KNOX_HD __host__ __device__
#define KNOX_XXH3_CONST __constant__
#include "knox_regex.h"
#include "knox_xxh3.h"

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

int tile_rows(int width) { return 64 * (16 / width) * kUnroll * kWavesPerBlock; }

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

// exact 128-bit integer sum (|S| < 2^106, true for any column that fits one GPU) -> double-double with hi + lo == S
__device__ __forceinline__ void int128_to_dd(uint64_t xlo, int64_t xhi, double &hi, double &lo) {
    const double p1 = (double)xhi * 18446744073709551616.0;            // exact: |xhi| < 2^42
    const double p2 = (double)(uint32_t)(xlo >> 32) * 4294967296.0;    // exact
    const double p3 = (double)(uint32_t)xlo;                            // exact
    double s = p1 + p2;
    double bb = s - p1;
    const double e1 = (p1 - (s - bb)) + (p2 - bb);                      // TwoSum: exact
    const double s2 = s + p3;
    bb = s2 - s;
    const double e2 = (s - (s2 - bb)) + (p3 - bb);
    const double l = e1 + e2;                                           // integers below 2^54: exact
    hi = s2 + l;                                                        // renormalize (Fast2Sum, |s2| >= |l|)
    lo = l - (hi - s2);
}

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
        if (__any((uint64_t)imin < wm.smin || (uint64_t)imax > wm.smax)) {
#pragma unroll 1
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t rbase = row_base + row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
#pragma unroll
                for (int j = 0; j < K; j++) acc_update<W, A_MINMAX>(acc, elem<W>(r[u], j), (q[u] >> j) & 1u, rbase + j, FLV);
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
            trig = (uint64_t)imin < wm.smin || (uint64_t)imax > wm.smax;
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
#pragma unroll 1
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t rbase = row_base + row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
#pragma unroll
                for (int j = 0; j < K; j++) acc_update<W, A_MINMAX>(acc, elem<W>(r[u], j), (q[u] >> j) & 1u, rbase + j, FLV);
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
        // okey_raw of these types IS the W-byte key zero-extended: comparable with the wave-wide 64-bit bounds
        if (__any((uint64_t)imin < wm.smin || (uint64_t)imax > wm.smax)) {
#pragma unroll 1
            for (int u = 0; u < kUnroll; u++) {
                const uint64_t rbase = row_base + row0 + (uint64_t)u * RPL + (uint64_t)lane * K;
#pragma unroll
                for (int j = 0; j < K; j++) acc_update<W, A_MINMAX>(acc, elem<W>(r[u], j), (q[u] >> j) & 1u, rbase + j, FLV);
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

template <int W, int NL, unsigned AGG, bool SEP, bool ANYF, unsigned FLV>
__global__ void __launch_bounds__(kBlock) scan_kernel(const ScanArgs a) {
    using U = typename WT<W>::U;
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;                       // rows per wave-load
    constexpr int WAVE_ROWS = RPL * kUnroll;
    constexpr uint32_t QMASK = (K == 32) ? 0xffffffffu : ((1u << K) - 1u);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // the wave's bitset words of up to kMaxIters iterations, written out as one contiguous burst
    __shared__ __attribute__((aligned(16))) uint32_t stage[kWavesPerBlock][kMaxIters * WAVE_ROWS / 32];

    Acc acc;
    acc.init();
    WaveMM wm;
    wm.init();
    FirstLast fl;
    fl.init();
    uint32_t cnt32 = 0;
    const uint32_t iters = a.iters;

    U lo[NL], diff[NL];
    bool isf[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        lo[l] = (U)a.leaf[l].lo;
        diff[l] = (U)a.leaf[l].diff;
        isf[l] = a.leaf[l].is_float != 0;
    }
    const uint32_t tt = (uint32_t)a.tt;  // <= 16 table entries for <= 4 leaves
    // single leaf: q = in-range bits, then (q & m1) | (~q & m0) applies hit / negated / constant
    const uint32_t m1 = (tt & 2u) ? QMASK : 0u, m0 = (tt & 1u) ? QMASK : 0u;
    Swar<(W <= 2 ? W : 1)> sw;
    if constexpr (W <= 2 && NL == 1) sw.init((uint32_t)lo[0], (uint32_t)diff[0], (tt & 2u) != 0, (tt & 1u) != 0);
    const int qshift = K * (lane & (G - 1));
    // order-key range the selected values of a plain single-leaf range predicate can take (narrow integer types): the
    // leaf's [lo, lo + diff] in key space when it does not wrap, the whole type otherwise
    uint64_t mm_lo = 0, mm_hi = (uint64_t)WT<W>::MASK;
    if constexpr (W <= 2 && NL == 1 && FLV != FL_FLOAT) {
        if (tt == 2u) {
            const uint64_t klo = (uint64_t)((lo[0] ^ (FLV == FL_SIGNED ? WT<W>::SIGN : (U)0)) & WT<W>::MASK);
            if (klo + (uint64_t)diff[0] <= (uint64_t)WT<W>::MASK) { mm_lo = klo; mm_hi = klo + (uint64_t)diff[0]; }
        }
    }

    uint32_t lb = blockIdx.x;
#ifdef KNOX_DIAG
    // diagnostics (debug bit3): XCD-contiguous mapping -- workgroups are dealt round-robin over the 8
    // XCDs, so logical block (b%8)*(G/8) + b/8 gives every XCD one contiguous run of tiles per round
    if ((a.debug & 8u) && (gridDim.x & 7u) == 0) lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    // per-wave-tile popcounts, laid out [workgroup][wave][round] so that a wave's counts are contiguous
    uint32_t *tcbase = a.tile_counts ? a.tile_counts + ((uint64_t)lb * kWavesPerBlock + (uint64_t)wave) * a.tc_rounds : nullptr;
    uint32_t tcreg = 0, round = 0, cnt_prev = 0;
    // packs pruned by the caller's statistics are never touched: with a live-pack list the virtual tile index walks
    // only the surviving packs (their bitset ranges were zeroed by the host).  A wave owns iters*WAVE_ROWS consecutive
    // rows of its tile (one contiguous bitset burst): the first of them
    auto wave_row0 = [&](uint64_t vt) -> uint64_t {
        uint64_t tile = vt;
        if (a.live_packs != nullptr) tile = ((uint64_t)a.live_packs[vt >> a.tpp_shift] << a.tpp_shift) + (vt & ((1ull << a.tpp_shift) - 1));
        return (tile * kWavesPerBlock + (uint64_t)wave) * iters * WAVE_ROWS;
    };
    // the kUnroll 16-byte loads per column of one wave iteration
    auto load_iter = [&](u32x4 (&r)[NL][kUnroll], u32x4 (&ra)[kUnroll], uint64_t row0) {
        const uint64_t vec0 = row0 / K + lane;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const u32x4 *p = (const u32x4 *)a.leaf[l].col + vec0;
#pragma unroll
#ifdef KNOX_LOAD_POLICY
            for (int u = 0; u < kUnroll; u++) r[l][u] = ld16_policy(p + u * 64);
#else
            for (int u = 0; u < kUnroll; u++) r[l][u] = ld16<true>(p + u * 64);
#endif
        }
#ifdef KNOX_LOAD_POLICY
        static_assert(kUnroll == 4, "ld_fence takes 4 registers");
#pragma unroll
        for (int l = 0; l < NL; l++) ld_fence(r[l][0], r[l][1], r[l][2], r[l][3]);
#endif
        if constexpr (SEP && AGG != 0) {
            const u32x4 *p = (const u32x4 *)a.agg_col + vec0;
#pragma unroll
            for (int u = 0; u < kUnroll; u++) ra[u] = ld16<true>(p + u * 64);
        }
    };
    // one wave iteration over loaded registers: predicate, bitset words into the LDS stage, aggregates; the last
    // iteration of a tile writes the wave's burst
    auto body = [&](const u32x4 (&r)[NL][kUnroll], const u32x4 (&ra)[kUnroll], uint64_t wrow0, uint32_t it) {
        const uint64_t row0 = wrow0 + (uint64_t)it * WAVE_ROWS;  // wave-uniform
        uint32_t q[kUnroll];
        constexpr bool kSwar = W <= 2 && NL == 1 && !SEP && (AGG == 0 || AGG == A_ISUM);
        constexpr bool kSwarAgg = W <= 2 && NL == 1 && !SEP && !kSwar && FLV != FL_FLOAT;  // every other aggregate class of the narrow types
        if constexpr (kSwarAgg) {
            constexpr int WS = W <= 2 ? W : 1;
            constexpr unsigned FS = FLV == FL_FLOAT ? FL_UNSIGNED : FLV;
            // Every selected value lies inside the predicate's own range, so once the wave holds a value AT both ends of
            // that range (strict compares: first occurrence wins) no later row can change its min / max: the min / max
            // work is dropped for the rest of the wave's rows (wave-uniform branch; low-cardinality columns get there fast)
            if ((AGG & A_MINMAX) != 0 && wm.smin <= mm_lo && wm.smax >= mm_hi)
                agg_iter_swar<WS, (AGG & ~A_MINMAX), FS>(acc, wm, fl, sw, r[0], q, cnt32, row0, a.row_base, lane);
            else
                agg_iter_swar<WS, AGG, FS>(acc, wm, fl, sw, r[0], q, cnt32, row0, a.row_base, lane);
        } else if constexpr (kSwar) {
            uint32_t sum32 = 0;  // <= 64 values x 255 resp. 32 x 65535 per iteration
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                q[u] = swar_load<W, AGG == A_ISUM>(sw, r[0][u], sum32);
                cnt32 += __builtin_popcount(q[u]);
            }
            if constexpr (AGG == A_ISUM) acc.isum += sum32;
        } else {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            uint32_t qq = 0;
            if constexpr (NL == 1 && W >= 4 && (AGG & ~A_ISUM) != 0) {
                // VALU-bound instantiations: the compare's carry shifts straight into the word, qq = 2*qq + (d <= diff),
                // highest element first (one v_addc instead of a select + or per value)
#pragma unroll
                for (int j = K - 1; j >= 0; j--) {
                    U x = elem<W>(r[0][u], j);
                    if constexpr (ANYF) x = fkey<W>(x);
                    const U d = (U)(x - lo[0]);
                    if constexpr (W == 4) asm("v_cmp_ge_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(qq) : "v"(d), "s"(diff[0]) : "vcc");
                    else asm("v_cmp_ge_u64 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(qq) : "v"(d), "s"(diff[0]) : "vcc");
                }
            } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
                if constexpr (NL == 1) {
                    U x = elem<W>(r[0][u], j);
                    if constexpr (ANYF && W >= 4) x = fkey<W>(x);  // one leaf and ANYF: it is the float one
                    qq |= ((((U)(x - lo[0])) & WT<W>::MASK) <= diff[0] ? 1u : 0u) << j;
                } else {
                    uint32_t idx = 0;
#pragma unroll
                    for (int l = 0; l < NL; l++) {
                        U x = elem<W>(r[l][u], j);
                        if constexpr (ANYF && W >= 4) x = isf[l] ? fkey<W>(x) : x;
                        idx |= ((((U)(x - lo[l])) & WT<W>::MASK) <= diff[l] ? 1u : 0u) << l;
                    }
                    qq |= ((tt >> idx) & 1u) << j;
                }
            }
            }
            if constexpr (NL == 1) qq = (qq & m1) | (~qq & m0);
            q[u] = qq;
            cnt32 += __builtin_popcount(qq);
        }
        }  // !kSwar
        if (a.bits != nullptr) {
            // assemble the 32-bit words (4 independent DPP chains), stage them in LDS so the wave's
            // bitset leaves as ONE 16-byte-per-lane store
            uint32_t w[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) w[u] = q[u] << qshift;
            if constexpr (G >= 2) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0xB1>(w[u]);   // quad_perm [1,0,3,2]
            }
            if constexpr (G >= 4) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0x4E>(w[u]);   // quad_perm [2,3,0,1]
            }
            if constexpr (G >= 8) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0x141>(w[u]);  // row_half_mirror
            }
            if constexpr (G >= 16) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0x140>(w[u]);  // row_mirror
            }
            if ((lane & (G - 1)) == 0) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) stage[wave][it * (WAVE_ROWS / 32) + u * (64 / G) + (lane / G)] = w[u];
            }
        }
        if constexpr (AGG != 0 && !kSwar && !kSwarAgg) {
            if constexpr (SEP) agg_iter<W, AGG, FLV>(acc, wm, fl, ra, q, row0, a.row_base, lane);
            else agg_iter<W, AGG, FLV>(acc, wm, fl, r[0], q, row0, a.row_base, lane);
        }
        if (it + 1 == iters) {
        if (a.bits != nullptr) {
            // LDS ops of one wave execute in order: the reads below see the words written above.
            // One burst of iters * WAVE_ROWS/8 contiguous bytes per wave (1 KiB for 8-byte types at
            // 16 iterations): whole 128-B lines, 16-byte-per-lane streaming (nt) stores.
            const uint32_t nv = iters * (WAVE_ROWS / 128);  // 16-byte vectors
            u32x4 *dst = (u32x4 *)(a.bits + (wrow0 >> 5));
            // this lane's share of the burst's popcount is what its running count grew by (the wave total is all
            // the per-tile count needs): no popcounts over the staged words
            uint32_t tc = cnt32 - cnt_prev;
            cnt_prev = cnt32;
            for (uint32_t v = lane; v < nv; v += 64) {
                const u32x4 x = ((const u32x4 *)stage[wave])[v];
#ifdef KNOX_DIAG
                // diagnostics build only (make diag): debug bit0 = words are assembled but never leave the CU
                if (a.debug & 2u) dst[v] = x;  // plain (L2 write-back) store
                else if (a.debug & 4u) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + v), "v"(x) : "memory");
                else if (!(a.debug & 1u) || x.x == 0x12345678u) __builtin_nontemporal_store(x, dst + v);
#else
                __builtin_nontemporal_store(x, dst + v);
#endif
            }
            if (tcbase != nullptr) {
                // popcount of this wave tile's burst, once per tile.  The counts of 32 consecutive
                // rounds are parked one per lane and leave as ONE full 128-byte line (no partial writes).
                // (DPP row butterfly + 4 readlanes: a ds_bpermute shuffle chain here costs ~700 cycles of
                // exposed latency per tile, 5 % of the kernel at 2 workgroups/CU)
                tc += dpp_mov<0xB1>(tc);
                tc += dpp_mov<0x4E>(tc);
                tc += dpp_mov<0x141>(tc);
                tc += dpp_mov<0x140>(tc);
                const uint32_t total = (uint32_t)(__builtin_amdgcn_readlane((int)tc, 0) + __builtin_amdgcn_readlane((int)tc, 16) +
                                                  __builtin_amdgcn_readlane((int)tc, 32) + __builtin_amdgcn_readlane((int)tc, 48));
                tcreg = lane == (int)(round & 31u) ? total : tcreg;
                if ((round & 31u) == 31u && lane < 32) tcbase[(round & ~31u) + lane] = tcreg;
                round++;
            }
        }
        }
    };
    // Aggregate-heavy single-column scans are software pipelined: the loads of the NEXT wave iteration (the next tile's
    // first one at a tile's end) are in flight while the current one is evaluated -- two register sets, roles fixed by an
    // even iteration count, no register moves.  With 2-3 workgroups per CU resident, waiting out the full HBM latency
    // once per iteration is what these instantiations were bound by (measured: float32 +4 %, int16 ALL +7 %, int32 ALL
    // +2 %).  The plain filter / filter+sum kernels do so little between loads that the extra loads in flight only
    // disturb the read / write mix at the memory controllers (int64 headline -10 %): they keep the simple loop.
    constexpr bool kPipe = NL == 1 && !SEP && (AGG & ~A_ISUM) != 0;
    if (kPipe && (iters & 1u) == 0) {
        u32x4 ra[kUnroll];
        u32x4 ra_a[NL][kUnroll], rb[NL][kUnroll];
        uint64_t vt = lb;
        if (vt < a.n_tiles) {
            uint64_t w0 = wave_row0(vt);
            load_iter(ra_a, ra, w0);
            for (;;) {
                const uint64_t vn = vt + gridDim.x;
                const bool more = vn < a.n_tiles;  // wave-uniform
                const uint64_t wn = more ? wave_row0(vn) : 0;
#pragma unroll 1
                for (uint32_t it = 0; it < iters; it += 2) {
                    load_iter(rb, ra, w0 + (uint64_t)(it + 1) * WAVE_ROWS);
                    body(ra_a, ra, w0, it);
                    // always issued (at the very end: a reload of this tile's first rows, never used), so the waits below
                    // count on a fixed number of loads in flight and only wait for the older set
                    load_iter(ra_a, ra, it + 2 < iters ? w0 + (uint64_t)(it + 2) * WAVE_ROWS : more ? wn : w0);
                    body(rb, ra, w0, it + 1);
                }
                if (!more) break;
                vt = vn;
                w0 = wn;
            }
        }
    } else {
        for (uint64_t vt = lb; vt < a.n_tiles; vt += gridDim.x) {
            const uint64_t w0 = wave_row0(vt);
#pragma unroll 1
            for (uint32_t it = 0; it < iters; it++) {
                u32x4 r[NL][kUnroll];
                u32x4 ra[kUnroll];
                load_iter(r, ra, w0 + (uint64_t)it * WAVE_ROWS);
                body(r, ra, w0, it);
            }
        }
    }
    if (tcbase != nullptr && (round & 31u) != 0 && lane < (int)(round & 31u)) tcbase[(round & ~31u) + lane] = tcreg;
    acc.count = cnt32;
    if constexpr ((AGG & A_FIRSTLAST) != 0) fl.template finish<K>(acc, a.row_base, lane);
    block_reduce_store(acc, a.partials + blockIdx.x);
}

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
template <int W> __device__ __forceinline__ typename WT<W>::U load_elem(const void *col, uint64_t row) {
    if constexpr (W == 8) return ((const uint64_t *)col)[row];
    else if constexpr (W == 4) return ((const uint32_t *)col)[row];
    else if constexpr (W == 2) return ((const uint16_t *)col)[row];
    else return ((const uint8_t *)col)[row];
}

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

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
#define KNOX_LAUNCH(KERN, GRID, ...)                                                           \
    do {                                                                                       \
        hipLaunchKernelGGL(KERN, dim3(GRID), dim3(kBlock), 0, s, __VA_ARGS__);                 \
        return hipGetLastError();                                                              \
    } while (0)

template <int W, int NL, unsigned AGG, bool SEP, unsigned FLV>
static hipError_t scan_anyf(hipStream_t s, int grid, const ScanArgs &a) {
    bool anyf = false;
    for (int l = 0; l < NL; l++) anyf = anyf || a.leaf[l].is_float != 0;
    if constexpr (W >= 4) {
        if (anyf) KNOX_LAUNCH((scan_kernel<W, NL, AGG, SEP, true, FLV>), grid, a);
    }
    KNOX_LAUNCH((scan_kernel<W, NL, AGG, SEP, false, FLV>), grid, a);
}

template <int W, int NL, unsigned AGG, unsigned FLV>
static hipError_t scan_sep(hipStream_t s, bool sep, int grid, const ScanArgs &a) {
    return sep ? scan_anyf<W, NL, AGG, true, FLV>(s, grid, a) : scan_anyf<W, NL, AGG, false, FLV>(s, grid, a);
}

template <int W, int NL>
static hipError_t scan_agg(hipStream_t s, unsigned agg, bool sep, int grid, const ScanArgs &a) {
    // aggregate classes instantiated: none | isum | full-without-isum (per flavour) | all (ints)
    constexpr unsigned FF = A_FSUM | A_MINMAX | A_FIRSTLAST;
    if (agg == 0) return scan_anyf<W, NL, 0, false, FL_SIGNED>(s, grid, a);
    if (agg == A_ISUM) return scan_sep<W, NL, A_ISUM, FL_SIGNED>(s, sep, grid, a);
    if constexpr (W >= 4) {
        if (a.agg_flavor == FL_FLOAT) return scan_sep<W, NL, FF, FL_FLOAT>(s, sep, grid, a);
    }
    if ((agg & A_ISUM) == 0) {
        return a.agg_flavor == FL_SIGNED ? scan_sep<W, NL, FF, FL_SIGNED>(s, sep, grid, a) : scan_sep<W, NL, FF, FL_UNSIGNED>(s, sep, grid, a);
    }
    return a.agg_flavor == FL_SIGNED ? scan_sep<W, NL, A_ALL, FL_SIGNED>(s, sep, grid, a) : scan_sep<W, NL, A_ALL, FL_UNSIGNED>(s, sep, grid, a);
}

template <int W>
static hipError_t scan_nl(hipStream_t s, int nl, unsigned agg, bool sep, int grid, const ScanArgs &a) {
    switch (nl) {
    case 1: return scan_agg<W, 1>(s, agg, sep, grid, a);
    case 2: return scan_agg<W, 2>(s, agg, sep, grid, a);
    case 3: return scan_agg<W, 3>(s, agg, sep, grid, a);
    case 4: return scan_agg<W, 4>(s, agg, sep, grid, a);
    default: return hipErrorInvalidValue;
    }
}

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

// The allocator's placement probe (knox_abi.cpp): the headline code path under its own symbol
// (FLV is irrelevant to an integer sum), so that profiles keep probe launches and real scans apart.
hipError_t launch_probe_scan(hipStream_t s, int grid, const ScanArgs &a) {
    if (grid <= 0) return hipSuccess;
    KNOX_LAUNCH((scan_kernel<8, 1, A_ISUM, false, false, FL_UNSIGNED>), grid, a);
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
// Time-window aggregation ("next" row f1): pkg/series/series.go:192-246 -> NativeBucket.Push
// (internal/reducer/bucket_native.go:104-159) keeps one reducer per window [from + k*step,
// from + (k+1)*step).  With a non-decreasing time column every window is ONE contiguous row range:
// the ranges come from a binary search per window edge, are cut into work items, each work item is
// reduced by one workgroup with the same exact machinery as the global aggregates, and the
// partials of a window are merged in row order.  Out-of-order timestamps (which the reference
// handles by inserting reducers, :123-141) are detected and refused: the caller keeps the host path.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) sorted_check_kernel(const int64_t *time, uint64_t n_rows, uint32_t *flag) {
    const int lane = threadIdx.x & 63;
    bool bad = false;
    const uint64_t npairs = n_rows >> 1;  // 16-byte vectors of two timestamps
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < ((npairs + 63) & ~63ull); v += (uint64_t)gridDim.x * blockDim.x) {
        int64_t e0 = INT64_MIN, e1 = INT64_MIN;
        const bool in = v < npairs;
        if (in) {
            const u32x4 r = ld16<true>((const u32x4 *)time + v);
            e0 = (int64_t)(((uint64_t)r.y << 32) | r.x);
            e1 = (int64_t)(((uint64_t)r.w << 32) | r.z);
            bad = bad || e0 > e1;
        }
        int64_t nxt = (int64_t)__shfl_down((unsigned long long)e0, 1, 64);  // first timestamp of the next lane's pair
        if (lane == 63 || v + 1 >= npairs) nxt = (2 * v + 2 < n_rows) ? time[2 * v + 2] : INT64_MAX;
        if (in) bad = bad || e1 > nxt;
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1u);
}

hipError_t launch_sorted_check(hipStream_t s, const int64_t *time, uint64_t n_rows, uint32_t *flag) {
    if (n_rows < 2) return hipSuccess;
    uint64_t g = ((n_rows >> 1) + kBlock - 1) / kBlock;
    if (g > 2048) g = 2048;
    if (g == 0) g = 1;
    KNOX_LAUNCH(sorted_check_kernel, (int)g, time, n_rows, flag);
}

__global__ void __launch_bounds__(kBlock) bucket_bounds_kernel(const int64_t *time, uint64_t n_rows, int64_t t_from, int64_t step,
                                                               uint64_t n_buckets, uint64_t *row_start) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_buckets) return;
    const int64_t edge = t_from + (int64_t)k * step;  // host checked: no overflow
    uint64_t lo = 0, hi = n_rows;                    // lower_bound(time, edge)
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
        // aligned middle: whole wave tiles, vector path
        const uint64_t a0 = (rb + WAVE_ROWS - 1) / WAVE_ROWS, a1 = re / WAVE_ROWS;  // tile indices [a0, a1)
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
        // ragged edges (or the whole item when it holds no aligned tile): one row per lane, rows
        // ascending per lane, exact per-row update
        const uint64_t e0 = a0 < a1 ? a0 * WAVE_ROWS : re;  // head edge [rb, e0)
        const uint64_t e1 = a0 < a1 ? a1 * WAVE_ROWS : re;  // tail edge [e1, re)
        for (int part = 0; part < 2; part++) {
            const uint64_t lo = part == 0 ? rb : e1, hi = part == 0 ? e0 : re;
            for (uint64_t row = lo + threadIdx.x; row < hi; row += blockDim.x) {
                const uint32_t hit = mask8 ? (uint32_t)((mask8[row >> 3] >> (row & 7)) & 1u) : 1u;
                acc.count += hit;
                acc_update<W, (AGG | A_FIRSTLAST)>(acc, load_elem<W>(a.agg_col, row), hit, a.row_base + row, FLV);
            }
        }
        acc.count += cnt32;
        if constexpr ((AGG & A_FIRSTLAST) != 0) fl.template finish<K>(acc, a.row_base, lane);
        block_reduce_store(acc, a.partials + it);
        __syncthreads();  // block_reduce_store's LDS is reused by the next item
    }
}

// many small windows: one WAVE per work item -- no workgroup barrier and no LDS round per item (a window of a few hundred
// rows costs a few loop iterations + one wave reduction); the rows are taken lane-strided, first / last ride in the Acc
template <int W, unsigned AGG, unsigned FLV>
__global__ void __launch_bounds__(kBlock) bucket_reduce_wave_kernel(const ScanArgs a, const BucketItem *items, uint32_t n_items) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = gridDim.x * kWavesPerBlock;
    const uint8_t *mask8 = (const uint8_t *)a.mask;
    for (uint32_t it = wave; it < n_items; it += n_waves) {
        const uint64_t rb = items[it].begin, re = items[it].end;
        Acc acc;
        acc.init();
        for (uint64_t row = rb + lane; row < re; row += 64) {
            const uint32_t hit = mask8 ? (uint32_t)((mask8[row >> 3] >> (row & 7)) & 1u) : 1u;
            acc.count += hit;
            acc_update<W, (AGG | A_FIRSTLAST)>(acc, load_elem<W>(a.agg_col, row), hit, a.row_base + row, FLV);
        }
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
template <int W>
__global__ void __launch_bounds__(kBlock) bucket_finalize_kernel(const AggPartial *partials, const uint32_t *item_start, uint32_t n_buckets,
                                                                 const void *agg_col, uint64_t row_base, int dtype, AggStateDev *state_out) {
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
    state_out[k] = s;
}

hipError_t launch_bucket_finalize(hipStream_t s, int width, const AggPartial *partials, const uint32_t *item_start, uint32_t n_buckets,
                                  const void *agg_col, uint64_t row_base, int dtype, AggStateDev *state_out) {
    if (n_buckets == 0) return hipSuccess;
    const int grid = (int)((n_buckets + kBlock - 1) / kBlock);
    switch (width) {
    case 8: KNOX_LAUNCH((bucket_finalize_kernel<8>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out);
    case 4: KNOX_LAUNCH((bucket_finalize_kernel<4>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out);
    case 2: KNOX_LAUNCH((bucket_finalize_kernel<2>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out);
    case 1: KNOX_LAUNCH((bucket_finalize_kernel<1>), grid, partials, item_start, n_buckets, agg_col, row_base, dtype, state_out);
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
static hipError_t bucket_moments_w(hipStream_t s, uint32_t flavor, int grid, const void *col, const uint32_t *mask, const BucketItem *items,
                                   uint32_t n_items, const AggStateDev *states, double *partials) {
    if constexpr (W >= 4) {
        if (flavor == FL_FLOAT) KNOX_LAUNCH((bucket_moments_kernel<W, FL_FLOAT>), grid, col, mask, items, n_items, states, partials);
    }
    if (flavor == FL_SIGNED) KNOX_LAUNCH((bucket_moments_kernel<W, FL_SIGNED>), grid, col, mask, items, n_items, states, partials);
    KNOX_LAUNCH((bucket_moments_kernel<W, FL_UNSIGNED>), grid, col, mask, items, n_items, states, partials);
}

hipError_t launch_bucket_moments(hipStream_t s, int width, uint32_t flavor, int grid, const void *col, const uint32_t *mask,
                                 const BucketItem *items, uint32_t n_items, const uint32_t *item_start, uint32_t n_buckets,
                                 AggStateDev *states, double *partials) {
    if (n_buckets == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (n_items != 0 && grid > 0) {
        switch (width) {
        case 8: e = bucket_moments_w<8>(s, flavor, grid, col, mask, items, n_items, states, partials); break;
        case 4: e = bucket_moments_w<4>(s, flavor, grid, col, mask, items, n_items, states, partials); break;
        case 2: e = bucket_moments_w<2>(s, flavor, grid, col, mask, items, n_items, states, partials); break;
        case 1: e = bucket_moments_w<1>(s, flavor, grid, col, mask, items, n_items, states, partials); break;
        default: return hipErrorInvalidValue;
        }
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(bucket_moments_finalize_kernel, dim3((n_buckets + kBlock - 1) / kBlock), dim3(kBlock), 0, s, (const double *)partials,
                       item_start, n_buckets, states);
    return hipGetLastError();
}

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
// the i256 twins): membership in a sorted set; keys[k * L + l] = limb l (most significant first, top limb
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
            x[0] ^= 1ull << 63;
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

hipError_t launch_wide_inset(hipStream_t s, int n_limbs, bool negate, const uint64_t *const *limbs, uint64_t n_rows, const uint64_t *keys,
                             uint32_t n_keys, uint8_t *bits, const uint8_t *mask) {
    if (n_rows == 0) return hipSuccess;
    uint64_t g = (n_rows + kBlock - 1) / kBlock;
    if (g > 2048 * 4) g = 2048 * 4;
    if (n_limbs == 2) KNOX_LAUNCH(wide_inset_kernel<2>, (int)g, negate ? 1 : 0, limbs[0], limbs[1], nullptr, nullptr, n_rows, keys, n_keys, bits, mask);
    if (n_limbs == 4) KNOX_LAUNCH(wide_inset_kernel<4>, (int)g, negate ? 1 : 0, limbs[0], limbs[1], limbs[2], limbs[3], n_rows, keys, n_keys, bits, mask);
    return hipErrorInvalidValue;
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
__global__ void __launch_bounds__(kBlock) popcount_seg_kernel(const uint64_t *bits, uint64_t nwords, uint64_t size_bits,
                                                              uint64_t seg_words, int64_t *counts, uint64_t n_seg) {
    __shared__ uint64_t sm[kBlock / 64];
    for (uint64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
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
        if (threadIdx.x == 0) counts[seg] = (int64_t)(sm[0] + sm[1] + sm[2] + sm[3]);
        __syncthreads();
    }
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

hipError_t launch_popcount_segments(hipStream_t s, const uint64_t *bits, uint64_t size_bits, uint64_t seg_bits,
                                    int64_t *counts, uint64_t n_seg) {
    if (n_seg == 0) return hipSuccess;
    const uint64_t nwords = (size_bits + 63) >> 6;
    uint64_t g = n_seg > 65535 ? 65535 : n_seg;
    KNOX_LAUNCH(popcount_seg_kernel, (int)g, bits, nwords, size_bits, seg_bits >> 6, counts, n_seg);
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

// ---------------------------------------------------------------------------------------------
// bytes column EQ / NE: pkg/stringx/match.go:20-84, layout pool.go:45-58 (ptr = off<<32 | len)
// one lane = one string; wave ballot -> 8 result bytes written by lanes 0..7
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bytes_equal(const uint8_t *p, const uint8_t *v, uint32_t n) {
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) {  // 16-byte pieces: half the load instructions of the row, the value side is wave-uniform
        u32x4 x, y;
        __builtin_memcpy(&x, p + i, 16);
        __builtin_memcpy(&y, v + i, 16);
        if (((x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w)) != 0) return false;
    }
    for (; i + 8 <= n; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, p + i, 8);
        __builtin_memcpy(&y, v + i, 8);
        if (x != y) return false;
    }
    for (; i < n; i++)
        if (p[i] != v[i]) return false;
    return true;
}

// bytes.Compare(p[:n], v[:m]): -1 / 0 / +1, byte-lexicographic, a proper prefix sorts first
__device__ __forceinline__ int bytes_cmp3(const uint8_t *p, uint32_t n, const uint8_t *v, uint32_t m) {
    const uint32_t c = n < m ? n : m;
    uint32_t i = 0;
    for (; i + 8 <= c; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, p + i, 8);
        __builtin_memcpy(&y, v + i, 8);
        if (x != y) return __builtin_bswap64(x) < __builtin_bswap64(y) ? -1 : 1;  // first differing byte decides
    }
    for (; i < c; i++)
        if (p[i] != v[i]) return p[i] < v[i] ? -1 : 1;
    return n == m ? 0 : (n < m ? -1 : 1);
}

// ---- walking a bytes column (BytesView) in groups of 64 rows = one wave ------------------------------------
// Without a live list group g covers rows [64g, 64g + 64).  With one, only the groups of the listed packs exist:
// group g = group (g % gpp) of pack live[g / gpp], gpp = pack_rows / 64.  A group never straddles two packs
// (pack_rows is a multiple of 64), so the pool base is wave-uniform.
__device__ __forceinline__ uint64_t bv_ngroups(const BytesView &v) {
    if (v.live != nullptr) return (uint64_t)v.n_live * (v.pack_rows >> 6);
    return (v.n_rows + 63) >> 6;
}
__device__ __forceinline__ uint64_t bv_group_row0(const BytesView &v, uint64_t g) {
    if (v.live == nullptr) return g << 6;
    uint64_t li, in;
    if (v.pack_shift != 255u) {
        li = g >> (v.pack_shift - 6u);
        in = g & ((1ull << (v.pack_shift - 6u)) - 1ull);
    } else {
        const uint64_t gpp = v.pack_rows >> 6;
        li = g / gpp;
        in = g - li * gpp;
    }
    return (uint64_t)v.live[li] * v.pack_rows + (in << 6);
}
__device__ __forceinline__ const uint8_t *bv_pool(const BytesView &v, uint64_t row0) {
    if (v.pool_base == nullptr) return v.buf;
    const uint64_t pack = v.pack_shift != 255u ? row0 >> v.pack_shift : row0 / v.pack_rows;
    return v.buf + v.pool_base[pack];
}
// the wave's 64 result bits of the group starting at row0 -> bitset bytes: overwrite (mask == null; MatchEqual's
// unmasked form overwrites every byte incl. a partial last one) or bits.Set for mask-selected rows only
__device__ __forceinline__ void bv_store_group(uint64_t ballot, uint64_t row0, uint64_t n_rows, int lane, uint8_t *bits, const uint8_t *mask,
                                               bool or_only) {
    if (mask == nullptr && !or_only && row0 + 64 <= n_rows) {
        if (lane == 0) *(uint64_t *)(bits + (row0 >> 3)) = ballot;  // row0 is a multiple of 64: one aligned 8-byte store
        return;
    }
    if (lane < 8) {
        const uint64_t r0 = row0 + 8 * (uint64_t)lane;
        if (r0 < n_rows) {
            uint8_t byte = (uint8_t)(ballot >> (8 * lane));
            const uint64_t bi = r0 >> 3;
            if (mask != nullptr) byte &= mask[bi];
            if (mask != nullptr || or_only) { if (byte) bits[bi] |= byte; }
            else bits[bi] = byte;
        }
    }
}

// mode = FilterMode: 1 eq, 2 ne (MatchEqual / MatchNotEqual stringx/match.go:20-84), 3 gt, 4 ge, 5 lt,
// 6 le, 9 range [val, val2] (MatchGreater.. / MatchBetween :86-251)
__global__ void __launch_bounds__(kBlock) bytes_match_kernel(int mode, BytesView v, const uint8_t *val, uint32_t vlen, const uint8_t *val2,
                                                             uint32_t vlen2, uint8_t *bits, const uint8_t *mask,
                                                             unsigned long long *count_out) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    uint64_t cnt = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row0 = bv_group_row0(v, g);
        const uint8_t *pool = bv_pool(v, row0);
        const uint64_t row = row0 + lane;
        bool m = false;
        if (row < v.n_rows) {
            const uint64_t p = v.ptr[row];
            const uint32_t ofs = (uint32_t)(p >> 32), len = (uint32_t)p;
            if (mode <= 2) {
                const bool eq = (len == vlen) && bytes_equal(pool + ofs, val, len);
                m = mode == 1 ? eq : !eq;
            } else {
                const int c = bytes_cmp3(pool + ofs, len, val, vlen);
                switch (mode) {
                case 3: m = c > 0; break;
                case 4: m = c >= 0; break;
                case 5: m = c < 0; break;
                case 6: m = c <= 0; break;
                default: m = c >= 0 && bytes_cmp3(pool + ofs, len, val2, vlen2) <= 0; break;
                }
            }
        }
        const uint64_t ballot = __ballot(m);
        bv_store_group(ballot, row0, v.n_rows, lane, bits, mask, false);
        if (lane == 0) cnt += __builtin_popcountll(ballot);
    }
    if (lane == 0 && cnt && count_out) atomicAdd(count_out, (unsigned long long)cnt);
}

// EQ / NE against a value of >= 16 bytes: the streaming form.  A wave takes kEqGroups groups of 64 rows per
// iteration: all row pointers first, then for every row whose length matches TWO independent 16-byte loads -- bytes
// [0, 16) and [min(16, vlen - 16), +16), which cover the whole value up to 32 bytes -- so 12 loads per lane are in
// flight instead of a pointer -> first piece -> second piece chain.  Values above 32 bytes finish in bytes_equal for
// the rows that still match (rare).  Per row: 8 B pointer + len B payload read, 1/8 B written.
constexpr int kEqGroups = 4;
__global__ void __launch_bounds__(kBlock) bytes_eq_kernel(int negate, BytesView v, const uint8_t *val, uint32_t vlen, uint8_t *bits,
                                                          const uint8_t *mask, unsigned long long *count_out) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    const uint32_t off_b = vlen >= 32u ? 16u : vlen - 16u;
    u32x4 va, vb;
    __builtin_memcpy(&va, val, 16);
    __builtin_memcpy(&vb, val + off_b, 16);
    uint64_t cnt = 0;
    for (uint64_t g0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kEqGroups; g0 < ngroups; g0 += nwaves_total * kEqGroups) {
        uint64_t row0[kEqGroups], p[kEqGroups];
        const uint8_t *src[kEqGroups];
#pragma unroll
        for (int k = 0; k < kEqGroups; k++) {
            const uint64_t g = g0 + k;
            row0[k] = g < ngroups ? bv_group_row0(v, g) : v.n_rows;
            const uint64_t row = row0[k] + lane;
            p[k] = row < v.n_rows ? __builtin_nontemporal_load(v.ptr + row) : ~0ull;  // len 2^32-1 != vlen: never loads
        }
        u32x4 xa[kEqGroups], xb[kEqGroups];
#pragma unroll
        for (int k = 0; k < kEqGroups; k++) {
            src[k] = (row0[k] < v.n_rows ? bv_pool(v, row0[k]) : v.buf) + (uint32_t)(p[k] >> 32);
            xa[k] = va;
            xb[k] = vb;
            if ((uint32_t)p[k] == vlen) {
                __builtin_memcpy(&xa[k], src[k], 16);
                __builtin_memcpy(&xb[k], src[k] + off_b, 16);
            }
        }
#pragma unroll
        for (int k = 0; k < kEqGroups; k++) {
            if (g0 + k >= ngroups) break;
            bool eq = (uint32_t)p[k] == vlen &&
                      (((xa[k].x ^ va.x) | (xa[k].y ^ va.y) | (xa[k].z ^ va.z) | (xa[k].w ^ va.w) | (xb[k].x ^ vb.x) | (xb[k].y ^ vb.y) |
                        (xb[k].z ^ vb.z) | (xb[k].w ^ vb.w)) == 0);
            if (eq && vlen > 32u) eq = bytes_equal(src[k] + 32, val + 32, vlen - 32u);
            const bool m = (row0[k] + lane < v.n_rows) && (negate ? !eq : eq);
            const uint64_t ballot = __ballot(m);
            bv_store_group(ballot, row0[k], v.n_rows, lane, bits, mask, false);
            if (lane == 0) cnt += __builtin_popcountll(ballot);
        }
    }
    if (lane == 0 && cnt && count_out) atomicAdd(count_out, (unsigned long long)cnt);
}

hipError_t launch_bytes_match(hipStream_t s, int mode, const BytesView &v, const uint8_t *val, uint32_t vlen, const uint8_t *val2, uint32_t vlen2,
                              uint8_t *bits, const uint8_t *mask, unsigned long long *count_out) {
    const uint64_t ngroups = v.live ? (uint64_t)v.n_live * (v.pack_rows >> 6) : (v.n_rows + 63) >> 6;
    if (ngroups == 0) return hipSuccess;
    if (mode <= 2 && vlen >= 16u) {
        uint64_t g = (ngroups + (uint64_t)kWavesPerBlock * kEqGroups - 1) / ((uint64_t)kWavesPerBlock * kEqGroups);
        // one wave iteration per wave: measured on 2^30 x 32-byte keys (tools/bench_cfg.py --cfg 4), workgroups that walk
        // the column grid-stride (2048 of them: 7.4-7.5 ms) are 15 % slower than one workgroup per 1024 rows in launch
        // order (6.50 ms = 6.6 TB/s = 83 % of the HBM peak): pointer -> payload is a dependent chain, and neighbours in
        // launch order keep the DRAM pages they touch close together
        if (g > ((uint64_t)1 << 20)) g = (uint64_t)1 << 20;
        KNOX_LAUNCH(bytes_eq_kernel, (int)g, mode == 2 ? 1 : 0, v, val, vlen, bits, mask, count_out);
    }
    uint64_t g = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > 2048 * 4) g = 2048 * 4;  // the ordered modes walk grid-stride: measured best (tools/bench_bytes.py: LT 5.5 TB/s, RANGE 4.8 TB/s)
    KNOX_LAUNCH(bytes_match_kernel, (int)g, mode, v, val, vlen, val2, vlen2, bits, mask, count_out);
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

hipError_t launch_inset(hipStream_t s, int width, uint32_t flavor, bool negate, const void *col, uint64_t n_rows,
                        const uint64_t *keys, uint32_t n_keys, uint8_t *bits, const uint8_t *mask) {
    if (n_rows == 0) return hipSuccess;
    uint64_t g = (n_rows + kBlock - 1) / kBlock;
    if (g > 2048 * 4) g = 2048 * 4;
    switch (width) {
    case 8: KNOX_LAUNCH((inset_kernel<8>), (int)g, flavor, negate ? 1 : 0, col, n_rows, keys, n_keys, bits, mask);
    case 4: KNOX_LAUNCH((inset_kernel<4>), (int)g, flavor, negate ? 1 : 0, col, n_rows, keys, n_keys, bits, mask);
    case 2: KNOX_LAUNCH((inset_kernel<2>), (int)g, flavor, negate ? 1 : 0, col, n_rows, keys, n_keys, bits, mask);
    case 1: KNOX_LAUNCH((inset_kernel<1>), (int)g, flavor, negate ? 1 : 0, col, n_rows, keys, n_keys, bits, mask);
    default: return hipErrorInvalidValue;
    }
}

// bytes IN / NOT IN (bytesInSetMatcher / bytesNotInSetMatcher.MatchVector, match_bytes.go:408-470):
// membership of the row's bytes in the set.  The reference consults a hash map and confirms with
// bytes.Equal, i.e. plain equality; here the set arrives sorted (bytes.Compare order) and unique in
// StringPool form and every lane binary-searches it.  Matching rows are OR-ed into bits (bits.Set).
__global__ void __launch_bounds__(kBlock) bytes_inset_kernel(int negate, BytesView v, const uint64_t *set_ptr, uint32_t n_set,
                                                             const uint8_t *set_buf, uint8_t *bits, const uint8_t *mask) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row0 = bv_group_row0(v, g);
        const uint8_t *pool = bv_pool(v, row0);
        const uint64_t row = row0 + lane;
        bool m = false;
        if (row < v.n_rows) {
            const uint64_t p = v.ptr[row];
            const uint8_t *x = pool + (uint32_t)(p >> 32);
            const uint32_t len = (uint32_t)p;
            uint32_t lo = 0, hi = n_set;
            bool found = false;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint64_t sp = set_ptr[mid];
                const int c = bytes_cmp3(set_buf + (uint32_t)(sp >> 32), (uint32_t)sp, x, len);
                if (c == 0) { found = true; break; }
                if (c < 0) lo = mid + 1; else hi = mid;
            }
            m = negate ? !found : found;
        }
        const uint64_t ballot = __ballot(m);
        bv_store_group(ballot, row0, v.n_rows, lane, bits, mask, true);
    }
}

hipError_t launch_bytes_inset(hipStream_t s, bool negate, const BytesView &v, const uint64_t *set_ptr, uint32_t n_set, const uint8_t *set_buf,
                              uint8_t *bits, const uint8_t *mask) {
    const uint64_t ngroups = v.live ? (uint64_t)v.n_live * (v.pack_rows >> 6) : (v.n_rows + 63) >> 6;
    if (ngroups == 0) return hipSuccess;
    uint64_t g = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > ((uint64_t)1 << 20)) g = (uint64_t)1 << 20;  // latency bound (binary search per row): one group per wave, +14 % over 8192 workgroups
    KNOX_LAUNCH(bytes_inset_kernel, (int)g, negate ? 1 : 0, v, set_ptr, n_set, set_buf, bits, mask);
}

// ---------------------------------------------------------------------------------------------
// bloom filter: internal/filter/bloom/bloom.go:136-150 (probe), :162-169 (add), xxh3.go:22-33
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rol64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xxh3_u64(uint64_t val) {
    const uint64_t input64 = (val >> 32) + (val << 32);
    uint64_t h = input64 ^ (0x1cad21f72c81017cull ^ 0xdb979083e96dd4deull);
    h ^= rol64(h, 49) ^ rol64(h, 24);
    h *= 0x9fb21c651e98df25ull;
    h ^= (h >> 35) + 8;
    h *= 0x9fb21c651e98df25ull;
    h ^= (h >> 28);
    return h;
}

__global__ void __launch_bounds__(kBlock) bloom_probe_kernel(const uint8_t *filters, const uint64_t *offsets, uint64_t n_packs,
                                                             uint64_t h, uint8_t *maybe_out) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_packs) return;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) {  // no filter for this pack: cannot exclude it
        maybe_out[p] = 1;
        return;
    }
    const uint8_t *buf = filters + o0;
    const uint8_t *bits = buf + 1;
    const uint32_t k = buf[0];
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    uint32_t h0 = (uint32_t)h;
    const uint32_t h1 = (uint32_t)(h >> 32);
    uint8_t ok = 1;
    for (uint32_t i = 0; i < k; i++) {
        if ((bits[(h0 & mask) >> 3] & (1u << (h0 & 7))) == 0) { ok = 0; break; }
        h0 += h1;
    }
    maybe_out[p] = ok;
}

hipError_t launch_bloom_probe(hipStream_t s, const uint8_t *filters, const uint64_t *offsets, uint64_t n_packs, uint64_t h,
                              uint8_t *maybe_out) {
    if (n_packs == 0) return hipSuccess;
    KNOX_LAUNCH(bloom_probe_kernel, (int)((n_packs + kBlock - 1) / kBlock), filters, offsets, n_packs, h, maybe_out);
}

// one workgroup per pack; filters pre-zeroed by the host; k byte = 4 (bloom.go:48-60)
// fixed-width value hashes (internal/hash/xxh3.go:35-58, hash.go:67-92): u32 = the 4..8 byte XXH3 branch,
// u16 / u8 = the 1..3 byte branch; floats hash their IEEE bits like the same-width integer
__device__ __forceinline__ uint64_t xxh3_rrmxmx(uint64_t h, uint64_t len) {
    h ^= rol64(h, 49) ^ rol64(h, 24);
    h *= 0x9fb21c651e98df25ull;
    h ^= (h >> 35) + len;
    h *= 0x9fb21c651e98df25ull;
    return h ^ (h >> 28);
}
__device__ __forceinline__ uint64_t xxh3_avalanche_small(uint64_t x) {
    x ^= x >> 33;
    x *= 14029467366897019727ull;
    x ^= x >> 29;
    x *= 1609587929392839161ull;
    return x ^ (x >> 32);
}
template <int W> __device__ __forceinline__ uint64_t xxh3_value(uint64_t v) {
    if constexpr (W == 8) return xxh3_u64(v);
    else if constexpr (W == 4) return xxh3_rrmxmx((v + (v << 32)) ^ (0x1cad21f72c81017cull ^ 0xdb979083e96dd4deull), 4);
    else if constexpr (W == 2) return xxh3_avalanche_small((((v * ((1ull << 24) + 1)) >> 8) + (2ull << 8)) ^ (0xbe4ba423ull ^ 0x396cfeb8ull));
    else return xxh3_avalanche_small(((v * ((1ull << 24) + (1ull << 16) + 1)) + (1ull << 8)) ^ (0xbe4ba423ull ^ 0x396cfeb8ull));
}

constexpr int kBloomSplit = 256;
template <int W>
__global__ void __launch_bounds__(kBlock) bloom_build_kernel(const void *col, uint64_t n_rows, uint64_t pack_rows, uint64_t n_packs,
                                                             uint8_t *filters, const uint64_t *offsets) {
    // kBloomSplit workgroups per pack (the bits are set with atomics, so they may share a filter), and XCD-aware: workgroups
    // are dealt round-robin over the 8 XCDs, so workgroup b works on pack 8*(b / (8*S)) + b % 8 -- each XCD sets the bits of
    // ONE pack's filter at a time, which then stays in that XCD's L2 (2 MiB of 4 MiB) instead of every atomic going to
    // HBM (one workgroup per pack, 2^30 rows: 0.22 s)
    const uint64_t p = (blockIdx.x / (8 * kBloomSplit)) * 8 + (blockIdx.x & 7), part = (blockIdx.x >> 3) % kBloomSplit;
    if (p >= n_packs) return;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    uint8_t *buf = filters + o0;
    if (threadIdx.x == 0 && part == 0) buf[0] = 4;
    const uint64_t bits_off = o0 + 1;  // byte offset of bit array inside `filters`
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    uint64_t r0 = p * pack_rows, r1 = r0 + pack_rows;
    if (r1 > n_rows) r1 = n_rows;
    for (uint64_t r = r0 + part * blockDim.x + threadIdx.x; r < r1; r += (uint64_t)blockDim.x * kBloomSplit) {
        const uint64_t h = xxh3_value<W>((uint64_t)load_elem<W>(col, r));
        uint32_t h0 = (uint32_t)h;
        const uint32_t h1 = (uint32_t)(h >> 32);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t byte_addr = bits_off + ((h0 & mask) >> 3);
            const uint32_t bit = 1u << (h0 & 7);
            // byte-granular OR through the aligned 32-bit word that contains the byte
            uint32_t *word = (uint32_t *)(filters + (byte_addr & ~3ull));
            atomicOr(word, bit << (8 * (byte_addr & 3)));
            h0 += h1;
        }
    }
}

// bytes column (StringPool): hash.Hash = XXH3-64 of every row's bytes (pack/stats/filter.go:352-354), one lane per row
__global__ void __launch_bounds__(kBlock) bloom_build_bytes_kernel(BytesView v, uint64_t n_packs, uint8_t *filters, const uint64_t *offsets) {
    const uint64_t p = (blockIdx.x / (8 * kBloomSplit)) * 8 + (blockIdx.x & 7), part = (blockIdx.x >> 3) % kBloomSplit;
    if (p >= n_packs) return;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    if (threadIdx.x == 0 && part == 0) filters[o0] = 4;
    const uint64_t bits_off = o0 + 1;
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    uint64_t r0 = p * v.pack_rows, r1 = r0 + v.pack_rows;
    if (r1 > v.n_rows) r1 = v.n_rows;
    const uint8_t *pool = v.pool_base ? v.buf + v.pool_base[p] : v.buf;
    for (uint64_t r = r0 + part * blockDim.x + threadIdx.x; r < r1; r += (uint64_t)blockDim.x * kBloomSplit) {
        const uint64_t pr = v.ptr[r];
        const uint64_t h = knox_xxh3::hash(pool + (uint32_t)(pr >> 32), (uint64_t)(uint32_t)pr);
        uint32_t h0 = (uint32_t)h;
        const uint32_t h1 = (uint32_t)(h >> 32);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t byte_addr = bits_off + ((h0 & mask) >> 3);
            uint32_t *word = (uint32_t *)(filters + (byte_addr & ~3ull));
            atomicOr(word, (1u << (h0 & 7)) << (8 * (byte_addr & 3)));
            h0 += h1;
        }
    }
}

hipError_t launch_bloom_build_bytes(hipStream_t s, const BytesView &v, uint8_t *filters, const uint64_t *offsets) {
    const uint64_t n_packs = (v.n_rows + v.pack_rows - 1) / v.pack_rows;
    if (n_packs == 0) return hipSuccess;
    KNOX_LAUNCH(bloom_build_bytes_kernel, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), v, n_packs, filters, offsets);
}

hipError_t launch_bloom_build(hipStream_t s, int width, const void *col, uint64_t n_rows, uint64_t pack_rows, uint8_t *filters,
                              const uint64_t *offsets) {
    const uint64_t n_packs = (n_rows + pack_rows - 1) / pack_rows;
    if (n_packs == 0) return hipSuccess;
    switch (width) {
    case 8: KNOX_LAUNCH(bloom_build_kernel<8>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    case 4: KNOX_LAUNCH(bloom_build_kernel<4>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    case 2: KNOX_LAUNCH(bloom_build_kernel<2>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    case 1: KNOX_LAUNCH(bloom_build_kernel<1>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// Bit-packed integer container decode (internal/encode/bitpack/decode.go:56-74 DecodeValue):
// value i = T((word[i*log2 >> 6] >> (i*log2 & 63) | word[+1] << (64 - shift)) & mask) + minv.
// One lane per value: the packed words are read through L1/L2 (neighbouring lanes share them),
// the materialized column is written with full-width coalesced stores.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) bitpack_decode_kernel(const uint64_t *packed, uint64_t n_words, uint32_t log2, uint64_t minv,
                                                                uint64_t n_values, void *dst) {
    const uint64_t mask = log2 >= 64 ? ~0ull : ((1ull << log2) - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        if (log2 > 0) {
            const uint64_t idx = i * log2, pos = idx >> 6;
            const uint32_t shift = (uint32_t)(idx & 63);
            uint64_t word = packed[pos] >> shift;
            if (64 - shift < log2 && pos + 1 < n_words) word |= packed[pos + 1] << (64 - shift);
            v = word & mask;
        }
        v += minv;
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

hipError_t launch_bitpack_decode(hipStream_t s, int width, const uint64_t *packed, uint64_t n_words, uint32_t log2, uint64_t minv,
                                 uint64_t n_values, void *dst) {
    if (n_values == 0) return hipSuccess;
    uint64_t g = (n_values + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    switch (width) {
    case 8: KNOX_LAUNCH((bitpack_decode_kernel<8>), (int)g, packed, n_words, log2, minv, n_values, dst);
    case 4: KNOX_LAUNCH((bitpack_decode_kernel<4>), (int)g, packed, n_words, log2, minv, n_values, dst);
    case 2: KNOX_LAUNCH((bitpack_decode_kernel<2>), (int)g, packed, n_words, log2, minv, n_values, dst);
    case 1: KNOX_LAUNCH((bitpack_decode_kernel<1>), (int)g, packed, n_words, log2, minv, n_values, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// Scan on bit-packed containers without materializing them (the reference compares on the packed
// form too: internal/encode/bitpack/cmp_*.go, int_bitpack.go Match*).  Value i of a pack is the
// log2-bit field at bit i*log2 of its word stream, plus For (bitpack/decode.go:56-74).
//
// Work shape: one wave per 4096 values; in step t lane l evaluates value t*64 + l, so
//   * the wave64 ballot of the hits IS the bitset word of rows [t*64, t*64+64) -- no bit shuffling;
//     lane t parks word t and after 64 steps the wave stores 64 words = 512 contiguous bytes;
//   * (l*log2) & 63 -- the field's shift -- is the same in every step (t*64*log2 is a multiple of 64
//     bits), only the word index advances by log2 per step; the lanes of a step read log2 + 1
//     consecutive words: one cache-line-sized access.
// Traffic: log2/8 + 1/8 bytes per row instead of 8 + 1/8.
// ---------------------------------------------------------------------------------------------
typedef uint64_t u64x2u __attribute__((ext_vector_type(2), aligned(8)));

template <bool SUM>
__global__ void __launch_bounds__(kBlock) packed_scan_kernel(const PackedScanArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_chunks = (uint64_t)a.n_packs * a.groups_per_pack;
    const uint64_t wave0 = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * kWavesPerBlock;
    const uint64_t n_words_bits = (a.n_rows + 63) >> 6;
    Acc acc;
    acc.init();
    uint64_t cnt = 0;
    for (uint64_t wc = wave0; wc < n_chunks; wc += n_waves) {
        const uint32_t pack = (uint32_t)(wc / a.groups_per_pack);
        const uint64_t base = (wc % a.groups_per_pack) * 4096;  // first value of the chunk inside its pack
        const PackedPack hp = a.packs[pack];
        const uint32_t b = hp.log2;
        const uint64_t mask = b >= 64 ? ~0ull : ((1ull << b) - 1);
        const uint32_t sh = (uint32_t)(lane * b) & 63u, wl = (uint32_t)(lane * b) >> 6;
        const uint64_t *wp = a.words + hp.word_off + (base >> 6) * b + wl;
        const uint64_t c0 = (hp.minv ^ a.key_xor) - a.lo;  // key(field + For) - lo = field + c0 when no carry crosses the sign flip:
                                                            // the flip is an xor of the top bit = an add of 2^63 mod 2^64, so it commutes
        // lane t ends up holding the ballot of step t
        uint32_t my_lo = 0, my_hi = 0;
        uint64_t sum = 0;  // of the FIELDS of the hits; For is added once per chunk: + For * hits
        const bool full = base + 4096 <= hp.n;  // only the last chunk of a short last pack needs the row bound
        // 8 steps per round: the eight 16-byte loads are issued before any of them is consumed
#pragma unroll 1
        for (uint32_t t0 = 0; t0 < 64; t0 += 8) {
            u64x2u w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = __builtin_nontemporal_load((const u64x2u *)(wp + (uint64_t)(t0 + k) * b));
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t t = t0 + k;
                uint64_t field = w[k].x >> sh;
                if (sh != 0) field |= w[k].y << (64 - sh);
                field &= mask;
                bool hit = (((field + c0) & a.wmask) <= a.diff) != (a.negate != 0);
                if (!full) hit = hit && (base + t * 64 + (uint32_t)lane < hp.n);
                const uint64_t word = __ballot(hit);
                // lane t parks the ballot of step t (v_writelane would do it in two instructions, but it takes
                // the lane select from m0 / an inline constant only: measured slower and fragile as inline asm)
                my_lo = lane == (int)t ? (uint32_t)word : my_lo;
                my_hi = lane == (int)t ? (uint32_t)(word >> 32) : my_hi;
                if constexpr (SUM) sum += hit ? field : 0ull;
            }
        }
        const uint64_t myword = ((uint64_t)my_hi << 32) | my_lo;
        const uint64_t wi = (((uint64_t)pack * a.pack_rows + base) >> 6) + (uint64_t)lane;
        if (a.bits != nullptr && wi < n_words_bits) __builtin_nontemporal_store(myword, a.bits + wi);
        const uint64_t hits = (uint64_t)__builtin_popcountll(myword);  // lane t: hits of step t
        cnt += hits;
        if constexpr (SUM) acc.isum += sum + hp.minv * hits;  // wrapping: sum over lanes = sum of fields + For * all hits
    }
    acc.count = cnt;
    block_reduce_store(acc, a.partials + blockIdx.x);
}

hipError_t launch_packed_scan(hipStream_t s, int grid, const PackedScanArgs &a) {
    if (grid <= 0) return hipSuccess;
    if (a.want_sum) KNOX_LAUNCH(packed_scan_kernel<true>, grid, a);
    KNOX_LAUNCH(packed_scan_kernel<false>, grid, a);
}

// ---------------------------------------------------------------------------------------------
// Simple8 containers (internal/encode/int_s8b.go:87-115, s8b/generic/decode.go:15-80, unpack_for.go):
// every 64-bit word carries a 4-bit selector (bits 60..63) and n values of b bits, LSB first;
// selectors 0 / 1 are runs of 128 zeros / ones.  value = field + minv.  A word's position in the
// output is the prefix sum of the counts before it: count -> exclusive scan -> unpack, one lane per word.
// ---------------------------------------------------------------------------------------------
__device__ __constant__ uint8_t kS8bN[16] = {128, 128, 60, 30, 20, 15, 12, 10, 8, 7, 6, 5, 4, 3, 2, 1};
__device__ __constant__ uint8_t kS8bBits[16] = {0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15, 20, 30, 60};

__global__ void __launch_bounds__(kBlock) s8b_count_kernel(const uint64_t *words, uint64_t n_words, uint32_t *counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) counts[i] = kS8bN[words[i] >> 60];
}

template <int W>
__global__ void __launch_bounds__(kBlock) s8b_unpack_kernel(const uint64_t *words, uint64_t n_words, const uint32_t *offsets, uint64_t minv,
                                                            uint64_t n, void *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const uint64_t v = words[i];
    const uint32_t sel = (uint32_t)(v >> 60);
    const uint32_t cnt = kS8bN[sel], b = kS8bBits[sel];
    const uint64_t mask = b == 60 ? 0xFFFFFFFFFFFFFFFull : ((1ull << b) - 1);
    const uint64_t pos = offsets[i];
    for (uint32_t j = 0; j < cnt && pos + j < n; j++) {
        const uint64_t field = sel < 2 ? (uint64_t)sel : ((v >> (j * b)) & mask);
        const uint64_t x = field + minv;
        if constexpr (W == 8) ((uint64_t *)dst)[pos + j] = x;
        else if constexpr (W == 4) ((uint32_t *)dst)[pos + j] = (uint32_t)x;
        else if constexpr (W == 2) ((uint16_t *)dst)[pos + j] = (uint16_t)x;
        else ((uint8_t *)dst)[pos + j] = (uint8_t)x;
    }
}

// scratch: n_words + 1 uint32 (counts -> offsets, total at [n_words]); n_out: device counter
hipError_t launch_s8b_decode(hipStream_t s, int width, const uint64_t *words, uint64_t n_words, uint64_t minv, uint64_t n, uint32_t *scratch,
                             unsigned long long *n_out, void *dst) {
    if (n_words == 0 || n == 0) return hipSuccess;
    const int g = (int)((n_words + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(s8b_count_kernel, dim3(g), dim3(kBlock), 0, s, words, n_words, scratch);
    hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, s, scratch, n_words, n_out);
    switch (width) {
    case 8: KNOX_LAUNCH(s8b_unpack_kernel<8>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    case 4: KNOX_LAUNCH(s8b_unpack_kernel<4>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    case 2: KNOX_LAUNCH(s8b_unpack_kernel<2>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    case 1: KNOX_LAUNCH(s8b_unpack_kernel<1>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// ALP float containers (internal/encode/float_alp.go:109-165, alp/decoder.go:40-130): value i =
// T(enc[i]) * F10[factor] * IF10[exponent] -- two separate multiplications in T's arithmetic, left
// to right (this file is compiled with -ffp-contract=off) -- then the exceptions are patched in.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) alp_decode_kernel(const void *enc, uint64_t f_bits, uint64_t e_bits, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if constexpr (W == 8) {
            const double f = __longlong_as_double((long long)f_bits), e = __longlong_as_double((long long)e_bits);
            double v = (double)((const int64_t *)enc)[i];
            v = v * f;
            v = v * e;
            ((double *)dst)[i] = v;
        } else {
            const float f = __uint_as_float((uint32_t)f_bits), e = __uint_as_float((uint32_t)e_bits);
            float v = (float)((const int32_t *)enc)[i];
            v = v * f;
            v = v * e;
            ((float *)dst)[i] = v;
        }
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) patch_scatter_kernel(const void *patches, const uint32_t *pos, uint64_t n_patch, uint64_t n, void *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_patch) return;
    const uint64_t p = pos[i];
    if (p >= n) return;
    if constexpr (W == 8) ((uint64_t *)dst)[p] = ((const uint64_t *)patches)[i];
    else ((uint32_t *)dst)[p] = ((const uint32_t *)patches)[i];
}

// ALP-RD (float_alprd.go:80-110, alp/rd.go:200-225): IEEE bits = uint(left) << shift | right
template <int W>
__global__ void __launch_bounds__(kBlock) alprd_merge_kernel(const uint16_t *left, const void *right, uint32_t shift, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if constexpr (W == 8) ((uint64_t *)dst)[i] = ((uint64_t)left[i] << shift) | ((const uint64_t *)right)[i];
        else ((uint32_t *)dst)[i] = ((uint32_t)left[i] << shift) | ((const uint32_t *)right)[i];
    }
}

hipError_t launch_alprd_merge(hipStream_t s, int width, const uint16_t *left, const void *right, uint32_t shift, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (width == 8) KNOX_LAUNCH(alprd_merge_kernel<8>, (int)g, left, right, shift, n, dst);
    if (width == 4) KNOX_LAUNCH(alprd_merge_kernel<4>, (int)g, left, right, shift, n, dst);
    return hipErrorInvalidValue;
}

hipError_t launch_alp_decode(hipStream_t s, int width, const void *enc, uint64_t f_bits, uint64_t e_bits, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (width == 8) KNOX_LAUNCH(alp_decode_kernel<8>, (int)g, enc, f_bits, e_bits, n, dst);
    if (width == 4) KNOX_LAUNCH(alp_decode_kernel<4>, (int)g, enc, f_bits, e_bits, n, dst);
    return hipErrorInvalidValue;
}

hipError_t launch_patch_scatter(hipStream_t s, int width, const void *patches, const uint32_t *pos, uint64_t n_patch, uint64_t n, void *dst) {
    if (n_patch == 0) return hipSuccess;
    const int g = (int)((n_patch + kBlock - 1) / kBlock);
    if (width == 8) KNOX_LAUNCH(patch_scatter_kernel<8>, g, patches, pos, n_patch, n, dst);
    if (width == 4) KNOX_LAUNCH(patch_scatter_kernel<4>, g, patches, pos, n_patch, n, dst);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// String containers -> StringPool rows (ptr = off << 32 | len into the column's payload buffer):
// compact (string_compact.go:64-105) row i = buf[ofs[i] : ofs[i] + len[i]], dictionary
// (string_dict.go:70-120) row i = dict[ofs[code[i]] : + len[code[i]]]; `base` = where this container's
// payload starts inside the column's buffer.  ptr_check flags any row that leaves the buffer.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) str_ptr_kernel(const uint32_t *ofs, const uint32_t *len, uint64_t n_entries, const uint32_t *code,
                                                         uint64_t n, uint64_t base, uint64_t *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = code ? (uint64_t)code[i] : i;
        if (k >= n_entries) k = n_entries - 1;  // corrupt code: stay inside the entry table (ptr_check sees the rest)
        uint64_t o = base + ofs[k];
        if (o > 0xffffffffull) o = 0xffffffffull;  // corrupt offset: keep it out of range instead of wrapping (ptr_check flags it)
        dst[i] = (o << 32) | (uint64_t)len[k];
    }
}

__global__ void __launch_bounds__(kBlock) ptr_check_kernel(const uint64_t *ptr, uint64_t n, uint64_t pool_len, uint32_t *flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = ptr[i];
        if ((p >> 32) + (p & 0xffffffffull) > pool_len) atomicOr(flag, 1u);
    }
}

hipError_t launch_str_ptr(hipStream_t s, const uint32_t *ofs, const uint32_t *len, uint64_t n_entries, const uint32_t *code, uint64_t n,
                          uint64_t base, uint64_t *dst) {
    if (n == 0) return hipSuccess;
    if (n_entries == 0) return hipErrorInvalidValue;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    KNOX_LAUNCH(str_ptr_kernel, (int)g, ofs, len, n_entries, code, n, base, dst);
}

hipError_t launch_ptr_check(hipStream_t s, const uint64_t *ptr, uint64_t n, uint64_t pool_len, uint32_t *flag) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    KNOX_LAUNCH(ptr_check_kernel, (int)g, ptr, n, pool_len, flag);
}

// ---------------------------------------------------------------------------------------------
// Dictionary and run-end containers (internal/encode/int_dict.go:76-103, int_runend.go:84-120 and the
// float twins float_dict.go:75-102, float_runend.go:80-110), children already decoded on the device:
//   dictionary  value i = dict[codes[i]]           (codes: uint16)
//   run-end     value i = values[k], k = first run with ends[k] >= i   (ends: uint32, inclusive)
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) dict_gather_kernel(const void *dict, uint64_t n_dict, const uint16_t *codes, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c = codes[i];
        if (c >= n_dict) c = n_dict - 1;  // corrupt input: stay inside the dictionary
        const auto v = load_elem<W>(dict, c);
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) runend_expand_kernel(const void *values, const uint32_t *ends, uint64_t n_runs, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n_runs;  // first k with ends[k] >= i (sort.Search)
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)ends[mid] >= i) hi = mid; else lo = mid + 1;
        }
        if (lo >= n_runs) lo = n_runs - 1;
        const auto v = load_elem<W>(values, lo);
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

hipError_t launch_dict_gather(hipStream_t s, int width, const void *dict, uint64_t n_dict, const uint16_t *codes, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    if (n_dict == 0) return hipErrorInvalidValue;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    switch (width) {
    case 8: KNOX_LAUNCH(dict_gather_kernel<8>, (int)g, dict, n_dict, codes, n, dst);
    case 4: KNOX_LAUNCH(dict_gather_kernel<4>, (int)g, dict, n_dict, codes, n, dst);
    case 2: KNOX_LAUNCH(dict_gather_kernel<2>, (int)g, dict, n_dict, codes, n, dst);
    case 1: KNOX_LAUNCH(dict_gather_kernel<1>, (int)g, dict, n_dict, codes, n, dst);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_runend_expand(hipStream_t s, int width, const void *values, const uint32_t *ends, uint64_t n_runs, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    if (n_runs == 0) return hipErrorInvalidValue;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    switch (width) {
    case 8: KNOX_LAUNCH(runend_expand_kernel<8>, (int)g, values, ends, n_runs, n, dst);
    case 4: KNOX_LAUNCH(runend_expand_kernel<4>, (int)g, values, ends, n_runs, n, dst);
    case 2: KNOX_LAUNCH(runend_expand_kernel<2>, (int)g, values, ends, n_runs, n, dst);
    case 1: KNOX_LAUNCH(runend_expand_kernel<1>, (int)g, values, ends, n_runs, n, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// Constant and delta containers (internal/encode/int_const.go:67-88, int_delta.go:70-95, float_const.go:67-86):
// value i = T(i) * delta + base, wrapping in T (delta = 0: constant; floats: base = the IEEE bits).
// Nothing but three varints crosses PCIe; the column is written at HBM rate.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) affine_fill_kernel(uint64_t base, uint64_t delta, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = i * delta + base;  // low W bytes = T(i) * delta + base in T's arithmetic
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

hipError_t launch_affine_fill(hipStream_t s, int width, uint64_t base, uint64_t delta, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    switch (width) {
    case 8: KNOX_LAUNCH(affine_fill_kernel<8>, (int)g, base, delta, n, dst);
    case 4: KNOX_LAUNCH(affine_fill_kernel<4>, (int)g, base, delta, n, dst);
    case 2: KNOX_LAUNCH(affine_fill_kernel<2>, (int)g, base, delta, n, dst);
    case 1: KNOX_LAUNCH(affine_fill_kernel<1>, (int)g, base, delta, n, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic columns (DESIGN.md "synthetic data"): r(i) = mix64(i + mix64(seed))
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t gen_u64(int kind, uint64_t seedmix, uint64_t i, int64_t lo, uint64_t span) {
    const uint64_t r = mix64(i + seedmix);
    switch (kind) {
    case 0: return span ? (uint64_t)lo + __umul64hi(r, span) : r;
    case 1: return (uint64_t)__double_as_longlong((double)(r >> 44) * 0x1p-10);
    case 2: return (uint64_t)__double_as_longlong((double)(r >> 11) * 0x1p-53);
    case 3: return i + (uint64_t)lo;
    case 4: {
        if ((r & 0xfffff) == 0) {
            const uint64_t sel = (r >> 20) % 5;
            return sel == 0 ? 0x7ff8000000000000ull : sel == 1 ? 0x7ff0000000000000ull : sel == 2 ? 0xfff0000000000000ull
                 : sel == 3 ? 0x8000000000000000ull : 0ull;
        }
        const double d = ((double)(int64_t)(r >> 11) * 0x1p-53 - 0.5) * 0x1p21;
        return (uint64_t)__double_as_longlong(d);
    }
    default: return 0;
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) generate_kernel(int to_f32, int kind, uint64_t seedmix, uint64_t first_row,
                                                          uint64_t n_rows, int64_t lo, uint64_t span, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t u = gen_u64(kind, seedmix, first_row + i, lo, span);
        if constexpr (W == 8) ((uint64_t *)dst)[i] = u;
        else if constexpr (W == 4) {
            if (to_f32) {
                const float f = (kind == 1 || kind == 2 || kind == 4) ? (float)__longlong_as_double((long long)u) : (float)(int64_t)u;
                ((float *)dst)[i] = f;
            } else ((uint32_t *)dst)[i] = (uint32_t)u;
        } else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)u;
        else ((uint8_t *)dst)[i] = (uint8_t)u;
    }
}

hipError_t launch_generate(hipStream_t s, int width, bool to_f32, int kind, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                           int64_t lo, uint64_t span, void *dst) {
    if (n_rows == 0) return hipSuccess;
    uint64_t g = (n_rows + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    const uint64_t sm = mix64(seed);
    switch (width) {
    case 8: KNOX_LAUNCH((generate_kernel<8>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    case 4: KNOX_LAUNCH((generate_kernel<4>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    case 2: KNOX_LAUNCH((generate_kernel<2>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    case 1: KNOX_LAUNCH((generate_kernel<1>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    default: return hipErrorInvalidValue;
    }
}

// synthetic bytes column (knox_kernels.h launch_generate_bytes; DESIGN.md "synthetic data" kind 5): one lane per row
__global__ void __launch_bounds__(kBlock) generate_bytes_kernel(uint64_t seedmix, uint64_t keymix, uint64_t first_row, uint64_t n_rows,
                                                                uint64_t pack_rows, uint32_t key_len, uint32_t window_bits,
                                                                uint32_t packs_per_window, uint64_t pool_stride, uint64_t *ptr, uint8_t *buf) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = first_row + i;
        const uint64_t pack = row / pack_rows, in = row - pack * pack_rows;
        const uint64_t k = ((pack / packs_per_window) << window_bits) | (mix64(row + seedmix) & ((1ull << window_bits) - 1ull));
        uint8_t *dst = buf + (i / pack_rows) * pool_stride + in * key_len;
        for (uint32_t j = 0; j * 8u < key_len; j++) {
            const uint64_t w = mix64(8ull * k + j + keymix);
            const uint32_t nb = key_len - j * 8u < 8u ? key_len - j * 8u : 8u;
            if ((key_len & 7u) == 0u) { ((uint64_t *)dst)[j] = w; continue; }  // pools are 64-byte aligned: whole words
            for (uint32_t b = 0; b < nb; b++) dst[j * 8u + b] = (uint8_t)(w >> (8u * b));
        }
        ptr[i] = ((in * key_len) << 32) | key_len;
    }
}

hipError_t launch_generate_bytes(hipStream_t s, uint64_t seed, uint64_t first_row, uint64_t n_rows, uint64_t pack_rows, uint32_t key_len,
                                 uint32_t window_bits, uint32_t packs_per_window, uint64_t pool_stride, uint64_t *ptr, uint8_t *buf) {
    if (n_rows == 0) return hipSuccess;
    if (pack_rows == 0 || first_row % pack_rows || packs_per_window == 0 || window_bits > 40) return hipErrorInvalidValue;
    uint64_t g = (n_rows + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    KNOX_LAUNCH(generate_bytes_kernel, (int)g, mix64(seed), mix64(seed ^ 0x6b657973ull), first_row, n_rows, pack_rows, key_len, window_bits,
                packs_per_window, pool_stride, ptr, buf);
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

// ---------------------------------------------------------------------------------------------
// bytes REGEXP (operator/filter/match_bytes.go:587-609: re.Match(row) for every row).  The host
// compiles the pattern into a DFA over rune classes (knox_regex.h); one lane = one row: decode a
// rune the way Go does (utf8.DecodeRune, invalid byte = U+FFFD of width 1), map it to its class
// (table for ASCII, binary search above), step table[state][class]; state 0 = ACCEPT (absorbing),
// flags bit1 = dead (no accepting state reachable) -> the lane stops early either way.  Tables of
// up to 48 KiB are staged in LDS, larger ones are read through L2.
// ---------------------------------------------------------------------------------------------
template <bool LDS, bool DIRECT>
__global__ void __launch_bounds__(kBlock) regexp_match_kernel(RegexProg pg, BytesView v, uint8_t *bits, const uint8_t *mask) {
    extern __shared__ __align__(16) uint8_t rx_lds[];
    const uint16_t *table = pg.table, *direct = pg.direct;
    const uint8_t *flags = pg.flags, *ascii = pg.ascii_class;
    if constexpr (LDS) {
        // image layout (host): [direct u16 * n_states*128 (DIRECT only)][table u16 * n_states*n_classes][flags n_states][ascii 128]
        const uint32_t db = DIRECT ? pg.n_states * 256u : 0u;
        const uint32_t tb = pg.n_states * pg.n_classes * 2, total = db + tb + pg.n_states + 128;
        const uint8_t *image = DIRECT ? (const uint8_t *)pg.direct : (const uint8_t *)pg.table;
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) rx_lds[i] = image[i];
        __syncthreads();
        direct = (const uint16_t *)rx_lds;
        table = (const uint16_t *)(rx_lds + db);
        flags = rx_lds + db + tb;
        ascii = flags + pg.n_states;
    }
    // the class intervals of the runes above ASCII: staged too when they are few (they almost always are)
    const uint32_t *bound_lo = pg.bound_lo;
    const uint8_t *bound_class = pg.bound_class;
    if constexpr (LDS) {
        if (pg.n_bounds <= 96u) {
            const uint32_t base = (pg.lds_image_bytes + 3u) & ~3u;
            uint32_t *blo = (uint32_t *)(rx_lds + base);
            uint8_t *bcl = rx_lds + base + 4u * pg.n_bounds;
            for (uint32_t i = threadIdx.x; i < pg.n_bounds; i += blockDim.x) { blo[i] = pg.bound_lo[i]; bcl[i] = pg.bound_class[i]; }
            __syncthreads();
            bound_lo = blo;
            bound_class = bcl;
        }
    }
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    const uint64_t n_rows = v.n_rows;
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row0 = bv_group_row0(v, g);
        const uint8_t *pool = bv_pool(v, row0);
        const uint64_t row = row0 + lane;
        bool m = false;
        if (row < n_rows && (mask == nullptr || ((mask[row >> 3] >> (row & 7)) & 1))) {
            const uint64_t p = v.ptr[row];
            const uint8_t *s = pool + (uint32_t)(p >> 32);
            const uint32_t len = (uint32_t)p;
            uint32_t state = 1, pos = 0;
            bool decided = false;
            // the first 32 bytes of the row are fetched at once (two unaligned 16-byte loads; the pool is padded) and
            // the 4-byte decode window is cut out of registers: one HBM round trip per row instead of one per 5 bytes.
            // Longer rows continue in 8-byte pieces.
            uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            if (len != 0) {
                u32x4 a, b;
                __builtin_memcpy(&a, s, 16);
                __builtin_memcpy(&b, s + 16, 16);
                w0 = ((uint64_t)a.y << 32) | a.x; w1 = ((uint64_t)a.w << 32) | a.z;
                w2 = ((uint64_t)b.y << 32) | b.x; w3 = ((uint64_t)b.w << 32) | b.z;
                if constexpr (DIRECT) {
                    // ASCII prefix of the row, one preloaded dword at a time (static register indexing, no shifts): up to
                    // four table steps per dword; ACCEPT (0) and the dead states are absorbing, so they are looked at
                    // once per dword.  The walk is VALU-issue bound (wave64: 4 cycles per instruction).  A partial last
                    // dword still takes four lookups (bytes past the row masked to 7 bits) and keeps the right one.
                    const uint32_t w32[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (!decided && pos == 4u * k && pos < len) {
                            const uint32_t win = w32[k], rem = len - pos;
                            const uint32_t hm = rem >= 4u ? 0x80808080u : (0x00808080u >> (8u * (3u - rem)));
                            if ((win & hm) == 0u) {
                                const uint32_t s1 = direct[state * 128u + (win & 0x7fu)];
                                const uint32_t s2 = direct[s1 * 128u + ((win >> 8) & 0x7fu)];
                                const uint32_t s3 = direct[s2 * 128u + ((win >> 16) & 0x7fu)];
                                const uint32_t s4 = direct[s3 * 128u + ((win >> 24) & 0x7fu)];
                                state = rem >= 4u ? s4 : rem == 3u ? s3 : rem == 2u ? s2 : s1;
                                pos += rem < 4u ? rem : 4u;
                                if (state == 0u) { m = true; decided = true; }
                                else if (flags[state] & 2) decided = true;
                            }
                        }
                    }
                }
            }
            uint64_t piece = 0;
            uint32_t pbase = 0x80000000u;
            while (!decided && pos < len) {
                uint32_t win;
                if (pos <= 28u) {
                    const uint32_t idx = pos >> 3, sh = (pos & 7u) * 8u;
                    const uint64_t lo = idx == 0 ? w0 : idx == 1 ? w1 : idx == 2 ? w2 : w3;
                    const uint64_t hi = idx == 0 ? w1 : idx == 1 ? w2 : w3;  // idx 3: pos <= 28 keeps the window inside w3
                    win = (uint32_t)(lo >> sh);
                    if (sh > 32u) win |= (uint32_t)(hi << (64u - sh));
                } else {
                    if (pos - pbase > 4u) {
                        __builtin_memcpy(&piece, s + pos, 8);
                        pbase = pos;
                    }
                    win = (uint32_t)(piece >> (8 * (pos - pbase)));
                }
                if constexpr (DIRECT) {
                    // four ASCII bytes inside the row: four table steps back to back.  ACCEPT (state 0) and the dead
                    // states are absorbing, so they are only looked at once per word -- the walk is VALU-issue bound
                    // (wave64: 4 cycles per instruction), every instruction per rune counts
                    if (pos + 4u <= len && (win & 0x80808080u) == 0u) {
                        state = direct[state * 128u + (win & 0xffu)];
                        state = direct[state * 128u + ((win >> 8) & 0xffu)];
                        state = direct[state * 128u + ((win >> 16) & 0xffu)];
                        state = direct[state * 128u + (win >> 24)];
                        pos += 4u;
                        if (state == 0u) { m = true; decided = true; break; }
                        if (flags[state] & 2) { decided = true; break; }
                        continue;
                    }
                }
                uint32_t w;
                const uint32_t r = knox_regex::decode_rune_word(win, len - pos, w);
                pos += w;
                uint32_t e;
                if (DIRECT && r < 0x80) {
                    e = direct[state * 128u + r];  // plain target state (no dead bit in the direct table)
                    if (e != 0 && (flags[e] & 2)) e |= knox_regex::kDeadBit;
                } else {
                    uint32_t cls;
                    if (r < 0x80) cls = ascii[r];
                    else {
                        uint32_t lo = 0, hi = pg.n_bounds;
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (bound_lo[mid] <= r) lo = mid; else hi = mid;
                        }
                        cls = bound_class[lo];
                    }
                    e = table[state * pg.n_classes + cls];
                }
                if (e == 0) { m = true; decided = true; break; }
                if (e & knox_regex::kDeadBit) { decided = true; break; }
                state = e;
            }
            if (!decided) m = flags[state] & 1;
        }
        const uint64_t ballot = __ballot(m);  // unmasked rows never match: with a mask this is bits.Set(i) for the masked rows that do
        bv_store_group(ballot, row0, n_rows, lane, bits, nullptr, mask != nullptr);
    }
}

hipError_t launch_regexp_match(hipStream_t s, const RegexProg &pg, uint32_t image_bytes, const BytesView &v, uint8_t *bits, const uint8_t *mask) {
    const uint64_t ngroups = v.live ? (uint64_t)v.n_live * (v.pack_rows >> 6) : (v.n_rows + 63) >> 6;
    if (ngroups == 0) return hipSuccess;
    uint64_t g = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > 2048 * 4) g = 2048 * 4;
    const uint32_t lds_bytes = ((image_bytes + 3u) & ~3u) + 96u * 5u + 16u;  // image + (<= 96) staged class intervals
    if (pg.direct != nullptr)  // host guarantees: the image with the direct table fits the LDS budget
        hipLaunchKernelGGL((regexp_match_kernel<true, true>), dim3((unsigned)g), dim3(kBlock), lds_bytes, s, pg, v, bits, mask);
    else if (image_bytes <= 48 * 1024)
        hipLaunchKernelGGL((regexp_match_kernel<true, false>), dim3((unsigned)g), dim3(kBlock), lds_bytes, s, pg, v, bits, mask);
    else
        hipLaunchKernelGGL((regexp_match_kernel<false, false>), dim3((unsigned)g), dim3(kBlock), 0, s, pg, v, bits, mask);
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

hipError_t launch_bucket_sequential(hipStream_t s, int dtype, int func, const int64_t *time, const void *col, const uint8_t *mask,
                                    const uint64_t *start, uint32_t n_buckets, void *out_v, uint8_t *out_seen) {
    const int tpb = 64;
    const int grid = (int)((n_buckets + tpb - 1) / tpb);
#define KNOX_BSEQ(T) hipLaunchKernelGGL((bucket_seq_kernel<T>), dim3(grid), dim3(tpb), 0, s, func, time, (const T *)col, mask, start, n_buckets, (T *)out_v, out_seen); break
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
