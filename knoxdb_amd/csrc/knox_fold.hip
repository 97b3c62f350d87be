// Time windows over out-of-order timestamps WITHOUT a sort (internal/reducer/bucket_native.go:123-141: Push finds a row's window and
// pushes the value into that window's reducer): wherever a reducer's fold does not depend on the order the rows arrive in, the rows
// are folded where they lie -- per-workgroup tables in LDS, LDS atomics, one streaming walk over the clock and the values.
//   Value() of count / first / last / min / max / integer sum / mean   window_direct_kernel
//   float sums / means, exact (fixed-point adders)                     window_fsum_kernel
//   float var / std, exact (sums of values and squares)                window_fvar_kernel
//   whole states, integer / float columns                             window_state_kernel / window_fstate_kernel
// What needs an order or more windows than LDS holds goes through knox_sort.hip.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "knox_finalize.h"
#include "knox_kernels.h"
#include "knox_window_dev.h"

namespace knox {

// ---- no sort at all: Value() per window over out-of-order time ---------------------------------------------------------------------
// NativeBucket.Push (bucket_native.go:123-141) finds a row's window and pushes the value into that window's reducer.  Where the
// reducer's fold does not depend on the order the rows arrive in -- count, the wrapping integer sum, the EXACT integer sum the mean is
// rounded from, the integer minimum / maximum, and first / last once they are read as "value of the lowest / highest row id" -- that
// is all there is to do, and the sort only moves bytes.  As long as the table fits LDS: every workgroup keeps one 64-bit slot (two
// for the exact sum of 8-byte values: low words and sign-correct high words, S = A + B * 2^32) and a count -- or one "seen" bit --
// per window, walks the column 512 rows per wave and round (the compaction pass's loads: two neighbouring rows per lane and piece,
// everything asked for up front), folds with LDS atomics (add / min / max: one or two 64-bit ones and at most one 32-bit one per
// selected row) and writes its table; one lane per window folds the tables and emits Value() through knox_finalize.h (first / last:
// the value is fetched from the winning row there).
// Round 4 had built the same for EVERY state field at once (eight 64-bit atomics and a 64-bit division per row, two passes: 4.07 ms
// for 2^28 rows -> 1024 windows against the sort's 2.88) and dropped it; with the division gone the walk runs at streaming rate
// (0.67-0.84 ms, with or without a selection, 1024 .. 16384 windows) and it is the sort that loses.  Float sums / means and var / std
// have their own kernels below (a double-double sum is not an atomic, an exact fixed-point integer is); the sort keeps the window counts
// whose table does not fit and the states that carry var / std moments.
enum { kDirectSum = 0, kDirectFirst = 1, kDirectLast = 2, kDirectMin = 3, kDirectMax = 4, kDirectMean = 5, kDirectCount = 8,   // == knox_reducer
       kDirectFMin = 16, kDirectFMax = 17 };   // min / max of a FLOAT column (below)
constexpr int kDirectBlock = 1024;
constexpr int kDirectWaves = kDirectBlock / 64;

// the walk all sort-free kernels share: 512 rows per wave and round (the compaction pass's loads: two neighbouring rows per lane and
// piece, everything asked for up front, values only where the pair holds a selected row), chunks dealt round-robin to the waves of the
// grid; push(timestamp, value, row) for every selected row; the rows behind the last whole chunk by workgroup 0
template <typename U, bool VALUES, typename Push>
__device__ __forceinline__ void direct_walk(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask, uint64_t n_rows, Push push,
                                            uint32_t sub = blockIdx.x, uint32_t nsub = gridDim.x) {      // (sub of nsub: the workgroups that share the walk)
    typedef long long t2 __attribute__((ext_vector_type(2)));
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t sh = (2u * lane) & 7u;
    const uint64_t n_chunks = n_rows / 512;
    for (uint64_t c = (uint64_t)sub * kDirectWaves + wave; c < n_chunks; c += (uint64_t)nsub * kDirectWaves) {
        const uint64_t c0 = c * 512;
        uint32_t mb[4];
        t2 t[4];
        CompactPair<U> x[4];
#pragma unroll
        for (int h = 0; h < 4; h++) mb[h] = mask != nullptr ? mask[(c0 + 128u * h + 2u * lane) >> 3] : 0xffu;
#pragma unroll
        for (int h = 0; h < 4; h++) t[h] = __builtin_nontemporal_load(reinterpret_cast<const t2 *>(time + c0 + 128u * h + 2u * lane));
#pragma unroll
        for (int h = 0; h < 4; h++) {
            x[h] = CompactPair<U>{0, 0};
            if constexpr (VALUES) {
                if (mask == nullptr || ((mb[h] >> sh) & 3u)) x[h] = __builtin_nontemporal_load(reinterpret_cast<const CompactPair<U> *>(val + c0 + 128u * h + 2u * lane));
            }
        }
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint32_t m2 = (mb[h] >> sh) & 3u;
            const uint64_t row = c0 + 128u * h + 2u * lane;
            if (m2 & 1u) push(t[h].x, x[h].x, row);
            if (m2 & 2u) push(t[h].y, x[h].y, row + 1);
        }
    }
    if (sub == 0) {
        for (uint64_t r = n_chunks * 512 + threadIdx.x; r < n_rows; r += kDirectBlock)
            if (mask == nullptr || ((mask[r >> 3] >> (r & 7)) & 1u)) push(time[r], VALUES ? val[r] : (U)0, r);
    }
}

// what a mode keeps per window
// Float min / max (reducer.go:256-296 as acc_update restates it, knox_dev.h): a NaN is never adopted, -0 and +0 compare equal (ONE
// order key), the first occurrence wins, and a NaN as the window's FIRST value sticks.  Value() needs no row id, so the order only
// matters where equal keys have different bits -- the two zeros: the slot folds the order key, one word the lowest selected row (its
// value decides the NaN rule), another the lowest row holding a zero (read only when the winning key is zero's).
template <int MODE, int W> struct DirectShape {
    static constexpr bool FMINMAX = MODE == kDirectFMin || MODE == kDirectFMax;
    static constexpr int NS = MODE == kDirectCount ? 0 : (MODE == kDirectMean && W == 8) ? 2 : 1;              // 64-bit slots
    static constexpr bool COUNTS = MODE == kDirectCount || MODE == kDirectMean;                                 // a 32-bit count
    static constexpr bool BITS = MODE == kDirectSum || MODE == kDirectMin || MODE == kDirectMax;                // a "seen" bit
    static constexpr bool VALUES = MODE == kDirectSum || MODE == kDirectMean || MODE == kDirectMin || MODE == kDirectMax || FMINMAX;   // the walk reads the values
    static constexpr uint64_t INIT = (MODE == kDirectMin || MODE == kDirectFirst || MODE == kDirectFMin) ? ~0ull : 0ull;   // first: lowest row; last: highest row + 1
    static constexpr uint32_t WINIT = FMINMAX ? ~0u : 0u;
    static constexpr int FUNC = MODE == kDirectFMin ? kDirectMin : MODE == kDirectFMax ? kDirectMax : MODE;     // the knox_reducer Value() is asked for
    __host__ __device__ static constexpr uint32_t words(uint32_t nb) { return FMINMAX ? 2 * nb : COUNTS ? nb : BITS ? (nb + 31) / 32 : 0; }
};

template <typename U, bool SIGNED, int MODE>
__global__ void __launch_bounds__(kDirectBlock) window_direct_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                     uint64_t n_rows, int64_t t_from, int64_t step, uint32_t nb_all, uint32_t cap,
                                                                     uint32_t n_ranges, uint64_t *__restrict__ slots_out, uint32_t *__restrict__ words_out) {
    using SH = DirectShape<MODE, (int)sizeof(U)>;
    constexpr int NS = SH::NS;
    // More windows than one table holds: the windows are cut into n_ranges ranges of `cap`, workgroup b folds range b % n_ranges -- a
    // window grid of its own (its first edge, its count) -- and the workgroups of a range share one walk over ALL rows: the column
    // is read n_ranges times, which up to four still beats the sort's passes.
    const uint32_t rid = blockIdx.x % n_ranges, sub = blockIdx.x / n_ranges, nsub = gridDim.x / n_ranges;
    t_from += (int64_t)((uint64_t)rid * cap) * step;
    const uint32_t nb = cap;                                                    // the table's stride; the range's last windows may lie behind nb_all
    const uint32_t nbr = rid * cap >= nb_all ? 0u : (nb_all - rid * cap < cap ? nb_all - rid * cap : cap);
    extern __shared__ uint64_t direct_tab[];           // NS x nb slots, then the counts / bits
    uint64_t *s0 = direct_tab, *s1 = s0 + nb;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(direct_tab + (size_t)NS * nb);
    const uint32_t nc = SH::words(nb);
    for (uint32_t i = threadIdx.x; i < NS * nb; i += kDirectBlock) direct_tab[i] = SH::INIT;
    for (uint32_t i = threadIdx.x; i < nc; i += kDirectBlock) cnt[i] = SH::WINIT;
    __syncthreads();
    const uint64_t span = (uint64_t)nbr * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    auto push = [&](int64_t t, U x, uint64_t row) {
        const uint32_t q = window_index(t, t_from, (uint64_t)step, span, inv);
        if (q == kNoKey) return;
        if constexpr (SH::FMINMAX) {
            constexpr U SIGN = (U)1 << (8 * sizeof(U) - 1);
            constexpr U EXPMASK = sizeof(U) == 8 ? (U)0x7ff0000000000000ull : (U)0x7f800000u;
            const U mag = x & (U)~SIGN;
            atomicMin(&cnt[q], (uint32_t)row);
            if (mag <= EXPMASK) {            // not a NaN
                const U fk = mag == 0 ? SIGN : ((x & SIGN) ? (U)~x : (U)(x | SIGN));
                unsigned long long *pk = reinterpret_cast<unsigned long long *>(&s0[q]);
                if constexpr (MODE == kDirectFMin) atomicMin(pk, (unsigned long long)fk);
                else atomicMax(pk, (unsigned long long)fk);
                if (mag == 0) atomicMin(&cnt[nb + q], (uint32_t)row);
            }
            return;
        }
        if constexpr (SH::BITS) atomicOr(&cnt[q >> 5], 1u << (q & 31));      // Value() asks only whether the window saw a row
        if constexpr (SH::COUNTS) atomicAdd(&cnt[q], 1u);
        unsigned long long *p0 = reinterpret_cast<unsigned long long *>(&s0[q]);
        if constexpr (MODE == kDirectSum || MODE == kDirectMean) {
            // the value in 64-bit two's complement; as a SUM it wraps to the column's width in Value() (`isum & m`), where the signed and
            // the zero-extended sum agree, and below 8 bytes it is exact as it stands
            const uint64_t sx = SIGNED ? (uint64_t)(int64_t)(typename std::make_signed<U>::type)x : (uint64_t)x;
            if constexpr (NS == 2) {
                atomicAdd(p0, (unsigned long long)(sx & 0xffffffffull));
                atomicAdd(reinterpret_cast<unsigned long long *>(&s1[q]), (unsigned long long)(SIGNED ? (uint64_t)((int64_t)sx >> 32) : (sx >> 32)));
            } else {
                atomicAdd(p0, (unsigned long long)sx);
            }
        } else if constexpr (MODE == kDirectMin || MODE == kDirectMax) {
            const uint64_t key = SIGNED ? ((uint64_t)x ^ (1ull << (8 * sizeof(U) - 1))) : (uint64_t)x;      // order key of the value, zero-extended
            if constexpr (MODE == kDirectMin) atomicMin(p0, (unsigned long long)key);
            else atomicMax(p0, (unsigned long long)key);
        } else if constexpr (MODE == kDirectFirst) {
            atomicMin(p0, (unsigned long long)row);
        } else if constexpr (MODE == kDirectLast) {
            atomicMax(p0, (unsigned long long)(row + 1));
        }
    };
    direct_walk<U, SH::VALUES>(time, val, mask, n_rows, push, sub, nsub);
    __syncthreads();
    uint64_t *so = slots_out + (size_t)blockIdx.x * NS * nb;
    uint32_t *co = words_out + (size_t)blockIdx.x * nc;
    for (uint32_t i = threadIdx.x; i < NS * nb; i += kDirectBlock) so[i] = direct_tab[i];
    for (uint32_t i = threadIdx.x; i < nc; i += kDirectBlock) co[i] = cnt[i];
}

// one lane per window: the workgroups' tables folded, Value() out (the state's other fields are never read for these reducers)
template <typename U, bool SIGNED, int MODE>
__global__ void __launch_bounds__(256) window_direct_finish_kernel(const uint64_t *__restrict__ slots, const uint32_t *__restrict__ words, uint32_t grid,
                                                                   uint32_t nb_all, uint32_t cap, uint32_t n_ranges, const U *__restrict__ val, int dtype,
                                                                   void *values, uint8_t *seen) {
    using SH = DirectShape<MODE, (int)sizeof(U)>;
    constexpr int NS = SH::NS;
    // 64 windows per workgroup, the tables dealt to its four waves (a lane per window and table in turn: 512 dependent-free but
    // serial loads per lane took 108 us for 16384 windows, a seventh of the walk itself); the waves' folds meet in LDS
    __shared__ uint64_t f_cnt[4][64], f_a[4][64], f_b[4][64];
    __shared__ uint32_t f_fr[4][64], f_zr[4][64];
    const uint32_t lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const uint32_t kw = blockIdx.x * 64 + lane;                  // the window; k: its place in its range's tables
    const bool valid = kw < nb_all;
    const uint32_t nb = cap, rid = kw / cap, k = kw - rid * cap, nsub = grid / n_ranges;
    const uint32_t nc = SH::words(nb);
    uint64_t count = 0, a = SH::INIT, b = 0;
    uint32_t frow = ~0u, zrow = ~0u;      // float min / max: the lowest selected row, the lowest row holding a zero
    auto fold = [&](uint64_t c2, uint64_t a2, uint64_t b2, uint32_t f2, uint32_t z2) {
        if constexpr (SH::BITS) count |= c2; else count += c2;
        if constexpr (MODE == kDirectMin || MODE == kDirectFirst || MODE == kDirectFMin) a = a2 < a ? a2 : a;
        else if constexpr (MODE == kDirectMax || MODE == kDirectLast || MODE == kDirectFMax) a = a2 > a ? a2 : a;
        else a += a2;
        b += b2;                               // two's complement: the signed high parts add as unsigned words
        frow = f2 < frow ? f2 : frow;
        zrow = z2 < zrow ? z2 : zrow;
    };
    if (valid) {
#pragma unroll 4
        for (uint32_t sb = part; sb < nsub; sb += 4) {
            const uint32_t g = sb * n_ranges + rid;
            uint64_t c2 = 0, a2 = SH::INIT, b2 = 0;
            uint32_t f2 = ~0u, z2 = ~0u;
            if constexpr (SH::FMINMAX) { f2 = words[(size_t)g * nc + k]; z2 = words[(size_t)g * nc + nb + k]; }
            if constexpr (SH::BITS) c2 = (words[(size_t)g * nc + (k >> 5)] >> (k & 31)) & 1u;     // (0 / 1: "seen")
            if constexpr (SH::COUNTS) c2 = words[(size_t)g * nc + k];
            if constexpr (NS >= 1) a2 = slots[((size_t)g * NS + 0) * nb + k];
            if constexpr (NS == 2) b2 = slots[((size_t)g * NS + 1) * nb + k];
            fold(c2, a2, b2, f2, z2);
        }
    }
    f_cnt[part][lane] = count; f_a[part][lane] = a; f_b[part][lane] = b; f_fr[part][lane] = frow; f_zr[part][lane] = zrow;
    __syncthreads();
    if (part != 0 || !valid) return;
#pragma unroll
    for (int w = 1; w < 4; w++) fold(f_cnt[w][lane], f_a[w][lane], f_b[w][lane], f_fr[w][lane], f_zr[w][lane]);
    AggStateDev s;
    s.dtype = dtype;
    s.flags = 1u;
    s.isum = 0;
    s.fsum_hi = 0.0;
    s.fsum_lo = 0.0;
    s.min_bits = s.max_bits = s.first_bits = s.last_bits = 0;
    s.min_row = s.max_row = s.first_row = s.last_row = -1;
    s.m_shift = 0.0; s.m_dev_hi = 0.0; s.m_dev_lo = 0.0; s.m_sq_hi = 0.0; s.m_sq_lo = 0.0;
    if constexpr (MODE == kDirectSum || MODE == kDirectMean) {
        uint64_t xlo = a;
        int64_t xhi = SIGNED ? ((int64_t)a >> 63) : 0;       // one adder: the exact sum of values below 8 bytes (or a wrapped one nobody reads)
        if constexpr (NS == 2) {
            const uint64_t blo = b << 32;
            const int64_t bhi = SIGNED ? ((int64_t)b >> 32) : (int64_t)(b >> 32);
            xlo = a + blo;
            xhi = bhi + (int64_t)(xlo < a);
        }
        s.isum = xlo;
        knox_fin::int128_to_dd(xlo, xhi, s.fsum_hi, s.fsum_lo);
    } else if constexpr (MODE == kDirectMin || MODE == kDirectMax) {
        const uint64_t raw = SIGNED ? (a ^ (1ull << (8 * sizeof(U) - 1))) : a;      // the key map is its own inverse
        s.min_bits = s.max_bits = raw;
        s.min_row = s.max_row = count ? 0 : -1;
    } else if constexpr (SH::FMINMAX) {
        constexpr uint64_t SIGN = 1ull << (8 * sizeof(U) - 1);
        count = frow != ~0u;
        if (count) {
            s.first_bits = (uint64_t)val[frow];
            if (a != SH::INIT) {          // a value that is not a NaN was seen
                const uint64_t raw = a == SIGN ? (uint64_t)val[zrow] : ((a & SIGN) ? (a ^ SIGN) : (uint64_t)(U)~(U)a);
                s.min_bits = s.max_bits = raw;
                s.min_row = s.max_row = 0;
            }
        }
    } else if constexpr (MODE == kDirectFirst) {
        count = a != ~0ull;
        if (count) { s.first_bits = (uint64_t)val[a]; }
    } else if constexpr (MODE == kDirectLast) {
        count = a != 0;
        if (count) { s.last_bits = (uint64_t)val[a - 1]; s.last_row = (int64_t)(a - 1); }
    }
    s.count = (int64_t)count;
    s.first_row = count ? (MODE == kDirectFirst ? (int64_t)a : SH::FMINMAX ? (int64_t)frow : 0) : -1;
    uint64_t bits = 0;
    int sn = 0;
    int64_t row = -1;
    const int rc = knox_fin::agg_value_bits(s, SH::FUNC, &bits, &sn, &row);
    reinterpret_cast<U *>(values)[kw] = (U)bits;
    seen[kw] = (rc == 0 && sn) ? 1 : 0;
}

// grid == 0: the reducer has no order-free form for this column type, or the windows' table does not fit a workgroup's LDS (the
// caller sorts).  (ONE table in device memory for any number of windows was built as well: device-scope atomics of the eight XCDs
// on shared lines ran at 21 G adds/s -- 2^28 rows -> 16384 windows 25.5 ms against the sort's 5.2 -- dropped.)
WindowDirectGeom window_direct_geom(int width, int dtype, int func, uint32_t n_buckets, int n_cus, bool masked) {
    WindowDirectGeom g;
    g.grid = 0;
    g.n_slots = g.n_words = g.cap = 0;
    g.n_ranges = 1;
    g.lds = g.scratch = 0;
    const bool integer = dtype >= 1 && dtype <= 8, numeric = dtype >= 1 && dtype <= 10;
    const bool fminmax = !integer && (func == kDirectMin || func == kDirectMax);
    const bool ok = (func == kDirectCount || func == kDirectFirst || func == kDirectLast || func == kDirectMin || func == kDirectMax) ? numeric
                  : (func == kDirectSum || func == kDirectMean) ? integer : false;
    if (!ok || (width != 8 && width != 4 && width != 2 && width != 1)) return g;
    g.n_slots = func == kDirectCount ? 0 : (func == kDirectMean && width == 8) ? 2 : 1;
    auto words_of = [&](uint32_t nb) {
        return fminmax ? 2 * nb : (func == kDirectCount || func == kDirectMean) ? nb : (func == kDirectFirst || func == kDirectLast) ? 0u : (nb + 31) / 32;
    };
    auto lds_of = [&](uint32_t nb) { return ((size_t)nb * 8u * g.n_slots + (size_t)words_of(nb) * 4u + 7) & ~(size_t)7; };
    // more windows than one table holds: up to four ranges without a selection, two behind one (every range reads all rows: measured against
    // the sort's passes, 2^28 rows: 0.7 ms per range, the sort 5-7 ms for every row and 2.6-3.6 ms at 25 % selected)
    const uint32_t max_ranges = masked ? 2u : 4u;
    uint32_t r = 1;
    uint32_t cap = n_buckets;
    while (lds_of(cap) > 156 * 1024) {
        if (++r > max_ranges) return g;
        cap = (((n_buckets + r - 1) / r) + 31) & ~31u;
    }
    g.n_ranges = r;
    g.cap = cap;
    g.n_words = words_of(cap);
    g.lds = lds_of(cap);
    g.grid = (uint32_t)n_cus * (g.lds <= 40 * 1024 ? 2u : 1u);
    g.grid = g.grid / r * r;
    g.scratch = (size_t)g.grid * ((size_t)cap * 8u * g.n_slots + (size_t)g.n_words * 4u) + 512;
    return g;
}

// scratch: g.scratch bytes, 8-byte aligned
hipError_t launch_window_direct(hipStream_t s, int width, int dtype, int func, WindowDirectGeom g, const int64_t *time, const void *val, const uint8_t *mask,
                                uint64_t n_rows, int64_t t_from, int64_t step, uint32_t n_buckets, void *scratch, void *values, uint8_t *seen) {
    if (n_buckets == 0 || g.grid == 0) return hipErrorInvalidValue;
    const bool is_signed = dtype >= 1 && dtype <= 4, is_float = dtype == 9 || dtype == 10;
    uint64_t *slots = (uint64_t *)scratch;
    uint32_t *words = (uint32_t *)(slots + (size_t)g.grid * g.n_slots * g.cap);
    const dim3 fgrid((n_buckets + 63) / 64), fblock(256);
#define KNOX_DIRECT(U, S, MODE)                                                                                                                     \
    do {                                                                                                                                            \
        if (g.lds > 48 * 1024)                                                                                                                      \
            (void)hipFuncSetAttribute((const void *)window_direct_kernel<U, S, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);      \
        hipLaunchKernelGGL((window_direct_kernel<U, S, MODE>), dim3(g.grid), dim3(kDirectBlock), g.lds, s, time, (const U *)val, mask, n_rows, t_from, step, \
                           n_buckets, g.cap, g.n_ranges, slots, words);                                                                             \
        hipLaunchKernelGGL((window_direct_finish_kernel<U, S, MODE>), fgrid, fblock, 0, s, (const uint64_t *)slots, (const uint32_t *)words, g.grid, \
                           n_buckets, g.cap, g.n_ranges, (const U *)val, dtype, values, seen);                                                      \
    } while (0)
#define KNOX_DIRECT_S(U, MODE)                                                                \
    do {                                                                                      \
        if (is_signed) KNOX_DIRECT(U, true, MODE); else KNOX_DIRECT(U, false, MODE);          \
    } while (0)
#define KNOX_DIRECT_W(U)                                                                      \
    do {                                                                                      \
        switch (func) {                                                                       \
        case kDirectCount: KNOX_DIRECT(U, false, kDirectCount); break;                        \
        case kDirectFirst: KNOX_DIRECT(U, false, kDirectFirst); break;                        \
        case kDirectLast: KNOX_DIRECT(U, false, kDirectLast); break;                          \
        case kDirectSum: KNOX_DIRECT_S(U, kDirectSum); break;                                 \
        case kDirectMean: KNOX_DIRECT_S(U, kDirectMean); break;                               \
        case kDirectMin: if (is_float) KNOX_DIRECT(U, false, kDirectFMin); else KNOX_DIRECT_S(U, kDirectMin); break;   \
        case kDirectMax: if (is_float) KNOX_DIRECT(U, false, kDirectFMax); else KNOX_DIRECT_S(U, kDirectMax); break;   \
        default: return hipErrorInvalidValue;                                                 \
        }                                                                                     \
    } while (0)
    switch (width) {
    case 8: KNOX_DIRECT_W(uint64_t); break;
    case 4: KNOX_DIRECT_W(uint32_t); break;
    case 2: KNOX_DIRECT_W(uint16_t); break;
    case 1: KNOX_DIRECT_W(uint8_t); break;
    default: return hipErrorInvalidValue;
    }
#undef KNOX_DIRECT_W
#undef KNOX_DIRECT_S
#undef KNOX_DIRECT
    return hipGetLastError();
}

// ---- float sums and means without the sort: exact --------------------------------------------------------------------------------
// A double-double running sum is not an atomic, but an INTEGER is: every finite value is an integer multiple of 2^(B - 1075), B = the
// lowest biased exponent the table represents, and a window's sum is kept as a 160-bit fixed-point number in five 64-bit adders of
// 32-bit digits (a value's 53-bit mantissa, shifted to its place, is cut into at most three 32-bit pieces: three LDS adds; negative
// values add the pieces' two's complements; 2^28 pieces of 32 bits stay far inside 64).  B and the number of digits come from the
// column's smallest and largest finite exponent (one pass over the values, kept with the column): as many digits as the span needs --
// three for values within 2^43 of each other --, at most five (107 binades; a wider column takes the sort).  The fold is exact, so the order of the rows is irrelevant and the result
// is the correctly rounded exact sum -- what the sorted path's double-double sums deliver unless cancellation eats more than ~50 bits.
// Non-finite values and negative zeros never enter the adders: a word of flags (+inf, -inf, NaN seen) and a count of negative zeros
// per window reproduce IEEE's answers (inf - inf and NaN -> NaN; a sum of nothing but -0 is -0).
constexpr int kFsumBins = 5;
constexpr int kFsumRange = 107;

// the value as float64 bits: NUM 0 = a float column (exact: a float32 denormal is a normal double), 1 / 2 = a signed / unsigned integer
// column -- Go's float64(v), which is what the reference's VarReducer works on (exact up to 2^53, rounded to nearest above)
template <typename U, int NUM = 0> __device__ __forceinline__ uint64_t as_f64_bits(U x) {
    if constexpr (NUM == 1) return (uint64_t)__double_as_longlong((double)(int64_t)(typename std::make_signed<U>::type)x);
    else if constexpr (NUM == 2) return (uint64_t)__double_as_longlong((double)(uint64_t)x);
    else if constexpr (sizeof(U) == 8) return (uint64_t)x;
    else return (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)x));
}

// largest and smallest biased float64 exponent among the finite non-zero values of a column: out[0] = max (0: there is none),
// out[1] = ~min (an atomicMax as well; denormals count as exponent 1, whose last mantissa bit is theirs too)
template <typename U, int NUM>
__global__ void __launch_bounds__(256) float_exprange_kernel(const U *__restrict__ val, uint64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0, mn = 0x7ffu;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    constexpr int UN = 4;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * UN) {
        U x[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) x[u] = i0 + u * stride < n ? val[i0 + u * stride] : (U)0;
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const uint64_t d = as_f64_bits<U, NUM>(x[u]);
            const uint32_t be = (uint32_t)(d >> 52) & 0x7ffu;
            const uint32_t e = be == 0x7ffu ? 0u : be != 0 ? be : ((d << 12) != 0 ? 1u : 0u);
            m = e > m ? e : m;
            mn = (e != 0 && e < mn) ? e : mn;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(m, off, 64), o2 = __shfl_xor(mn, off, 64);
        m = o > m ? o : m;
        mn = o2 < mn ? o2 : mn;
    }
    if ((threadIdx.x & 63) == 0 && m != 0) {
        atomicMax(out, m);
        atomicMax(out + 1, ~mn);
    }
}

// out: two words, zeroed by the caller.  dtype: a float column, or an integer column (its values as Go's float64(v))
hipError_t launch_float_exprange(hipStream_t s, int width, int dtype, const void *val, uint64_t n_rows, uint32_t *out) {
    uint64_t g = (n_rows + 256 * 4 - 1) / (256 * 4);
    if (g > 4096) g = 4096;
    if (g == 0) g = 1;
    const int num = dtype >= 9 ? 0 : dtype <= 4 ? 1 : 2;
#define KNOX_EXPR(U, NUM) hipLaunchKernelGGL((float_exprange_kernel<U, NUM>), dim3((unsigned)g), dim3(256), 0, s, (const U *)val, n_rows, out)
    if (num == 0 && width == 8) KNOX_EXPR(uint64_t, 0);
    else if (num == 0 && width == 4) KNOX_EXPR(uint32_t, 0);
    else if (num != 0 && width == 8) { if (num == 1) KNOX_EXPR(uint64_t, 1); else KNOX_EXPR(uint64_t, 2); }
    else if (num != 0 && width == 4) { if (num == 1) KNOX_EXPR(uint32_t, 1); else KNOX_EXPR(uint32_t, 2); }
    else if (num != 0 && width == 2) { if (num == 1) KNOX_EXPR(uint16_t, 1); else KNOX_EXPR(uint16_t, 2); }
    else if (num != 0 && width == 1) { if (num == 1) KNOX_EXPR(uint8_t, 1); else KNOX_EXPR(uint8_t, 2); }
    else return hipErrorInvalidValue;
#undef KNOX_EXPR
    return hipGetLastError();
}

// the adders of one window -> the exact sum as a double-double (hi = the sum rounded to nearest, lo = what is left, rounded), with
// IEEE's answers for the values that never entered the adders: flags bit 0 / 1 / 2 = +inf / -inf / NaN seen
__device__ __forceinline__ void fixed_sum_to_dd(const uint64_t (&b)[kFsumBins], uint32_t base, uint64_t count, uint64_t neg_zeros, uint32_t flags, double &rh_out,
                                                double &rl_out) {
    // T = sum of b[i] * 2^(32 i), every b[i] a signed 64-bit number: 192 bits in two's complement (lo | hi)
    uint64_t lo = 0;
    __int128 hi = 0;
    auto add = [&](uint64_t lo_add, __int128 hi_add) {
        const uint64_t nl = lo + lo_add;
        hi += hi_add + (nl < lo ? 1 : 0);
        lo = nl;
    };
    const __int128 v0 = (__int128)(int64_t)b[0], v1 = (__int128)(int64_t)b[1] * ((__int128)1 << 32);
    add((uint64_t)v0, v0 >> 64);
    add((uint64_t)v1, v1 >> 64);
    add(0, (__int128)(int64_t)b[2]);
    add(0, (__int128)(int64_t)b[3] * ((__int128)1 << 32));
    add(0, (__int128)(int64_t)b[4] * ((__int128)1 << 64));
    const bool neg = hi < 0;
    if (neg) {
        lo = ~lo + 1;
        hi = ~hi + (lo == 0 ? 1 : 0);
    }
    const unsigned __int128 mh = (unsigned __int128)hi;
    const uint32_t piece[6] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)mh, (uint32_t)(mh >> 32), (uint32_t)(mh >> 64), (uint32_t)(mh >> 96)};
    // the six digits, from the top, into a three-term expansion (error-free two_sum steps), then a double-double
    double s = ldexp((double)piece[5], 160), c = 0.0, c2 = 0.0;
#pragma unroll
    for (int j = 4; j >= 0; j--) {
        const knox_fin::dd t1 = knox_fin::two_sum(s, ldexp((double)piece[j], 32 * j));
        const knox_fin::dd t2 = knox_fin::two_sum(c, t1.lo);
        s = t1.hi;
        c = t2.hi;
        c2 += t2.lo;
    }
    knox_fin::dd r = knox_fin::fast_two_sum(s, c + c2);
    const int sc = (int)base - 1075;
    double rh = ldexp(r.hi, sc), rl = ldexp(r.lo, sc);
    if (neg) { rh = -rh; rl = -rl; }
    if (rh == 0.0) rh = (count != 0 && neg_zeros == count) ? -0.0 : 0.0;     // nothing but negative zeros: -0
    if (flags != 0) {                                                   // IEEE: NaN, inf - inf -> NaN; else the infinity
        rl = 0.0;
        rh = ((flags & 4u) || (flags & 3u) == 3u) ? __longlong_as_double(0x7ff8000000000000ll) : (flags & 1u) ? __longlong_as_double(0x7ff0000000000000ll)
                                                                                                              : __longlong_as_double((long long)0xfff0000000000000ull);
    }
    rh_out = rh;
    rl_out = rl;
}

// one non-zero value (-1)^neg * m * 2^(lsb - 1075) into the adders of window q (m <= 64 bits; lsb >= base)
__device__ __forceinline__ void fixed_sum_push_m(uint64_t *bin, uint32_t nb, uint32_t q, uint32_t nbins, uint32_t base, uint64_t m, uint32_t lsb, bool neg) {
    const uint32_t sft = lsb - base, i0 = sft >> 5, r = sft & 31u;
    const unsigned __int128 p = (unsigned __int128)m << r;  // <= 95 bits
    const uint64_t piece[3] = {(uint64_t)p & 0xffffffffull, (uint64_t)(p >> 32) & 0xffffffffull, (uint64_t)(p >> 64)};
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (piece[j] != 0 && i0 + j < nbins)
            atomicAdd(reinterpret_cast<unsigned long long *>(&bin[(size_t)(i0 + j) * nb + q]), (unsigned long long)(neg ? 0ull - piece[j] : piece[j]));
}
// ... a finite non-zero float64 given by its bits d (lsb >= base: base is the column's smallest)
__device__ __forceinline__ void fixed_sum_push(uint64_t *bin, uint32_t nb, uint32_t q, uint32_t nbins, uint32_t base, uint64_t d) {
    const uint32_t be = (uint32_t)(d >> 52) & 0x7ffu;
    const uint64_t frac = d & 0xfffffffffffffull;
    fixed_sum_push_m(bin, nb, q, nbins, base, be != 0 ? (frac | (1ull << 52)) : frac, be != 0 ? be : 1u, (d >> 63) != 0);
}

template <typename U>
__global__ void __launch_bounds__(kDirectBlock) window_fsum_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                   uint64_t n_rows, int64_t t_from, int64_t step, uint32_t nb_all, uint32_t cap,
                                                                   uint32_t n_ranges, uint32_t base, uint32_t nbins, uint64_t *__restrict__ slots_out,
                                                                   uint32_t *__restrict__ words_out) {
    // (ranges of windows as in window_direct_kernel: workgroup b folds range b % n_ranges over all rows)
    const uint32_t rid = blockIdx.x % n_ranges, sub = blockIdx.x / n_ranges, nsub = gridDim.x / n_ranges;
    t_from += (int64_t)((uint64_t)rid * cap) * step;
    const uint32_t nb = cap;
    const uint32_t nbr = rid * cap >= nb_all ? 0u : (nb_all - rid * cap < cap ? nb_all - rid * cap : cap);
    extern __shared__ uint64_t direct_tab[];                                   // [nbins][nb] adders, then [3][nb] words: count, negative zeros, flags
    uint64_t *bin = direct_tab;
    uint32_t *word = reinterpret_cast<uint32_t *>(direct_tab + (size_t)nbins * nb);
    for (uint32_t i = threadIdx.x; i < nbins * nb; i += kDirectBlock) bin[i] = 0;
    for (uint32_t i = threadIdx.x; i < 3 * nb; i += kDirectBlock) word[i] = 0;
    __syncthreads();
    const uint64_t span = (uint64_t)nbr * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    auto push = [&](int64_t t, U x, uint64_t) {
        const uint32_t q = window_index(t, t_from, (uint64_t)step, span, inv);
        if (q == kNoKey) return;
        atomicAdd(&word[q], 1u);
        const uint64_t d = as_f64_bits(x);
        const uint32_t be = (uint32_t)(d >> 52) & 0x7ffu;
        const uint64_t frac = d & 0xfffffffffffffull;
        const bool neg = (d >> 63) != 0;
        if (be == 0x7ffu) { atomicOr(&word[2 * nb + q], frac != 0 ? 4u : neg ? 2u : 1u); return; }
        if (be == 0 && frac == 0) { if (neg) atomicAdd(&word[nb + q], 1u); return; }
        fixed_sum_push(bin, nb, q, nbins, base, d);
    };
    direct_walk<U, true>(time, val, mask, n_rows, push, sub, nsub);
    __syncthreads();
    uint64_t *so = slots_out + (size_t)blockIdx.x * nbins * nb;
    uint32_t *wo = words_out + (size_t)blockIdx.x * 3 * nb;
    for (uint32_t i = threadIdx.x; i < nbins * nb; i += kDirectBlock) so[i] = bin[i];
    for (uint32_t i = threadIdx.x; i < 3 * nb; i += kDirectBlock) wo[i] = word[i];
}

// 64 windows per workgroup, the tables dealt to its four waves; the 160-bit sum -> a double-double -> Value() (knox_finalize.h)
template <typename U>
__global__ void __launch_bounds__(256) window_fsum_finish_kernel(const uint64_t *__restrict__ slots, const uint32_t *__restrict__ words, uint32_t grid,
                                                                 uint32_t nb_all, uint32_t cap, uint32_t n_ranges, uint32_t base, uint32_t nbins, int dtype, int func,
                                                                 void *values, uint8_t *seen) {
    __shared__ uint64_t f_bin[4][kFsumBins][64];
    __shared__ uint32_t f_w[4][3][64];
    const uint32_t lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const uint32_t kw = blockIdx.x * 64 + lane;                  // the window; k: its place in its range's tables
    const bool valid = kw < nb_all;
    const uint32_t nb = cap, rid = kw / cap, k = kw - rid * cap, nsub = grid / n_ranges;
    uint64_t b[kFsumBins] = {0, 0, 0, 0, 0};
    uint32_t w[3] = {0, 0, 0};
    if (valid) {
#pragma unroll 2
        for (uint32_t sb = part; sb < nsub; sb += 4) {
            const uint32_t g = sb * n_ranges + rid;
#pragma unroll
            for (int i = 0; i < kFsumBins; i++)
                if ((uint32_t)i < nbins) b[i] += slots[((size_t)g * nbins + i) * nb + k];
            w[0] += words[((size_t)g * 3 + 0) * nb + k];
            w[1] += words[((size_t)g * 3 + 1) * nb + k];
            w[2] |= words[((size_t)g * 3 + 2) * nb + k];
        }
    }
#pragma unroll
    for (int i = 0; i < kFsumBins; i++) f_bin[part][i][lane] = b[i];
#pragma unroll
    for (int i = 0; i < 3; i++) f_w[part][i][lane] = w[i];
    __syncthreads();
    if (part != 0 || !valid) return;
#pragma unroll
    for (int p = 1; p < 4; p++) {
#pragma unroll
        for (int i = 0; i < kFsumBins; i++) b[i] += f_bin[p][i][lane];
        w[0] += f_w[p][0][lane];
        w[1] += f_w[p][1][lane];
        w[2] |= f_w[p][2][lane];
    }
    double rh, rl;
    fixed_sum_to_dd(b, base, w[0], w[1], w[2], rh, rl);
    AggStateDev st;
    st.dtype = dtype;
    st.flags = 1u;
    st.count = (int64_t)w[0];
    st.isum = 0;
    st.fsum_hi = rh;
    st.fsum_lo = rl;
    st.min_bits = st.max_bits = st.first_bits = st.last_bits = 0;
    st.min_row = st.max_row = st.last_row = -1;
    st.first_row = w[0] ? 0 : -1;
    st.m_shift = 0.0; st.m_dev_hi = 0.0; st.m_dev_lo = 0.0; st.m_sq_hi = 0.0; st.m_sq_lo = 0.0;
    uint64_t bits = 0;
    int sn = 0;
    int64_t row = -1;
    const int rc = knox_fin::agg_value_bits(st, func, &bits, &sn, &row);
    reinterpret_cast<U *>(values)[kw] = (U)bits;
    seen[kw] = (rc == 0 && sn) ? 1 : 0;
}

// exp_max / exp_min: the column's largest / smallest finite non-zero biased float64 exponent (launch_float_exprange).  grid 0: the
// exponents span more than the five digits hold (160 bits = 107 binades + a mantissa), the table does not fit LDS, or the column holds
// nothing the fixed-point form is good for (no finite non-zero value; values so small that the scaled result would be denormal)
WindowFsumGeom window_fsum_geom(uint32_t exp_max, uint32_t exp_min, uint32_t n_buckets, int n_cus, bool masked) {
    WindowFsumGeom g;
    g.grid = 0;
    g.base = g.nbins = g.cap = 0;
    g.n_ranges = 1;
    g.lds = g.scratch = 0;
    if (exp_max < 200 || exp_min == 0 || exp_min > exp_max || exp_max - exp_min > (uint32_t)kFsumRange) return g;
    g.base = exp_min;
    g.nbins = (exp_max - exp_min + 53 + 31) / 32;
    const uint32_t max_ranges = masked ? 2u : 4u;      // (as window_direct_geom)
    uint32_t r = 1, cap = n_buckets;
    while ((size_t)cap * (8u * g.nbins + 12u) > 156 * 1024) {
        if (++r > max_ranges) return g;
        cap = (((n_buckets + r - 1) / r) + 31) & ~31u;
    }
    g.n_ranges = r;
    g.cap = cap;
    g.lds = (size_t)cap * (8u * g.nbins + 12u);
    g.grid = (uint32_t)n_cus * (g.lds <= 40 * 1024 ? 2u : 1u) / r * r;
    g.scratch = (size_t)g.grid * g.lds + 512;
    return g;
}

// func: KNOX_R_SUM / KNOX_R_MEAN over a float column
hipError_t launch_window_fsum(hipStream_t s, int width, int dtype, int func, WindowFsumGeom g, const int64_t *time, const void *val, const uint8_t *mask,
                              uint64_t n_rows, int64_t t_from, int64_t step, uint32_t n_buckets, void *scratch, void *values, uint8_t *seen) {
    if (n_buckets == 0 || g.grid == 0 || (func != 0 && func != 5)) return hipErrorInvalidValue;
    uint64_t *slots = (uint64_t *)scratch;
    uint32_t *words = (uint32_t *)(slots + (size_t)g.grid * g.nbins * g.cap);
    const dim3 fgrid((n_buckets + 63) / 64), fblock(256);
#define KNOX_FSUM(U)                                                                                                                                \
    do {                                                                                                                                            \
        if (g.lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)window_fsum_kernel<U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); \
        hipLaunchKernelGGL((window_fsum_kernel<U>), dim3(g.grid), dim3(kDirectBlock), g.lds, s, time, (const U *)val, mask, n_rows, t_from, step, n_buckets, \
                           g.cap, g.n_ranges, g.base, g.nbins, slots, words);                                                                       \
        hipLaunchKernelGGL((window_fsum_finish_kernel<U>), fgrid, fblock, 0, s, (const uint64_t *)slots, (const uint32_t *)words, g.grid, n_buckets, g.cap, \
                           g.n_ranges, g.base, g.nbins, dtype, func, values, seen);                                                                 \
    } while (0)
    if (width == 8) KNOX_FSUM(uint64_t);
    else if (width == 4) KNOX_FSUM(uint32_t);
    else return hipErrorInvalidValue;
#undef KNOX_FSUM
    return hipGetLastError();
}

// ---- var / std without the sort: exact (float columns; integer columns as Go's float64(v), what the reference's VarReducer sees) ------
// VarReducer.Value (reducer.go:376-381) is m2 / (n - 1), m2 the sum of squared deviations from the mean: n * m2 = n * S2 - S1^2 with
// S1 = the sum of the values and S2 = the sum of their squares -- both exact in fixed point (a square has a 106-bit mantissa and twice
// the exponent: up to five 32-bit digits per value, adders for twice the span), so T = n * S2 - S1^2 is an exact INTEGER (no cancellation
// whatever the mean), formed by limb arithmetic in the finish and rounded once into a double-double; m2 = T / n and the division by
// n - 1 run in double-double (knox_finalize.h var_value with D = 0).  One walk, nine LDS atomics per selected row at most.
constexpr int kFvarBins2 = 10;        // digits of S2: 2 * 107 binades + 106 bits
constexpr int kFvarLimbs = 14;        // 32-bit limbs of T (S1^2 < 2^378, n * S2 < 2^(320 + 60))

template <typename U, int NUM>
__global__ void __launch_bounds__(kDirectBlock) window_fvar_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                   uint64_t n_rows, int64_t t_from, int64_t step, uint32_t nb, uint32_t base, uint32_t nbins1,
                                                                   uint32_t nbins2, uint64_t *__restrict__ slots_out, uint32_t *__restrict__ words_out) {
    extern __shared__ uint64_t direct_tab[];                                   // [nbins1] adders of S1 | [nbins2] adders of S2, then [2][nb] words: count, flags
    const uint32_t ns = nbins1 + nbins2;
    uint64_t *bin1 = direct_tab, *bin2 = direct_tab + (size_t)nbins1 * nb;
    uint32_t *word = reinterpret_cast<uint32_t *>(direct_tab + (size_t)ns * nb);
    for (uint32_t i = threadIdx.x; i < ns * nb; i += kDirectBlock) direct_tab[i] = 0;
    for (uint32_t i = threadIdx.x; i < 2 * nb; i += kDirectBlock) word[i] = 0;
    __syncthreads();
    const uint64_t span = (uint64_t)nb * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    auto push = [&](int64_t t, U x, uint64_t) {
        const uint32_t q = window_index(t, t_from, (uint64_t)step, span, inv);
        if (q == kNoKey) return;
        atomicAdd(&word[q], 1u);
        uint64_t m;
        uint32_t lsb;
        bool neg;
        {
            const uint64_t d = as_f64_bits<U, NUM>(x);
            const uint32_t be = (uint32_t)(d >> 52) & 0x7ffu;
            const uint64_t frac = d & 0xfffffffffffffull;
            if (be == 0x7ffu) { atomicOr(&word[nb + q], 1u); return; }        // any non-finite value: NaN, like Welford
            if (be == 0 && frac == 0) return;
            lsb = be != 0 ? be : 1u;
            m = be != 0 ? (frac | (1ull << 52)) : frac;
            neg = (d >> 63) != 0;
        }
        fixed_sum_push_m(bin1, nb, q, nbins1, base, m, lsb, neg);
        // the square: m^2 (<= 128 bits) * 2^(2 * (lsb - 1075)), in units of 2^(2 * (base - 1075))
        const unsigned __int128 m2 = (unsigned __int128)m * m;
        const uint32_t sft = 2u * (lsb - base), i0 = sft >> 5, r = sft & 31u;
        const uint64_t lo64 = (uint64_t)m2, hi64 = (uint64_t)(m2 >> 64);
        const uint64_t w0 = lo64 << r, w1 = (hi64 << r) | (r ? lo64 >> (64 - r) : 0ull), w2 = r ? hi64 >> (64 - r) : 0ull;
        const uint64_t piece[5] = {w0 & 0xffffffffull, w0 >> 32, w1 & 0xffffffffull, w1 >> 32, w2};
#pragma unroll
        for (int j = 0; j < 5; j++)
            if (piece[j] != 0 && i0 + j < nbins2) atomicAdd(reinterpret_cast<unsigned long long *>(&bin2[(size_t)(i0 + j) * nb + q]), (unsigned long long)piece[j]);
    };
    direct_walk<U, true>(time, val, mask, n_rows, push);
    __syncthreads();
    uint64_t *so = slots_out + (size_t)blockIdx.x * ns * nb;
    uint32_t *wo = words_out + (size_t)blockIdx.x * 2 * nb;
    for (uint32_t i = threadIdx.x; i < ns * nb; i += kDirectBlock) so[i] = direct_tab[i];
    for (uint32_t i = threadIdx.x; i < 2 * nb; i += kDirectBlock) wo[i] = word[i];
}

// 32 windows per workgroup, the tables dealt to eight groups of 32 lanes (17 words per window and table: with four waves per 64 windows
// the fold took 155 us for 1024 windows, a tenth of the walk); then one lane per window: T = n * S2 - S1^2 in 32-bit limbs
template <typename U>
__global__ void __launch_bounds__(256) window_fvar_finish_kernel(const uint64_t *__restrict__ slots, const uint32_t *__restrict__ words, uint32_t grid, uint32_t nb,
                                                                 uint32_t base, uint32_t nbins1, uint32_t nbins2, int dtype, int func, void *values, uint8_t *seen) {
    constexpr int PARTS = 8, WPB = 32;
    __shared__ uint64_t f_b[PARTS][kFsumBins + kFvarBins2 + 1][WPB];
    __shared__ uint32_t f_fl[PARTS][WPB];
    const uint32_t lane = threadIdx.x % WPB, part = threadIdx.x / WPB;
    const uint32_t k = blockIdx.x * WPB + lane;
    const bool valid = k < nb;
    const uint32_t ns = nbins1 + nbins2;
    uint64_t b1[kFsumBins] = {0, 0, 0, 0, 0}, b2[kFvarBins2] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t count = 0;
    uint32_t flags = 0;
    if (valid) {
#pragma unroll 2
        for (uint32_t g = part; g < grid; g += PARTS) {
            const uint64_t *p = slots + (size_t)g * ns * nb + k;
#pragma unroll
            for (int i = 0; i < kFsumBins; i++)
                if ((uint32_t)i < nbins1) b1[i] += p[(size_t)i * nb];
#pragma unroll
            for (int i = 0; i < kFvarBins2; i++)
                if ((uint32_t)i < nbins2) b2[i] += p[(size_t)(nbins1 + i) * nb];
            count += words[(size_t)g * 2 * nb + k];
            flags |= words[(size_t)g * 2 * nb + nb + k];
        }
    }
#pragma unroll
    for (int i = 0; i < kFsumBins; i++) f_b[part][i][lane] = b1[i];
#pragma unroll
    for (int i = 0; i < kFvarBins2; i++) f_b[part][kFsumBins + i][lane] = b2[i];
    f_b[part][kFsumBins + kFvarBins2][lane] = count;
    f_fl[part][lane] = flags;
    __syncthreads();
    if (part != 0 || !valid) return;
#pragma unroll
    for (int w = 1; w < PARTS; w++) {
#pragma unroll
        for (int i = 0; i < kFsumBins; i++) b1[i] += f_b[w][i][lane];
#pragma unroll
        for (int i = 0; i < kFvarBins2; i++) b2[i] += f_b[w][kFsumBins + i][lane];
        count += f_b[w][kFsumBins + kFvarBins2][lane];
        flags |= f_fl[w][lane];
    }
    // |S1| as limbs: the signed adders -> 192-bit two's complement -> magnitude
    uint32_t s1[7] = {0, 0, 0, 0, 0, 0, 0};
    {
        uint64_t lo = 0;
        __int128 hi = 0;
        auto add = [&](uint64_t lo_add, __int128 hi_add) {
            const uint64_t nl = lo + lo_add;
            hi += hi_add + (nl < lo ? 1 : 0);
            lo = nl;
        };
        const __int128 v0 = (__int128)(int64_t)b1[0], v1 = (__int128)(int64_t)b1[1] * ((__int128)1 << 32);
        add((uint64_t)v0, v0 >> 64);
        add((uint64_t)v1, v1 >> 64);
        add(0, (__int128)(int64_t)b1[2]);
        add(0, (__int128)(int64_t)b1[3] * ((__int128)1 << 32));
        add(0, (__int128)(int64_t)b1[4] * ((__int128)1 << 64));
        if (hi < 0) {
            lo = ~lo + 1;
            hi = ~hi + (lo == 0 ? 1 : 0);
        }
        const unsigned __int128 mh = (unsigned __int128)hi;
        s1[0] = (uint32_t)lo; s1[1] = (uint32_t)(lo >> 32);
        s1[2] = (uint32_t)mh; s1[3] = (uint32_t)(mh >> 32); s1[4] = (uint32_t)(mh >> 64); s1[5] = (uint32_t)(mh >> 96);
    }
    // n * S2 (S2 = sum of b2[i] << 32 i, every b2[i] < 2^61)
    uint32_t T[kFvarLimbs];
#pragma unroll
    for (int i = 0; i < kFvarLimbs; i++) T[i] = 0;
    {
        uint32_t s2[kFvarBins2 + 2];
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < kFvarBins2 + 2; i++) {
            const uint64_t lo = i < kFvarBins2 ? (b2[i] & 0xffffffffull) : 0ull, hi_prev = (i >= 1 && i - 1 < kFvarBins2) ? (b2[i - 1] >> 32) : 0ull;
            const uint64_t v = lo + hi_prev + carry;      // < 2^34
            s2[i] = (uint32_t)v;
            carry = v >> 32;
        }
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < kFvarLimbs; i++) {
            const uint64_t v = (i < kFvarBins2 + 2 ? (uint64_t)s2[i] * count : 0ull) + c;    // count < 2^32
            T[i] = (uint32_t)v;
            c = v >> 32;
        }
    }
    // - S1^2 (schoolbook over the six limbs, subtracted limb by limb with borrow: T >= 0 by Cauchy-Schwarz)
    {
        uint32_t sq[kFvarLimbs];
#pragma unroll
        for (int i = 0; i < kFvarLimbs; i++) sq[i] = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const uint64_t v = (uint64_t)s1[i] * s1[j] + sq[i + j] + c;
                sq[i + j] = (uint32_t)v;
                c = v >> 32;
            }
            sq[i + 6] += (uint32_t)c;       // (i + 6 <= 11; no carry out: the partial sums stay below 2^(32 (i + 7)))
        }
        int64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < kFvarLimbs; i++) {
            const int64_t v = (int64_t)T[i] - (int64_t)sq[i] + borrow;
            T[i] = (uint32_t)v;
            borrow = v >> 32;               // 0 or -1
        }
    }
    // T -> double-double (from the top limb down, error-free steps), / n, scaled by 2^(2 (base - 1075))
    double s = 0.0, c = 0.0, c2 = 0.0;
#pragma unroll
    for (int j = kFvarLimbs - 1; j >= 0; j--) {
        const knox_fin::dd t1 = knox_fin::two_sum(s, ldexp((double)T[j], 32 * j));
        const knox_fin::dd t2 = knox_fin::two_sum(c, t1.lo);
        s = t1.hi;
        c = t2.hi;
        c2 += t2.lo;
    }
    knox_fin::dd m2 = knox_fin::fast_two_sum(s, c + c2);
    if (count > 0) m2 = knox_fin::dd_div_d(m2, (double)count);
    const int sc = (int)base - 1075;
    m2.hi = ldexp(ldexp(m2.hi, sc), sc);
    m2.lo = ldexp(ldexp(m2.lo, sc), sc);
    if (flags != 0) { m2.hi = __longlong_as_double(0x7ff8000000000000ll); m2.lo = 0.0; }
    AggStateDev st;
    st.dtype = dtype;
    st.flags = 3u;                 // bit 1: the moments are present
    st.count = (int64_t)count;
    st.isum = 0;
    st.fsum_hi = 0.0; st.fsum_lo = 0.0;
    st.min_bits = st.max_bits = st.first_bits = st.last_bits = 0;
    st.min_row = st.max_row = st.last_row = -1;
    st.first_row = count ? 0 : -1;
    st.m_shift = 0.0; st.m_dev_hi = 0.0; st.m_dev_lo = 0.0;      // deviations from the exact mean: their sum is 0
    st.m_sq_hi = m2.hi; st.m_sq_lo = m2.lo;
    uint64_t bits = 0;
    int sn = 0;
    int64_t row = -1;
    const int rc = knox_fin::agg_value_bits(st, func, &bits, &sn, &row);
    reinterpret_cast<U *>(values)[k] = (U)bits;
    seen[k] = (rc == 0 && sn) ? 1 : 0;
}

WindowFvarGeom window_fvar_geom(uint32_t exp_max, uint32_t exp_min, uint32_t n_buckets, int n_cus) {
    WindowFvarGeom g;
    g.grid = 0;
    g.base = g.nbins1 = g.nbins2 = 0;
    g.lds = g.scratch = 0;
    if (exp_max < 200 || exp_min == 0 || exp_min > exp_max || exp_max - exp_min > (uint32_t)kFsumRange) return g;
    g.base = exp_min;
    g.nbins1 = (exp_max - exp_min + 53 + 31) / 32;
    g.nbins2 = (2 * (exp_max - exp_min) + 106 + 31) / 32;
    g.lds = (size_t)n_buckets * (8u * (g.nbins1 + g.nbins2) + 8u);
    if (g.lds > 156 * 1024) return g;
    g.grid = (uint32_t)n_cus * (g.lds <= 80 * 1024 ? 2u : 1u);      // (measured: two workgroups per CU help this kernel -- 1.28 -> 1.13 ms -- and none of the others)
    g.scratch = (size_t)g.grid * g.lds + 512;
    return g;
}

// func: KNOX_R_VAR / KNOX_R_STD over a float or an integer column
hipError_t launch_window_fvar(hipStream_t s, int width, int dtype, int func, WindowFvarGeom g, const int64_t *time, const void *val, const uint8_t *mask,
                              uint64_t n_rows, int64_t t_from, int64_t step, uint32_t n_buckets, void *scratch, void *values, uint8_t *seen) {
    if (n_buckets == 0 || g.grid == 0 || (func != 6 && func != 7)) return hipErrorInvalidValue;
    uint64_t *slots = (uint64_t *)scratch;
    uint32_t *words = (uint32_t *)(slots + (size_t)g.grid * (g.nbins1 + g.nbins2) * n_buckets);
    const int num = dtype >= 9 ? 0 : dtype <= 4 ? 1 : 2;
#define KNOX_FVAR(U, NUM)                                                                                                                           \
    do {                                                                                                                                            \
        if (g.lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)window_fvar_kernel<U, NUM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); \
        hipLaunchKernelGGL((window_fvar_kernel<U, NUM>), dim3(g.grid), dim3(kDirectBlock), g.lds, s, time, (const U *)val, mask, n_rows, t_from, step, n_buckets, \
                           g.base, g.nbins1, g.nbins2, slots, words);                                                                               \
        hipLaunchKernelGGL((window_fvar_finish_kernel<U>), dim3((n_buckets + 31) / 32), dim3(256), 0, s, (const uint64_t *)slots, (const uint32_t *)words, g.grid, \
                           n_buckets, g.base, g.nbins1, g.nbins2, dtype, func, values, seen);                                                       \
    } while (0)
    if (num == 0 && width == 8) KNOX_FVAR(uint64_t, 0);
    else if (num != 0 && width == 8) { if (num == 1) KNOX_FVAR(uint64_t, 1); else KNOX_FVAR(uint64_t, 2); }
    else if (num == 0 && width == 4) KNOX_FVAR(uint32_t, 0);
    else if (num != 0 && width == 4) { if (num == 1) KNOX_FVAR(uint32_t, 1); else KNOX_FVAR(uint32_t, 2); }
    else if (num != 0 && width == 2) { if (num == 1) KNOX_FVAR(uint16_t, 1); else KNOX_FVAR(uint16_t, 2); }
    else if (num != 0 && width == 1) { if (num == 1) KNOX_FVAR(uint8_t, 1); else KNOX_FVAR(uint8_t, 2); }
    else return hipErrorInvalidValue;
#undef KNOX_FVAR
    return hipGetLastError();
}

// ---- ... and the whole state of a FLOAT column: the exact adders for the sum, the order keys of acc_update (knox_dev.h) for min / max --
// a NaN is never a candidate, both zeros share one key, the lowest row among equal keys wins --, first / last rows.  float32: (key, row)
// in one 64-bit atomic, one walk; float64: a second walk for the rows of min / max, as for 8-byte integers.
template <typename U> struct FStateShape {
    static constexpr bool WIDE = sizeof(U) == 8;
    static constexpr U SIGN = (U)1 << (8 * sizeof(U) - 1);
    static constexpr U EXPMASK = WIDE ? (U)0x7ff0000000000000ull : (U)0x7f800000u;
    static constexpr int NW = 5;                                  // count, first row, last row + 1, negative zeros, non-finite flags
    __device__ static bool is_nan(U x) { return (U)(x & (U)~SIGN) > EXPMASK; }
    __device__ static uint64_t key(U x) {                         // x is not a NaN
        const U mag = x & (U)~SIGN;
        return (uint64_t)(mag == 0 ? SIGN : ((x & SIGN) ? (U)~x : (U)(x | SIGN)));
    }
};

template <typename U>
__global__ void __launch_bounds__(kDirectBlock) window_fstate_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                     uint64_t n_rows, int64_t t_from, int64_t step, uint32_t nb, uint32_t base, uint32_t nbins,
                                                                     uint64_t *__restrict__ slots_out, uint32_t *__restrict__ words_out) {
    using SH = FStateShape<U>;
    extern __shared__ uint64_t direct_tab[];                                       // [nbins] adders | min | max, then the words
    const uint32_t ns = nbins + 2;
    uint64_t *slot = direct_tab;
    uint32_t *word = reinterpret_cast<uint32_t *>(direct_tab + (size_t)ns * nb);
    for (uint32_t i = threadIdx.x; i < ns * nb; i += kDirectBlock) slot[i] = (i / nb) == nbins ? ~0ull : 0ull;
    for (uint32_t i = threadIdx.x; i < SH::NW * nb; i += kDirectBlock) word[i] = (i / nb) == 1 ? ~0u : 0u;
    __syncthreads();
    const uint64_t span = (uint64_t)nb * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    auto push = [&](int64_t t, U x, uint64_t row) {
        const uint32_t q = window_index(t, t_from, (uint64_t)step, span, inv);
        if (q == kNoKey) return;
        const uint32_t r32 = (uint32_t)row;
        atomicAdd(&word[q], 1u);
        atomicMin(&word[nb + q], r32);
        atomicMax(&word[2 * nb + q], r32 + 1u);
        if (!SH::is_nan(x)) {
            const uint64_t k = SH::key(x);
            unsigned long long *pmin = reinterpret_cast<unsigned long long *>(&slot[(size_t)nbins * nb + q]);
            unsigned long long *pmax = reinterpret_cast<unsigned long long *>(&slot[(size_t)(nbins + 1) * nb + q]);
            if constexpr (SH::WIDE) {
                atomicMin(pmin, (unsigned long long)k);
                atomicMax(pmax, (unsigned long long)k);
            } else {
                atomicMin(pmin, (unsigned long long)((k << 32) | r32));
                atomicMax(pmax, (unsigned long long)((k << 32) | (uint32_t)~r32));
            }
        }
        const uint64_t d = as_f64_bits(x);
        const uint32_t be = (uint32_t)(d >> 52) & 0x7ffu;
        const uint64_t frac = d & 0xfffffffffffffull;
        const bool neg = (d >> 63) != 0;
        if (be == 0x7ffu) { atomicOr(&word[4 * nb + q], frac != 0 ? 4u : neg ? 2u : 1u); return; }
        if (be == 0 && frac == 0) { if (neg) atomicAdd(&word[3 * nb + q], 1u); return; }
        fixed_sum_push(slot, nb, q, nbins, base, d);
    };
    direct_walk<U, true>(time, val, mask, n_rows, push);
    __syncthreads();
    uint64_t *so = slots_out + (size_t)blockIdx.x * ns * nb;
    uint32_t *wo = words_out + (size_t)blockIdx.x * SH::NW * nb;
    for (uint32_t i = threadIdx.x; i < ns * nb; i += kDirectBlock) so[i] = slot[i];
    for (uint32_t i = threadIdx.x; i < SH::NW * nb; i += kDirectBlock) wo[i] = word[i];
}

// the workgroups' tables folded -> merged[kFsumBins + 2 + 5][nb] (64-bit words; unused adders 0)
template <typename U>
__global__ void __launch_bounds__(256) window_fstate_merge_kernel(const uint64_t *__restrict__ slots, const uint32_t *__restrict__ words, uint32_t grid, uint32_t nb,
                                                                  uint32_t nbins, uint64_t *__restrict__ merged) {
    constexpr int NF = kFsumBins + 2 + 5;
    __shared__ uint64_t part_f[4][NF][64];
    const uint32_t lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const uint32_t k = blockIdx.x * 64 + lane;
    const bool valid = k < nb;
    const uint32_t ns = nbins + 2;
    uint64_t f[NF];
#pragma unroll
    for (int i = 0; i < NF; i++) f[i] = 0;
    f[kFsumBins] = ~0ull;              // min key (pack)
    f[kFsumBins + 3] = 0xffffffffull;  // first row
    auto fold = [&](int i, uint64_t v) {
        if (i == kFsumBins || i == kFsumBins + 3) f[i] = v < f[i] ? v : f[i];
        else if (i == kFsumBins + 1 || i == kFsumBins + 4) f[i] = v > f[i] ? v : f[i];
        else if (i == kFsumBins + 6) f[i] |= v;
        else f[i] += v;
    };
    if (valid) {
#pragma unroll 2
        for (uint32_t g = part; g < grid; g += 4) {
            const uint64_t *p = slots + (size_t)g * ns * nb + k;
            const uint32_t *q = words + (size_t)g * 5 * nb + k;
#pragma unroll
            for (int i = 0; i < kFsumBins; i++)
                if ((uint32_t)i < nbins) fold(i, p[(size_t)i * nb]);
            fold(kFsumBins, p[(size_t)nbins * nb]);
            fold(kFsumBins + 1, p[(size_t)(nbins + 1) * nb]);
#pragma unroll
            for (int i = 0; i < 5; i++) fold(kFsumBins + 2 + i, q[(size_t)i * nb]);
        }
    }
#pragma unroll
    for (int i = 0; i < NF; i++) part_f[part][i][lane] = f[i];
    __syncthreads();
    if (part != 0 || !valid) return;
#pragma unroll
    for (int w = 1; w < 4; w++) {
#pragma unroll
        for (int i = 0; i < NF; i++) fold(i, part_f[w][i][lane]);
    }
#pragma unroll
    for (int i = 0; i < NF; i++) merged[(size_t)i * nb + k] = f[i];
}

// float64, second walk: the lowest row among those that hold the window's min / max key
template <typename U>
__global__ void __launch_bounds__(kDirectBlock) window_fstate_rows_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                          uint64_t n_rows, int64_t t_from, int64_t step, uint32_t nb,
                                                                          const uint64_t *__restrict__ merged, uint32_t *__restrict__ rows_out) {
    using SH = FStateShape<U>;
    extern __shared__ uint64_t direct_tab[];
    uint64_t *gmin = direct_tab, *gmax = direct_tab + nb;
    uint32_t *rows = reinterpret_cast<uint32_t *>(direct_tab + 2 * (size_t)nb);      // min row | max row
    for (uint32_t i = threadIdx.x; i < nb; i += kDirectBlock) {
        gmin[i] = merged[(size_t)kFsumBins * nb + i];
        gmax[i] = merged[(size_t)(kFsumBins + 1) * nb + i];
        rows[i] = ~0u;
        rows[nb + i] = ~0u;
    }
    __syncthreads();
    const uint64_t span = (uint64_t)nb * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    auto push = [&](int64_t t, U x, uint64_t row) {
        const uint32_t q = window_index(t, t_from, (uint64_t)step, span, inv);
        if (q == kNoKey || SH::is_nan(x)) return;
        const uint64_t key = SH::key(x);
        if (key == gmin[q]) atomicMin(&rows[q], (uint32_t)row);
        if (key == gmax[q]) atomicMin(&rows[nb + q], (uint32_t)row);
    };
    direct_walk<U, true>(time, val, mask, n_rows, push);
    __syncthreads();
    uint32_t *ro = rows_out + (size_t)blockIdx.x * 2 * nb;
    for (uint32_t i = threadIdx.x; i < 2 * nb; i += kDirectBlock) ro[i] = rows[i];
}

template <typename U>
__global__ void __launch_bounds__(256) window_fstate_finish_kernel(const uint64_t *__restrict__ merged, const uint32_t *__restrict__ rows2, uint32_t grid, uint32_t nb,
                                                                   uint32_t base, const U *__restrict__ val, uint64_t row_base, int dtype,
                                                                   AggStateDev *__restrict__ state_out) {
    using SH = FStateShape<U>;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    uint64_t b[kFsumBins];
#pragma unroll
    for (int i = 0; i < kFsumBins; i++) b[i] = merged[(size_t)i * nb + k];
    const uint64_t minv = merged[(size_t)kFsumBins * nb + k], maxv = merged[(size_t)(kFsumBins + 1) * nb + k];
    const uint64_t count = merged[(size_t)(kFsumBins + 2) * nb + k], firstrow = merged[(size_t)(kFsumBins + 3) * nb + k];
    const uint64_t lastrow1 = merged[(size_t)(kFsumBins + 4) * nb + k], negz = merged[(size_t)(kFsumBins + 5) * nb + k];
    const uint32_t flags = (uint32_t)merged[(size_t)(kFsumBins + 6) * nb + k];
    const bool any = count != 0, cand = minv != ~0ull;        // cand: a value that is not a NaN was seen
    uint64_t minrow = 0, maxrow = 0;
    if constexpr (SH::WIDE) {
        uint32_t r0 = ~0u, r1 = ~0u;
#pragma unroll 8
        for (uint32_t g = 0; g < grid; g++) {
            const uint32_t a0 = rows2[(size_t)g * 2 * nb + k], a1 = rows2[(size_t)g * 2 * nb + nb + k];
            r0 = a0 < r0 ? a0 : r0;
            r1 = a1 < r1 ? a1 : r1;
        }
        minrow = r0;
        maxrow = r1;
    } else {
        minrow = (uint32_t)minv;
        maxrow = (uint32_t)~(uint32_t)maxv;
    }
    AggStateDev s;
    s.dtype = dtype;
    s.flags = 1u;
    s.count = (int64_t)count;
    s.isum = 0;
    fixed_sum_to_dd(b, base, count, negz, flags, s.fsum_hi, s.fsum_lo);
    s.min_row = (any && cand) ? (int64_t)(row_base + minrow) : -1;
    s.max_row = (any && cand) ? (int64_t)(row_base + maxrow) : -1;
    s.first_row = any ? (int64_t)(row_base + firstrow) : -1;
    s.last_row = any ? (int64_t)(row_base + lastrow1 - 1) : -1;
    s.min_bits = (any && cand) ? (uint64_t)val[minrow] : 0;
    s.max_bits = (any && cand) ? (uint64_t)val[maxrow] : 0;
    s.first_bits = any ? (uint64_t)val[firstrow] : 0;
    s.last_bits = any ? (uint64_t)val[lastrow1 - 1] : 0;
    s.m_shift = 0.0; s.m_dev_hi = 0.0; s.m_dev_lo = 0.0; s.m_sq_hi = 0.0; s.m_sq_lo = 0.0;
    state_out[k] = s;
}

WindowFsumGeom window_fstate_geom(uint32_t exp_max, uint32_t exp_min, uint32_t n_buckets, int n_cus) {
    WindowFsumGeom g = window_fsum_geom(exp_max, exp_min, 1, n_cus, false);      // base / nbins, or grid 0
    if (g.grid == 0) return g;
    g.grid = 0;
    g.lds = (size_t)n_buckets * (8u * (g.nbins + 2) + 20u);
    if (g.lds > 156 * 1024) return g;
    g.grid = (uint32_t)n_cus * (g.lds <= 40 * 1024 ? 2u : 1u);
    g.scratch = (size_t)g.grid * g.lds + (size_t)(kFsumBins + 7) * n_buckets * 8 + (size_t)g.grid * 2 * n_buckets * 4 + 1024;
    return g;
}

hipError_t launch_window_fstate(hipStream_t s, int width, int dtype, WindowFsumGeom g, const int64_t *time, const void *val, const uint8_t *mask, uint64_t n_rows,
                                int64_t t_from, int64_t step, uint32_t n_buckets, uint64_t row_base, void *scratch, AggStateDev *states_out) {
    if (n_buckets == 0 || g.grid == 0) return hipErrorInvalidValue;
    uint64_t *slots = (uint64_t *)scratch;
    uint32_t *words = (uint32_t *)(slots + (size_t)g.grid * (g.nbins + 2) * n_buckets);
    uint64_t *merged = (uint64_t *)((char *)scratch + (((size_t)g.grid * g.lds + 255) & ~(size_t)255));
    uint32_t *rows2 = (uint32_t *)(merged + (size_t)(kFsumBins + 7) * n_buckets);
    const dim3 fgrid((n_buckets + 255) / 256), mgrid((n_buckets + 63) / 64), fblock(256);
    const size_t lds2 = (size_t)n_buckets * 24;
    hipError_t ze = launch_zero_bytes(s, merged, (size_t)(kFsumBins + 7) * n_buckets * 8);      // (adders the column does not need stay 0)
    if (ze != hipSuccess) return ze;
#define KNOX_FSTATE(U)                                                                                                                              \
    do {                                                                                                                                            \
        if (g.lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)window_fstate_kernel<U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); \
        if (lds2 > 48 * 1024) (void)hipFuncSetAttribute((const void *)window_fstate_rows_kernel<U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        hipLaunchKernelGGL((window_fstate_kernel<U>), dim3(g.grid), dim3(kDirectBlock), g.lds, s, time, (const U *)val, mask, n_rows, t_from, step, n_buckets, \
                           g.base, g.nbins, slots, words);                                                                                          \
        hipLaunchKernelGGL((window_fstate_merge_kernel<U>), mgrid, fblock, 0, s, (const uint64_t *)slots, (const uint32_t *)words, g.grid, n_buckets, g.nbins, \
                           merged);                                                                                                                 \
        if (sizeof(U) == 8)                                                                                                                         \
            hipLaunchKernelGGL((window_fstate_rows_kernel<U>), dim3(g.grid), dim3(kDirectBlock), lds2, s, time, (const U *)val, mask, n_rows, t_from, step, \
                               n_buckets, (const uint64_t *)merged, rows2);                                                                         \
        hipLaunchKernelGGL((window_fstate_finish_kernel<U>), fgrid, fblock, 0, s, (const uint64_t *)merged, (const uint32_t *)rows2, g.grid, n_buckets, g.base, \
                           (const U *)val, row_base, dtype, states_out);                                                                            \
    } while (0)
    if (width == 8) KNOX_FSTATE(uint64_t);
    else if (width == 4) KNOX_FSTATE(uint32_t);
    else return hipErrorInvalidValue;
#undef KNOX_FSTATE
    return hipGetLastError();
}

// ---- no sort at all, the whole state: every field of knox_agg_state as an order-free fold ---------------------------------------------
// count, sums, first / last as above; "first occurrence wins" for min / max (reducer.go:256-296) asks for the LOWEST row holding the
// window's smallest / largest value.  Below 8 bytes (order key, row id) fit ONE 64-bit atomic -- key << 32 | row under min,
// key << 32 | ~row under max -- and one walk does it all (four 32-bit and three 64-bit atomics per selected row).  8-byte values: the
// first walk folds the keys alone, one lane per window merges the workgroups' tables, and a second walk takes the lowest row among
// the rows whose key EQUALS the window's merged min / max (atomics only for those).  Integer columns, no var / std pass, as many
// windows as LDS holds a table for (36-44 B per window: ~3600-4400).
template <typename U, bool SIGNED> struct StateShape {
    static constexpr bool WIDE = sizeof(U) == 8;
    static constexpr int NS = WIDE ? 4 : 3;                       // WIDE: A, B, minkey, maxkey; else: Z (zero-extended sum), minpack, maxpack
    static constexpr int NW = WIDE ? 3 : (SIGNED ? 4 : 3);        // count, first row, last row + 1 [, values with the sign bit set]
    static constexpr uint64_t SIGN = 1ull << (8 * sizeof(U) - 1);
    __device__ static uint64_t key(U x) { return SIGNED ? ((uint64_t)x ^ SIGN) : (uint64_t)x; }
};

template <typename U, bool SIGNED>
__global__ void __launch_bounds__(kDirectBlock) window_state_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                    uint64_t n_rows, int64_t t_from, int64_t step, uint32_t nb, uint64_t *__restrict__ slots_out,
                                                                    uint32_t *__restrict__ words_out) {
    using SH = StateShape<U, SIGNED>;
    constexpr int NS = SH::NS, NW = SH::NW;
    extern __shared__ uint64_t direct_tab[];
    uint64_t *slot = direct_tab;                                             // [NS][nb]
    uint32_t *word = reinterpret_cast<uint32_t *>(direct_tab + (size_t)NS * nb);   // [NW][nb]
    for (uint32_t i = threadIdx.x; i < NS * nb; i += kDirectBlock) {
        const uint32_t f = i / nb;
        slot[i] = (SH::WIDE ? f == 2 : f == 1) ? ~0ull : 0ull;              // the minimum starts at the top
    }
    for (uint32_t i = threadIdx.x; i < NW * nb; i += kDirectBlock) word[i] = (i / nb) == 1 ? ~0u : 0u;   // so does the first row
    __syncthreads();
    const uint64_t span = (uint64_t)nb * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    auto A64 = [&](int f, uint32_t q) { return reinterpret_cast<unsigned long long *>(&slot[(size_t)f * nb + q]); };
    auto push = [&](int64_t t, U x, uint64_t row) {
        const uint32_t q = window_index(t, t_from, (uint64_t)step, span, inv);
        if (q == kNoKey) return;
        const uint32_t r32 = (uint32_t)row;
        const uint64_t key = SH::key(x);
        atomicAdd(&word[q], 1u);
        atomicMin(&word[nb + q], r32);
        atomicMax(&word[2 * nb + q], r32 + 1u);
        if constexpr (SH::WIDE) {
            const uint64_t sx = (uint64_t)x;
            atomicAdd(A64(0, q), (unsigned long long)(sx & 0xffffffffull));
            atomicAdd(A64(1, q), (unsigned long long)(SIGNED ? (uint64_t)((int64_t)sx >> 32) : (sx >> 32)));
            atomicMin(A64(2, q), (unsigned long long)key);
            atomicMax(A64(3, q), (unsigned long long)key);
        } else {
            if constexpr (SIGNED) { if ((uint64_t)x & SH::SIGN) atomicAdd(&word[3 * nb + q], 1u); }
            atomicAdd(A64(0, q), (unsigned long long)(uint64_t)x);
            atomicMin(A64(1, q), (unsigned long long)((key << 32) | r32));
            atomicMax(A64(2, q), (unsigned long long)((key << 32) | (uint32_t)~r32));
        }
    };
    direct_walk<U, true>(time, val, mask, n_rows, push);
    __syncthreads();
    uint64_t *so = slots_out + (size_t)blockIdx.x * NS * nb;
    uint32_t *wo = words_out + (size_t)blockIdx.x * NW * nb;
    for (uint32_t i = threadIdx.x; i < NS * nb; i += kDirectBlock) so[i] = slot[i];
    for (uint32_t i = threadIdx.x; i < NW * nb; i += kDirectBlock) wo[i] = word[i];
}

// one lane per window: the workgroups' tables folded -> merged[NS + NW][nb] (64-bit words)
template <typename U, bool SIGNED>
__global__ void __launch_bounds__(256) window_state_merge_kernel(const uint64_t *__restrict__ slots, const uint32_t *__restrict__ words, uint32_t grid, uint32_t nb,
                                                                 uint64_t *__restrict__ merged) {
    using SH = StateShape<U, SIGNED>;
    constexpr int NS = SH::NS, NW = SH::NW, NF = NS + NW;
    // 64 windows per workgroup, the tables dealt to its four waves (see window_direct_finish_kernel)
    __shared__ uint64_t part_f[4][NF][64];
    const uint32_t lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const uint32_t k = blockIdx.x * 64 + lane;
    const bool valid = k < nb;
    uint64_t f[NF];
    auto is_min = [](int i) { return i < NS ? (SH::WIDE ? i == 2 : i == 1) : (i - NS) == 1; };
    auto is_max = [](int i) { return i < NS ? (SH::WIDE ? i == 3 : i == 2) : (i - NS) == 2; };
#pragma unroll
    for (int i = 0; i < NF; i++) f[i] = is_min(i) ? (i < NS ? ~0ull : 0xffffffffull) : 0ull;
    if (valid) {
#pragma unroll 2
        for (uint32_t g = part; g < grid; g += 4) {
            const uint64_t *p = slots + (size_t)g * NS * nb + k;
            const uint32_t *q = words + (size_t)g * NW * nb + k;
            uint64_t v[NF];
#pragma unroll
            for (int i = 0; i < NS; i++) v[i] = p[(size_t)i * nb];
#pragma unroll
            for (int i = 0; i < NW; i++) v[NS + i] = q[(size_t)i * nb];
#pragma unroll
            for (int i = 0; i < NF; i++) f[i] = is_min(i) ? (v[i] < f[i] ? v[i] : f[i]) : is_max(i) ? (v[i] > f[i] ? v[i] : f[i]) : f[i] + v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NF; i++) part_f[part][i][lane] = f[i];
    __syncthreads();
    if (part != 0 || !valid) return;
#pragma unroll
    for (int w = 1; w < 4; w++) {
#pragma unroll
        for (int i = 0; i < NF; i++) {
            const uint64_t v = part_f[w][i][lane];
            f[i] = is_min(i) ? (v < f[i] ? v : f[i]) : is_max(i) ? (v > f[i] ? v : f[i]) : f[i] + v;
        }
    }
#pragma unroll
    for (int i = 0; i < NF; i++) merged[(size_t)i * nb + k] = f[i];
}

// 8-byte values, second walk: the lowest row among those that hold the window's min / max key
template <typename U, bool SIGNED>
__global__ void __launch_bounds__(kDirectBlock) window_state_rows_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                         uint64_t n_rows, int64_t t_from, int64_t step, uint32_t nb,
                                                                         const uint64_t *__restrict__ merged, uint32_t *__restrict__ rows_out) {
    using SH = StateShape<U, SIGNED>;
    extern __shared__ uint64_t direct_tab[];
    uint64_t *gmin = direct_tab, *gmax = direct_tab + nb;
    uint32_t *rows = reinterpret_cast<uint32_t *>(direct_tab + 2 * (size_t)nb);      // min row | max row
    for (uint32_t i = threadIdx.x; i < nb; i += kDirectBlock) {
        gmin[i] = merged[2 * (size_t)nb + i];
        gmax[i] = merged[3 * (size_t)nb + i];
        rows[i] = ~0u;
        rows[nb + i] = ~0u;
    }
    __syncthreads();
    const uint64_t span = (uint64_t)nb * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    auto push = [&](int64_t t, U x, uint64_t row) {
        const uint32_t q = window_index(t, t_from, (uint64_t)step, span, inv);
        if (q == kNoKey) return;
        const uint64_t key = SH::key(x);
        if (key == gmin[q]) atomicMin(&rows[q], (uint32_t)row);
        if (key == gmax[q]) atomicMin(&rows[nb + q], (uint32_t)row);
    };
    direct_walk<U, true>(time, val, mask, n_rows, push);
    __syncthreads();
    uint32_t *ro = rows_out + (size_t)blockIdx.x * 2 * nb;
    for (uint32_t i = threadIdx.x; i < 2 * nb; i += kDirectBlock) ro[i] = rows[i];
}

// the states, as bucket_finalize_kernel writes them (rows2: the second walk's tables, 8-byte values only)
template <typename U, bool SIGNED>
__global__ void __launch_bounds__(256) window_state_finish_kernel(const uint64_t *__restrict__ merged, const uint32_t *__restrict__ rows2, uint32_t grid, uint32_t nb,
                                                                  const U *__restrict__ val, uint64_t row_base, int dtype, AggStateDev *__restrict__ state_out) {
    using SH = StateShape<U, SIGNED>;
    constexpr int NS = SH::NS;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    const uint64_t count = merged[(size_t)(NS + 0) * nb + k], firstrow = merged[(size_t)(NS + 1) * nb + k], lastrow1 = merged[(size_t)(NS + 2) * nb + k];
    uint64_t xlo, minkey, maxkey, minrow, maxrow;
    int64_t xhi;
    if constexpr (SH::WIDE) {
        const uint64_t a = merged[k], b = merged[(size_t)nb + k];
        const uint64_t blo = b << 32;
        const int64_t bhi = SIGNED ? ((int64_t)b >> 32) : (int64_t)(b >> 32);
        xlo = a + blo;
        xhi = bhi + (int64_t)(xlo < a);
        minkey = merged[2 * (size_t)nb + k];
        maxkey = merged[3 * (size_t)nb + k];
        uint32_t r0 = ~0u, r1 = ~0u;
#pragma unroll 8
        for (uint32_t g = 0; g < grid; g++) {      // (independent loads: eight tables per round trip)
            const uint32_t a0 = rows2[(size_t)g * 2 * nb + k], a1 = rows2[(size_t)g * 2 * nb + nb + k];
            r0 = a0 < r0 ? a0 : r0;
            r1 = a1 < r1 ? a1 : r1;
        }
        minrow = r0;
        maxrow = r1;
    } else {
        const uint64_t z = merged[k], minpack = merged[(size_t)nb + k], maxpack = merged[2 * (size_t)nb + k];
        xlo = z;
        xhi = 0;
        if constexpr (SIGNED) {        // exact sum = zero-extended sum - 2^(8W) * (values with the sign bit set)
            const uint64_t neg = merged[(size_t)(NS + 3) * nb + k] << (8 * sizeof(U));
            xhi = -(int64_t)(xlo < neg);
            xlo -= neg;
        }
        minkey = minpack >> 32;
        maxkey = maxpack >> 32;
        minrow = (uint32_t)minpack;
        maxrow = (uint32_t)~(uint32_t)maxpack;
    }
    AggStateDev s;
    s.dtype = dtype;
    s.flags = 1u;
    s.count = (int64_t)count;
    s.isum = SH::WIDE ? xlo : merged[k];      // the wrapping sum of the zero-extended values
    s.fsum_hi = 0.0;
    s.fsum_lo = 0.0;
    knox_fin::int128_to_dd(xlo, xhi, s.fsum_hi, s.fsum_lo);
    const bool any = count != 0;
    s.min_row = any ? (int64_t)(row_base + minrow) : -1;
    s.max_row = any ? (int64_t)(row_base + maxrow) : -1;
    s.first_row = any ? (int64_t)(row_base + firstrow) : -1;
    s.last_row = any ? (int64_t)(row_base + lastrow1 - 1) : -1;
    s.min_bits = any ? (SIGNED ? (minkey ^ SH::SIGN) : minkey) : 0;     // the key map is its own inverse
    s.max_bits = any ? (SIGNED ? (maxkey ^ SH::SIGN) : maxkey) : 0;
    s.first_bits = any ? (uint64_t)val[firstrow] : 0;
    s.last_bits = any ? (uint64_t)val[lastrow1 - 1] : 0;
    s.m_shift = 0.0; s.m_dev_hi = 0.0; s.m_dev_lo = 0.0; s.m_sq_hi = 0.0; s.m_sq_lo = 0.0;
    state_out[k] = s;
}

WindowStateGeom window_state_geom(int width, int dtype, uint32_t n_buckets, int n_cus) {
    WindowStateGeom g;
    g.grid = 0;
    g.lds = g.scratch = 0;
    if (dtype < 1 || dtype > 8 || (width != 8 && width != 4 && width != 2 && width != 1)) return g;
    const bool is_signed = dtype <= 4;
    g.n_slots = width == 8 ? 4 : 3;
    g.n_words = width == 8 ? 3 : (is_signed ? 4 : 3);
    g.lds = (size_t)n_buckets * (8u * g.n_slots + 4u * g.n_words);
    if (g.lds > 156 * 1024) return g;
    g.grid = (uint32_t)n_cus * (g.lds <= 40 * 1024 ? 2u : 1u);
    // workgroup tables | merged | second walk's tables
    g.scratch = (size_t)g.grid * g.lds + (size_t)(g.n_slots + g.n_words) * n_buckets * 8 + (size_t)g.grid * 2 * n_buckets * 4 + 1024;
    return g;
}

hipError_t launch_window_state(hipStream_t s, int width, int dtype, WindowStateGeom g, const int64_t *time, const void *val, const uint8_t *mask, uint64_t n_rows,
                               int64_t t_from, int64_t step, uint32_t n_buckets, uint64_t row_base, void *scratch, AggStateDev *states_out) {
    if (n_buckets == 0 || g.grid == 0) return hipErrorInvalidValue;
    const bool is_signed = dtype >= 1 && dtype <= 4;
    uint64_t *slots = (uint64_t *)scratch;
    uint32_t *words = (uint32_t *)(slots + (size_t)g.grid * g.n_slots * n_buckets);
    uint64_t *merged = (uint64_t *)((char *)scratch + (((size_t)g.grid * g.lds + 255) & ~(size_t)255));
    uint32_t *rows2 = (uint32_t *)(merged + (size_t)(g.n_slots + g.n_words) * n_buckets);
    const dim3 fgrid((n_buckets + 255) / 256), fblock(256);
    const size_t lds2 = (size_t)n_buckets * 24;
#define KNOX_STATE(U, S)                                                                                                                            \
    do {                                                                                                                                            \
        if (g.lds > 48 * 1024) {                                                                                                                    \
            (void)hipFuncSetAttribute((const void *)window_state_kernel<U, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);             \
            (void)hipFuncSetAttribute((const void *)window_state_rows_kernel<U, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);        \
        }                                                                                                                                           \
        hipLaunchKernelGGL((window_state_kernel<U, S>), dim3(g.grid), dim3(kDirectBlock), g.lds, s, time, (const U *)val, mask, n_rows, t_from, step, n_buckets, \
                           slots, words);                                                                                                           \
        hipLaunchKernelGGL((window_state_merge_kernel<U, S>), dim3((n_buckets + 63) / 64), fblock, 0, s, (const uint64_t *)slots, (const uint32_t *)words, g.grid, \
                           n_buckets, merged);                                                                                                      \
        if (sizeof(U) == 8)                                                                                                                         \
            hipLaunchKernelGGL((window_state_rows_kernel<U, S>), dim3(g.grid), dim3(kDirectBlock), lds2, s, time, (const U *)val, mask, n_rows, t_from, step,  \
                               n_buckets, (const uint64_t *)merged, rows2);                                                                         \
        hipLaunchKernelGGL((window_state_finish_kernel<U, S>), fgrid, fblock, 0, s, (const uint64_t *)merged, (const uint32_t *)rows2, g.grid, n_buckets, \
                           (const U *)val, row_base, dtype, states_out);                                                                            \
    } while (0)
    switch (width) {
    case 8: if (is_signed) KNOX_STATE(uint64_t, true); else KNOX_STATE(uint64_t, false); break;
    case 4: if (is_signed) KNOX_STATE(uint32_t, true); else KNOX_STATE(uint32_t, false); break;
    case 2: if (is_signed) KNOX_STATE(uint16_t, true); else KNOX_STATE(uint16_t, false); break;
    case 1: if (is_signed) KNOX_STATE(uint8_t, true); else KNOX_STATE(uint8_t, false); break;
    default: return hipErrorInvalidValue;
    }
#undef KNOX_STATE
    return hipGetLastError();
}

}  // namespace knox
