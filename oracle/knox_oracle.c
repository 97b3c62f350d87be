/*
 * knox_oracle.c -- CPU ORACLE for the KnoxDB pack-scan hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C restatement of the
 * reference's Go algorithms (blockwatch-cc/knoxdb, /root/reference) used only as the
 * checker in tests/, in __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.
 * Nothing under knoxdb_amd/ may include, link, import or call it.
 *
 * Parity pin: the compare kernels and the bitset helpers are checked against the
 * reference's own golden vectors (tests/golden/cmp_cases.json <- internal/cmp/tests/<type>.go,
 * tests/golden/bitset_cases.json <- internal/bitset/tests/{pop,run}.go) and the fixed-width
 * xxh3 against internal/hash/xxh3_test.go:14-31.  The reducers have NO tests in the
 * reference (internal/reducer holds 0 test lines): their restatement is "parity unpinned"
 * and is cross-checked only by hand-derived cases and an exact-arithmetic checker.
 * XXH3-64 of byte strings comes from the un-vendored dependency github.com/zeebo/xxh3
 * v1.1.0 (go.mod:18); it is restated from the published XXH3 specification and pinned
 * against vectors generated with the independent python `xxhash` package
 * (tests/golden/make_xxh3_bytes_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Built with -ffp-contract=off: Go on amd64 never fuses a*b+c.
 *
 * Go -> C precedence traps are parenthesised explicitly.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KO_EXPORT __attribute__((visibility("default")))

/* types.BlockType  internal/types/block.go:20-36 */
enum { KO_I64 = 1, KO_I32 = 2, KO_I16 = 3, KO_I8 = 4, KO_U64 = 5, KO_U32 = 6, KO_U16 = 7, KO_U8 = 8,
       KO_F64 = 9, KO_F32 = 10, KO_BYTES = 12 };
/* types.FilterMode internal/types/mode.go:13-27 */
enum { KO_EQ = 1, KO_NE = 2, KO_GT = 3, KO_GE = 4, KO_LT = 5, KO_LE = 6, KO_RG = 9 };
/* reducer funcs    internal/reducer/reducer.go:25-45 */
enum { KO_R_SUM = 0, KO_R_FIRST = 1, KO_R_LAST = 2, KO_R_MIN = 3, KO_R_MAX = 4, KO_R_MEAN = 5,
       KO_R_VAR = 6, KO_R_STD = 7, KO_R_COUNT = 8,
       KO_R_FIRST_JOIN = 9, KO_R_LAST_JOIN = 10, KO_R_MIN_JOIN = 11, KO_R_MAX_JOIN = 12, KO_R_MEAN_JOIN = 13,
       KO_R_VAR_JOIN = 14, KO_R_STD_JOIN = 15 };

/* ------------------------------------------------------------------------------------------
 * bitset helpers: internal/bitset/generic/utils.go:12-24
 * ---------------------------------------------------------------------------------------- */
static inline int64_t bit_field_len(int64_t n) { return ((n + 7) & ~(int64_t)7) >> 3; }
static inline uint8_t bytemask(int64_t size) { return (uint8_t)(0xff >> (7 - ((size - 1) & 7))); }
static inline int pop8(uint8_t b) { return __builtin_popcount(b); }

KO_EXPORT int64_t ko_bit_field_len(int64_t n) { return bit_field_len(n); }
KO_EXPORT int ko_bytemask(int64_t size) { return bytemask(size); }

/* ------------------------------------------------------------------------------------------
 * compare kernels: internal/cmp/number.go:13-243 (integers), internal/cmp/float.go:13-242
 * (floats).  8 compares -> one byte (bit k = row idx+k), full bytes are overwritten, the
 * tail rows are OR-ed into byte n/8 (callers pass a zeroed bitset), returns the popcount.
 * ---------------------------------------------------------------------------------------- */
#define KO_CMP1(NAME, T, EXPR)                                                                  \
    static int64_t NAME(const T *src, int64_t len, T val, uint8_t *res) {                       \
        int64_t cnt = 0, n = len / 8, idx = 0;                                                  \
        for (int64_t i = 0; i < n; i++) {                                                       \
            uint8_t b = 0;                                                                      \
            for (int k = 0; k < 8; k++) {                                                       \
                T v = src[idx + k];                                                             \
                b = (uint8_t)(b + ((uint8_t)((EXPR) ? 1 : 0) << k));                            \
            }                                                                                   \
            res[i] = b;                                                                         \
            cnt += pop8(b);                                                                     \
            idx += 8;                                                                           \
        }                                                                                       \
        if (len % 8 > 0) {                                                                      \
            for (int64_t i = 0; idx + i < len; i++) {                                           \
                T v = src[idx + i];                                                             \
                if (EXPR) { res[n] |= (uint8_t)(0x1 << i); cnt++; }                             \
            }                                                                                   \
        }                                                                                       \
        return cnt;                                                                             \
    }

/* integer Between: number.go:211-243  diff := U(b-a); hit iff U(v-a) <= diff */
#define KO_CMP_BW_INT(NAME, T, U)                                                               \
    static int64_t NAME(const T *src, int64_t len, T a, T b, uint8_t *res) {                    \
        U diff = (U)((U)b - (U)a);                                                              \
        int64_t cnt = 0, n = len / 8, idx = 0;                                                  \
        for (int64_t i = 0; i < n; i++) {                                                       \
            uint8_t x = 0;                                                                      \
            for (int k = 0; k < 8; k++) {                                                       \
                U d = (U)((U)src[idx + k] - (U)a);                                              \
                x = (uint8_t)(x + ((uint8_t)(d <= diff ? 1 : 0) << k));                         \
            }                                                                                   \
            res[i] = x;                                                                         \
            cnt += pop8(x);                                                                     \
            idx += 8;                                                                           \
        }                                                                                       \
        if (len % 8 > 0) {                                                                      \
            for (int64_t i = 0; idx + i < len; i++) {                                           \
                U d = (U)((U)src[idx + i] - (U)a);                                              \
                if (d <= diff) { res[n] |= (uint8_t)(0x1 << i); cnt++; }                        \
            }                                                                                   \
        }                                                                                       \
        return cnt;                                                                             \
    }

/* float Between: float.go:211-242  a <= v && v <= b */
#define KO_CMP_BW_FLT(NAME, T)                                                                  \
    static int64_t NAME(const T *src, int64_t len, T a, T b, uint8_t *res) {                    \
        int64_t cnt = 0, n = len / 8, idx = 0;                                                  \
        for (int64_t i = 0; i < n; i++) {                                                       \
            uint8_t x = 0;                                                                      \
            for (int k = 0; k < 8; k++) {                                                       \
                T v = src[idx + k];                                                             \
                x = (uint8_t)(x + ((uint8_t)((a <= v && v <= b) ? 1 : 0) << k));                \
            }                                                                                   \
            res[i] = x;                                                                         \
            cnt += pop8(x);                                                                     \
            idx += 8;                                                                           \
        }                                                                                       \
        if (len % 8 > 0) {                                                                      \
            for (int64_t i = 0; idx + i < len; i++) {                                           \
                T v = src[idx + i];                                                             \
                if (a <= v && v <= b) { res[n] |= (uint8_t)(0x1 << i); cnt++; }                 \
            }                                                                                   \
        }                                                                                       \
        return cnt;                                                                             \
    }

#define KO_CMP_FAMILY(P, T)                                                                     \
    KO_CMP1(P##_eq, T, v == val)                                                                \
    KO_CMP1(P##_ne, T, v != val)                                                                \
    KO_CMP1(P##_lt, T, v < val)                                                                 \
    KO_CMP1(P##_le, T, v <= val)                                                                \
    KO_CMP1(P##_gt, T, v > val)                                                                 \
    KO_CMP1(P##_ge, T, v >= val)

KO_CMP_FAMILY(cmp_i64, int64_t)  KO_CMP_BW_INT(cmp_i64_bw, int64_t, uint64_t)
KO_CMP_FAMILY(cmp_i32, int32_t)  KO_CMP_BW_INT(cmp_i32_bw, int32_t, uint32_t)
KO_CMP_FAMILY(cmp_i16, int16_t)  KO_CMP_BW_INT(cmp_i16_bw, int16_t, uint16_t)
KO_CMP_FAMILY(cmp_i8, int8_t)    KO_CMP_BW_INT(cmp_i8_bw, int8_t, uint8_t)
KO_CMP_FAMILY(cmp_u64, uint64_t) KO_CMP_BW_INT(cmp_u64_bw, uint64_t, uint64_t)
KO_CMP_FAMILY(cmp_u32, uint32_t) KO_CMP_BW_INT(cmp_u32_bw, uint32_t, uint32_t)
KO_CMP_FAMILY(cmp_u16, uint16_t) KO_CMP_BW_INT(cmp_u16_bw, uint16_t, uint16_t)
KO_CMP_FAMILY(cmp_u8, uint8_t)   KO_CMP_BW_INT(cmp_u8_bw, uint8_t, uint8_t)
KO_CMP_FAMILY(cmp_f64, double)   KO_CMP_BW_FLT(cmp_f64_bw, double)
KO_CMP_FAMILY(cmp_f32, float)    KO_CMP_BW_FLT(cmp_f32_bw, float)

#define KO_DISPATCH(P, T)                                                                       \
    {                                                                                           \
        const T *s = (const T *)src;                                                            \
        T va = a ? *(const T *)a : (T)0, vb = b ? *(const T *)b : (T)0;                         \
        switch (op) {                                                                           \
        case KO_EQ: return P##_eq(s, n, va, bits);                                              \
        case KO_NE: return P##_ne(s, n, va, bits);                                              \
        case KO_LT: return P##_lt(s, n, va, bits);                                              \
        case KO_LE: return P##_le(s, n, va, bits);                                              \
        case KO_GT: return P##_gt(s, n, va, bits);                                              \
        case KO_GE: return P##_ge(s, n, va, bits);                                              \
        case KO_RG: return P##_bw(s, n, va, vb, bits);                                          \
        default: return -1;                                                                     \
        }                                                                                       \
    }

/* Exported dispatch mirrors the exported func vars of internal/cmp/cmp.go:6-114
 * (`Int64Less(src, val, bits) int64`, `Int64Between(src, a, b, bits) int64`).
 * `bits` must hold (n+7)/8 bytes and be zeroed by the caller (reader.go:334). */
KO_EXPORT int64_t ko_cmp(int dtype, int op, const void *src, int64_t n, const void *a, const void *b,
                         uint8_t *bits) {
    if (n <= 0 || src == NULL) return 0; /* cmp/tests/int64.go:212-213: l0 / nil -> 0, nothing written */
    switch (dtype) {
    case KO_I64: KO_DISPATCH(cmp_i64, int64_t)
    case KO_I32: KO_DISPATCH(cmp_i32, int32_t)
    case KO_I16: KO_DISPATCH(cmp_i16, int16_t)
    case KO_I8:  KO_DISPATCH(cmp_i8, int8_t)
    case KO_U64: KO_DISPATCH(cmp_u64, uint64_t)
    case KO_U32: KO_DISPATCH(cmp_u32, uint32_t)
    case KO_U16: KO_DISPATCH(cmp_u16, uint16_t)
    case KO_U8:  KO_DISPATCH(cmp_u8, uint8_t)
    case KO_F64: KO_DISPATCH(cmp_f64, double)
    case KO_F32: KO_DISPATCH(cmp_f32, float)
    default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * bitset combinators: internal/bitset/generic/bitset.go:13-330
 * ---------------------------------------------------------------------------------------- */
KO_EXPORT void ko_bits_and(uint8_t *dst, const uint8_t *src, int64_t size) { /* :13-49 */
    if (size <= 0) return;
    int64_t l = (size + 7) >> 3;
    for (int64_t i = 0; i < l; i++) dst[i] &= src[i];
    dst[l - 1] &= bytemask(size);
}
KO_EXPORT void ko_bits_andnot(uint8_t *dst, const uint8_t *src, int64_t size) { /* :115-151 dst &^= src */
    if (size <= 0) return;
    int64_t l = (size + 7) >> 3;
    for (int64_t i = 0; i < l; i++) dst[i] &= (uint8_t)~src[i];
    dst[l - 1] &= bytemask(size);
}
KO_EXPORT void ko_bits_or(uint8_t *dst, const uint8_t *src, int64_t size) { /* :153-189 */
    if (size <= 0) return;
    int64_t l = (size + 7) >> 3;
    for (int64_t i = 0; i < l; i++) dst[i] |= src[i];
    dst[l - 1] &= bytemask(size);
}
KO_EXPORT void ko_bits_xor(uint8_t *dst, const uint8_t *src, int64_t size) { /* :257-293 */
    if (size <= 0) return;
    int64_t l = (size + 7) >> 3;
    for (int64_t i = 0; i < l; i++) dst[i] ^= src[i];
    dst[l - 1] &= bytemask(size);
}
KO_EXPORT void ko_bits_neg(uint8_t *src, int64_t size) { /* :295-330 */
    if (size <= 0) return;
    int64_t l = (size + 7) >> 3;
    for (int64_t i = 0; i < l; i++) src[i] = (uint8_t)~src[i];
    src[l - 1] &= bytemask(size);
}
/* AndFlag :51-113 -> flags bit0 = any, bit1 = all */
KO_EXPORT int ko_bits_and_flag(uint8_t *dst, const uint8_t *src, int64_t size) {
    int64_t l = size >> 3;
    uint8_t any = 0, all = 0xff;
    for (int64_t i = 0; i < l; i++) {
        dst[i] &= src[i];
        any |= dst[i];
        all &= dst[i];
    }
    if ((size & 7) != 0) {
        dst[l] &= src[l];
        dst[l] &= bytemask(size);
        any |= dst[l];
        all &= (uint8_t)(dst[l] | (uint8_t)~bytemask(size));
    }
    return (any != 0 ? 1 : 0) | (all == 0xff ? 2 : 0);
}
/* OrFlag :191-255 */
KO_EXPORT int ko_bits_or_flag(uint8_t *dst, const uint8_t *src, int64_t size) {
    int64_t l = size >> 3;
    uint8_t any = 0, all = 0xff;
    for (int64_t i = 0; i < l; i++) {
        dst[i] |= src[i];
        any |= dst[i];
        all &= dst[i];
    }
    if ((size & 7) != 0) {
        dst[l] |= src[l];
        dst[l] &= bytemask(size);
        any |= dst[l];
        all &= (uint8_t)(dst[l] | (uint8_t)~bytemask(size));
    }
    return (any != 0 ? 1 : 0) | (all == 0xff ? 2 : 0);
}
/* PopCount :332-353: 64-bit words over src[:(size-1)>>3], remaining bytes except the last,
 * then the last byte masked by bytemask(size).  len(src) == (size+7)>>3. */
KO_EXPORT int64_t ko_bits_popcount(const uint8_t *src, int64_t size) {
    if (size <= 0) return 0;
    int64_t len = (size + 7) >> 3, cnt = 0;
    int64_t words = ((size - 1) >> 3) / 8;
    for (int64_t w = 0; w < words; w++) {
        uint64_t v;
        memcpy(&v, src + 8 * w, 8);
        cnt += __builtin_popcountll(v);
    }
    for (int64_t i = (len - 1) & ~(int64_t)7; i < len - 1; i++) cnt += pop8(src[i]);
    cnt += pop8((uint8_t)(src[len - 1] & bytemask(size)));
    return cnt;
}
/* Indexes :355-396: masks the last byte IN PLACE, emits ascending row ids.  The reference walks
 * a 256x8 decode table of 1-based bit positions added to a running base that starts at
 * 0xffffffff; emitting (byte_index*8 + bit) for each set bit LSB-first is the same sequence. */
KO_EXPORT int64_t ko_bits_indexes(uint8_t *src, int64_t size, uint32_t *dst) {
    if (size <= 0) return 0;
    int64_t len = (size + 7) >> 3, j = 0;
    src[len - 1] &= bytemask(size);
    for (int64_t i = 0; i < len; i++) {
        uint8_t b = src[i];
        while (b) {
            int k = __builtin_ctz(b);
            dst[j++] = (uint32_t)(i * 8 + k);
            b &= (uint8_t)(b - 1);
        }
    }
    return j;
}
/* Bitset.One :520-531 / fill helper: all bytes 0xff, last masked */
KO_EXPORT void ko_bits_one(uint8_t *dst, int64_t size) {
    if (size <= 0) return;
    int64_t l = (size + 7) >> 3;
    memset(dst, 0xff, (size_t)l);
    dst[l - 1] &= bytemask(size);
}

/* ------------------------------------------------------------------------------------------
 * reducers: internal/reducer/reducer.go:138-429 -- scalar fold over the selected rows in
 * ascending row order.  `bits` == NULL selects every row.  The reducer's time.Time bookkeeping
 * ("unseen" = zero time) is modelled by `seen`; `row` reports the row whose timestamp the
 * reducer would hold (first row for count/sum/first/mean/var/std, the adopted row for
 * min/max, the last row for last).  The result of type T is written to out[0..sizeof(T));
 * returns seen (Value()'s second result).
 * ---------------------------------------------------------------------------------------- */
#define KO_SELECTED(i) (bits == NULL || ((bits[(i) >> 3] >> ((i)&7)) & 1))

#define KO_REDUCE(NAME, T)                                                                      \
    static int NAME(int func, const T *src, int64_t n, const uint8_t *bits, void *out,          \
                    int64_t *row_out) {                                                         \
        int seen = 0;                                                                           \
        int64_t row = -1, cnt = 0;                                                              \
        T v = 0;                                                                                \
        double mean = 0, m2 = 0;                                                                \
        for (int64_t i = 0; i < n; i++) {                                                       \
            if (!KO_SELECTED(i)) continue;                                                      \
            T x = src[i];                                                                       \
            switch (func) {                                                                     \
            case KO_R_COUNT: /* :138-148 counts in type T */                                    \
                if (!seen) row = i;                                                             \
                v = (T)(v + 1);                                                                 \
                break;                                                                          \
            case KO_R_SUM: /* :168-178 r.v += v */                                              \
                if (!seen) row = i;                                                             \
                v = (T)(v + x);                                                                 \
                break;                                                                          \
            case KO_R_FIRST: /* :198-208 */                                                     \
                if (!seen) { row = i; v = x; }                                                  \
                break;                                                                          \
            case KO_R_LAST: /* :228-236 */                                                      \
                row = i; v = x;                                                                 \
                break;                                                                          \
            case KO_R_MAX: /* :256-266 replace iff unseen || r.v < v */                         \
                if (!seen || v < x) { row = i; v = x; }                                         \
                break;                                                                          \
            case KO_R_MIN: /* :286-296 replace iff unseen || r.v > v */                         \
                if (!seen || v > x) { row = i; v = x; }                                         \
                break;                                                                          \
            case KO_R_MEAN: { /* :318-331 Welford */                                            \
                if (cnt == 0) row = i;                                                          \
                cnt++;                                                                          \
                double delta = (double)x - mean;                                                \
                mean += delta / (double)cnt;                                                    \
                break;                                                                          \
            }                                                                                   \
            case KO_R_VAR: case KO_R_STD: { /* :352-429 */                                      \
                if (cnt == 0) row = i;                                                          \
                cnt++;                                                                          \
                double delta = (double)x - mean;                                                \
                mean += delta / (double)cnt;                                                    \
                m2 += delta * ((double)x - mean);                                               \
                break;                                                                          \
            }                                                                                   \
            default: return -1;                                                                 \
            }                                                                                   \
            seen = 1;                                                                           \
        }                                                                                       \
        T r = v;                                                                                \
        if (func == KO_R_MEAN) r = (T)mean; /* Value() = T(r.mean) :339 */                      \
        if (func == KO_R_VAR) r = (cnt < 2) ? (T)NAN : (T)(m2 / (double)(cnt - 1));             \
        if (func == KO_R_STD) r = (cnt < 2) ? (T)NAN : (T)sqrt(m2 / (double)(cnt - 1));         \
        memcpy(out, &r, sizeof(T));                                                             \
        if (row_out) *row_out = row;                                                            \
        return seen;                                                                            \
    }

/* Go converts NaN/out-of-range float->int implementation-defined; the path never feeds
 * var/std of <2 rows into an integer column in the tests, keep the cast quiet. */
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Woverflow"
KO_REDUCE(reduce_i64, int64_t)
KO_REDUCE(reduce_i32, int32_t)
KO_REDUCE(reduce_i16, int16_t)
KO_REDUCE(reduce_i8, int8_t)
KO_REDUCE(reduce_u64, uint64_t)
KO_REDUCE(reduce_u32, uint32_t)
KO_REDUCE(reduce_u16, uint16_t)
KO_REDUCE(reduce_u8, uint8_t)
KO_REDUCE(reduce_f64, double)
KO_REDUCE(reduce_f32, float)
#pragma GCC diagnostic pop

KO_EXPORT int ko_reduce(int dtype, int func, const void *src, int64_t n, const uint8_t *bits, void *out,
                        int64_t *row_out) {
    switch (dtype) {
    case KO_I64: return reduce_i64(func, (const int64_t *)src, n, bits, out, row_out);
    case KO_I32: return reduce_i32(func, (const int32_t *)src, n, bits, out, row_out);
    case KO_I16: return reduce_i16(func, (const int16_t *)src, n, bits, out, row_out);
    case KO_I8:  return reduce_i8(func, (const int8_t *)src, n, bits, out, row_out);
    case KO_U64: return reduce_u64(func, (const uint64_t *)src, n, bits, out, row_out);
    case KO_U32: return reduce_u32(func, (const uint32_t *)src, n, bits, out, row_out);
    case KO_U16: return reduce_u16(func, (const uint16_t *)src, n, bits, out, row_out);
    case KO_U8:  return reduce_u8(func, (const uint8_t *)src, n, bits, out, row_out);
    case KO_F64: return reduce_f64(func, (const double *)src, n, bits, out, row_out);
    case KO_F32: return reduce_f32(func, (const float *)src, n, bits, out, row_out);
    default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * Exact-arithmetic checker for float64 sum / mean (no reference counterpart: this is the
 * yardstick for the "<= 1 ulp" gate, SURVEY.md section 7 "Float aggregate parity").
 * A 2240-bit two's-complement fixed-point accumulator with LSB = 2^-1074 holds the exact sum
 * of any number of finite doubles; conversion back rounds to nearest-even ONCE.
 * ---------------------------------------------------------------------------------------- */
#define KO_ACC_LIMBS 36
typedef struct { uint64_t w[KO_ACC_LIMBS]; int has_nan, has_pinf, has_ninf; } ko_acc;

static void acc_add_shifted(ko_acc *a, uint64_t m, int shift, int negative) {
    int li = shift >> 6, off = shift & 63;
    uint64_t lo = m << off, hi = off ? (m >> (64 - off)) : 0;
    if (!negative) {
        unsigned __int128 c = (unsigned __int128)a->w[li] + lo;
        a->w[li] = (uint64_t)c;
        c = (unsigned __int128)a->w[li + 1] + hi + (uint64_t)(c >> 64);
        a->w[li + 1] = (uint64_t)c;
        uint64_t carry = (uint64_t)(c >> 64);
        for (int i = li + 2; carry && i < KO_ACC_LIMBS; i++) {
            a->w[i] += carry;
            carry = (a->w[i] == 0);
        }
    } else {
        uint64_t borrow = a->w[li] < lo;
        a->w[li] -= lo;
        uint64_t t = a->w[li + 1], s = hi + borrow; /* hi < 2^63 so no overflow */
        borrow = t < s;
        a->w[li + 1] = t - s;
        for (int i = li + 2; borrow && i < KO_ACC_LIMBS; i++) {
            borrow = (a->w[i] == 0);
            a->w[i] -= 1;
        }
    }
}
static void acc_add(ko_acc *a, double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    int e = (int)((u >> 52) & 0x7ff);
    uint64_t m = u & 0xfffffffffffffULL;
    int neg = (int)(u >> 63);
    if (e == 0x7ff) {
        if (m) a->has_nan = 1; else if (neg) a->has_ninf = 1; else a->has_pinf = 1;
        return;
    }
    if (e == 0) { if (!m) return; acc_add_shifted(a, m, 0, neg); return; }
    acc_add_shifted(a, m | (1ULL << 52), e - 1, neg);
}
/* |value| of a non-negative limb array with an extra sticky flag -> nearest-even double */
static double limbs_to_double(const uint64_t *w, int sticky_in) {
    int top = KO_ACC_LIMBS - 1;
    while (top >= 0 && w[top] == 0) top--;
    if (top < 0) return 0.0;
    int msb = top * 64 + (63 - __builtin_clzll(w[top])); /* bit index of the leading one */
    if (msb <= 52) { /* fits exactly (denormal or small normal): value = w[0] * 2^-1074 */
        return ldexp((double)w[0], -1074);
    }
    int lsb = msb - 52; /* lowest kept bit */
    uint64_t mant = 0;
    for (int b = msb; b >= lsb; b--) mant = (mant << 1) | ((w[b >> 6] >> (b & 63)) & 1);
    int rb = lsb - 1;
    int round_bit = (int)((w[rb >> 6] >> (rb & 63)) & 1);
    int sticky = sticky_in;
    for (int b = rb - 1; b >= 0 && !sticky; b--) sticky |= (int)((w[b >> 6] >> (b & 63)) & 1);
    if (round_bit && (sticky || (mant & 1))) mant++;
    return ldexp((double)mant, lsb - 1074); /* mant may be 2^53: ldexp handles it, inf on overflow */
}
static double acc_result(const ko_acc *a, int64_t div_n) {
    if (a->has_nan || (a->has_pinf && a->has_ninf)) return NAN;
    if (a->has_pinf) return INFINITY;
    if (a->has_ninf) return -INFINITY;
    uint64_t w[KO_ACC_LIMBS];
    memcpy(w, a->w, sizeof w);
    int neg = (int)(w[KO_ACC_LIMBS - 1] >> 63);
    if (neg) { /* two's complement negate */
        uint64_t carry = 1;
        for (int i = 0; i < KO_ACC_LIMBS; i++) {
            w[i] = ~w[i] + carry;
            carry = (carry && w[i] == 0);
        }
    }
    int sticky = 0;
    if (div_n > 1) { /* long division by a 63-bit divisor, remainder -> sticky */
        unsigned __int128 rem = 0;
        for (int i = KO_ACC_LIMBS - 1; i >= 0; i--) {
            unsigned __int128 cur = (rem << 64) | w[i];
            w[i] = (uint64_t)(cur / (uint64_t)div_n);
            rem = cur % (uint64_t)div_n;
        }
        sticky = rem != 0;
    }
    double r = limbs_to_double(w, sticky);
    return neg ? -r : r;
}
/* exact sum (and exact_sum / count, rounded once) of the selected float64 rows */
KO_EXPORT int64_t ko_exact_sum_f64(const double *src, int64_t n, const uint8_t *bits, double *sum_out,
                                   double *mean_out) {
    ko_acc a;
    memset(&a, 0, sizeof a);
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; i++) {
        if (!KO_SELECTED(i)) continue;
        acc_add(&a, src[i]);
        cnt++;
    }
    if (sum_out) *sum_out = acc_result(&a, 1);
    if (mean_out) *mean_out = cnt ? acc_result(&a, cnt) : 0.0;
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * xxh3 for fixed-width integers: internal/hash/xxh3.go:11-58, :244-254
 * ---------------------------------------------------------------------------------------- */
#define KEY32_000 0xbe4ba423ULL
#define KEY32_004 0x396cfeb8ULL
#define KEY64_008 0x1cad21f72c81017cULL
#define KEY64_016 0xdb979083e96dd4deULL
#define PRIME64_2 14029467366897019727ULL
#define PRIME64_3 1609587929392839161ULL
static inline uint64_t rol64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t avalanche_small(uint64_t x) { /* :247-254 */
    x ^= x >> 33; x *= PRIME64_2; x ^= x >> 29; x *= PRIME64_3; x ^= x >> 32;
    return x;
}
KO_EXPORT uint64_t ko_xxh3_u64(uint64_t val) { /* :22-33 */
    uint64_t input64 = (val >> 32) + (val << 32);
    uint64_t h = input64 ^ (KEY64_008 ^ KEY64_016);
    h ^= rol64(h, 49) ^ rol64(h, 24);
    h *= 0x9fb21c651e98df25ULL;
    h ^= (h >> 35) + 8;
    h *= 0x9fb21c651e98df25ULL;
    h ^= (h >> 28);
    return h;
}
KO_EXPORT uint64_t ko_xxh3_u32(uint32_t val) { /* :35-46 */
    uint64_t input64 = (uint64_t)val + ((uint64_t)val << 32);
    uint64_t h = input64 ^ (KEY64_008 ^ KEY64_016);
    h ^= rol64(h, 49) ^ rol64(h, 24);
    h *= 0x9fb21c651e98df25ULL;
    h ^= (h >> 35) + 4;
    h *= 0x9fb21c651e98df25ULL;
    h ^= (h >> 28);
    return h;
}
KO_EXPORT uint64_t ko_xxh3_u16(uint16_t val) { /* :48-52  uint64(val)*(1<<24+1)>>8 + 2<<8 */
    uint64_t h = (((uint64_t)val * ((1ULL << 24) + 1)) >> 8) + (2ULL << 8);
    h ^= KEY32_000 ^ KEY32_004;
    return avalanche_small(h);
}
KO_EXPORT uint64_t ko_xxh3_u8(uint8_t val) { /* :54-58  uint64(val)*(1<<24+1<<16+1) + 1<<8 */
    uint64_t h = ((uint64_t)val * ((1ULL << 24) + (1ULL << 16) + 1)) + (1ULL << 8);
    h ^= KEY32_000 ^ KEY32_004;
    return avalanche_small(h);
}

/* ------------------------------------------------------------------------------------------
 * XXH3-64 of a byte string, seed 0, default secret -- hash.Hash = zeebo/xxh3.Hash
 * (internal/hash/hash.go:26; dependency github.com/zeebo/xxh3 v1.1.0, go.mod:18, NOT vendored).
 * Restated from the published XXH3 specification (xxHash v0.8 doc/xxhash_spec.md).
 * ---------------------------------------------------------------------------------------- */
static const uint8_t K_SECRET[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};
#define XP32_1 0x9E3779B1U
#define XP32_2 0x85EBCA77U
#define XP32_3 0xC2B2AE3DU
#define XP64_1 0x9E3779B185EBCA87ULL
#define XP64_2 0xC2B2AE3D27D4EB4FULL
#define XP64_3 0x165667B19E3779F9ULL
#define XP64_4 0x85EBCA77C2B2AE63ULL
#define XP64_5 0x27D4EB2F165667C5ULL
#define XPMX1 0x165667919E3779F9ULL
#define XPMX2 0x9FB21C651E98DF25ULL
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t mul128_fold64(uint64_t a, uint64_t b) {
    unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
static inline uint64_t xxh64_avalanche(uint64_t h) {
    h ^= h >> 33; h *= XP64_2; h ^= h >> 29; h *= XP64_3; h ^= h >> 32;
    return h;
}
static inline uint64_t xxh3_avalanche(uint64_t h) {
    h ^= h >> 37; h *= XPMX1; h ^= h >> 32;
    return h;
}
static inline uint64_t rrmxmx(uint64_t h, uint64_t len) {
    h ^= rol64(h, 49) ^ rol64(h, 24);
    h *= XPMX2;
    h ^= (h >> 35) + len;
    h *= XPMX2;
    return h ^ (h >> 28);
}
static inline uint64_t mix16(const uint8_t *in, const uint8_t *sec) {
    return mul128_fold64(rd64(in) ^ rd64(sec), rd64(in + 8) ^ rd64(sec + 8));
}
static void xxh3_accumulate_512(uint64_t *acc, const uint8_t *in, const uint8_t *sec) {
    for (int i = 0; i < 8; i++) {
        uint64_t dv = rd64(in + 8 * i), dk = dv ^ rd64(sec + 8 * i);
        acc[i ^ 1] += dv;
        acc[i] += (dk & 0xFFFFFFFFULL) * (dk >> 32);
    }
}
static void xxh3_scramble(uint64_t *acc, const uint8_t *sec) {
    for (int i = 0; i < 8; i++) {
        uint64_t a = acc[i];
        a ^= a >> 47;
        a ^= rd64(sec + 8 * i);
        a *= XP32_1;
        acc[i] = a;
    }
}
static uint64_t xxh3_merge(const uint64_t *acc, const uint8_t *sec, uint64_t start) {
    uint64_t r = start;
    for (int i = 0; i < 4; i++) r += mul128_fold64(acc[2 * i] ^ rd64(sec + 16 * i), acc[2 * i + 1] ^ rd64(sec + 16 * i + 8));
    return xxh3_avalanche(r);
}
KO_EXPORT uint64_t ko_xxh3_bytes(const uint8_t *in, int64_t len) {
    const uint8_t *s = K_SECRET;
    if (len == 0) return xxh64_avalanche(rd64(s + 56) ^ rd64(s + 64));
    if (len <= 3) {
        uint8_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
        uint32_t comb = ((uint32_t)c1 << 16) | ((uint32_t)c2 << 24) | (uint32_t)c3 | ((uint32_t)len << 8);
        uint64_t flip = (uint64_t)(rd32(s) ^ rd32(s + 4));
        return xxh64_avalanche((uint64_t)comb ^ flip);
    }
    if (len <= 8) {
        uint32_t i1 = rd32(in), i2 = rd32(in + len - 4);
        uint64_t flip = rd64(s + 8) ^ rd64(s + 16);
        uint64_t i64 = (uint64_t)i2 + ((uint64_t)i1 << 32);
        return rrmxmx(i64 ^ flip, (uint64_t)len);
    }
    if (len <= 16) {
        uint64_t f1 = rd64(s + 24) ^ rd64(s + 32), f2 = rd64(s + 40) ^ rd64(s + 48);
        uint64_t lo = rd64(in) ^ f1, hi = rd64(in + len - 8) ^ f2;
        uint64_t acc = (uint64_t)len + __builtin_bswap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    if (len <= 128) {
        uint64_t acc = (uint64_t)len * XP64_1;
        if (len > 32) {
            if (len > 64) {
                if (len > 96) { acc += mix16(in + 48, s + 96); acc += mix16(in + len - 64, s + 112); }
                acc += mix16(in + 32, s + 64); acc += mix16(in + len - 48, s + 80);
            }
            acc += mix16(in + 16, s + 32); acc += mix16(in + len - 32, s + 48);
        }
        acc += mix16(in, s); acc += mix16(in + len - 16, s + 16);
        return xxh3_avalanche(acc);
    }
    if (len <= 240) {
        uint64_t acc = (uint64_t)len * XP64_1;
        int rounds = (int)(len / 16);
        for (int i = 0; i < 8; i++) acc += mix16(in + 16 * i, s + 16 * i);
        acc = xxh3_avalanche(acc);
        for (int i = 8; i < rounds; i++) acc += mix16(in + 16 * i, s + 16 * (i - 8) + 3);
        acc += mix16(in + len - 16, s + 136 - 17);
        return xxh3_avalanche(acc);
    }
    uint64_t acc[8] = {XP32_3, XP64_1, XP64_2, XP64_3, XP64_4, XP32_2, XP64_5, XP32_1};
    const int64_t stripes_per_block = (192 - 64) / 8, block_len = 64 * stripes_per_block;
    int64_t nblocks = (len - 1) / block_len;
    for (int64_t b = 0; b < nblocks; b++) {
        for (int64_t t = 0; t < stripes_per_block; t++) xxh3_accumulate_512(acc, in + b * block_len + t * 64, s + t * 8);
        xxh3_scramble(acc, s + 192 - 64);
    }
    int64_t nstripes = ((len - 1) - block_len * nblocks) / 64;
    for (int64_t t = 0; t < nstripes; t++) xxh3_accumulate_512(acc, in + nblocks * block_len + t * 64, s + t * 8);
    xxh3_accumulate_512(acc, in + len - 64, s + 192 - 64 - 7);
    return xxh3_merge(acc, s + 11, (uint64_t)len * XP64_1);
}

/* hash.HashT internal/hash/hash.go:67-92: ints hash their unsigned bit pattern by width,
 * floats hash their IEEE bytes with the []byte hash (:45-53). */
KO_EXPORT uint64_t ko_hash_value(int dtype, const void *val) {
    switch (dtype) {
    case KO_I64: case KO_U64: { uint64_t v; memcpy(&v, val, 8); return ko_xxh3_u64(v); }
    case KO_I32: case KO_U32: { uint32_t v; memcpy(&v, val, 4); return ko_xxh3_u32(v); }
    case KO_I16: case KO_U16: { uint16_t v; memcpy(&v, val, 2); return ko_xxh3_u16(v); }
    case KO_I8:  case KO_U8:  { uint8_t v;  memcpy(&v, val, 1); return ko_xxh3_u8(v); }
    case KO_F64: return ko_xxh3_bytes((const uint8_t *)val, 8);
    case KO_F32: return ko_xxh3_bytes((const uint8_t *)val, 4);
    default: return 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * bloom filter: internal/filter/bloom/bloom.go:46-60 (NewFilter), :130-176 (probe/add, k=4),
 * :182-184 (Contains).  Serialized form = [k][m/8 bytes], m a power of two.
 * ---------------------------------------------------------------------------------------- */
KO_EXPORT int64_t ko_bloom_pow2(int64_t m) { /* bloom.go:242-249 pow2(): next power of two >= m, minimum 8 */
    int64_t p = 8;
    while (p < m) p <<= 1;
    return p;
}
KO_EXPORT void ko_bloom_add(uint8_t *buf, int64_t buf_len, uint64_t h) { /* addUnroll4 :162-169 */
    uint8_t *bits = buf + 1;
    uint32_t mask = (uint32_t)((buf_len - 1) * 8 - 1), h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
    for (int i = 0; i < (int)buf[0]; i++) {
        bits[(h0 & mask) >> 3] |= (uint8_t)(1 << (h0 & 7));
        h0 += h1;
    }
}
KO_EXPORT int ko_bloom_contains(const uint8_t *buf, int64_t buf_len, uint64_t h) { /* :136-150, :182 */
    const uint8_t *bits = buf + 1;
    uint32_t mask = (uint32_t)((buf_len - 1) * 8 - 1), h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
    for (int i = 0; i < (int)buf[0]; i++) {
        if ((bits[(h0 & mask) >> 3] & (1 << (h0 & 7))) == 0) return 0;
        h0 += h1;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * bytes column: pkg/stringx/pool.go:45-58 (ptr = off<<32 | len), match.go:20-251
 * mask == NULL: every byte of bits (incl. partial last byte) is overwritten, returns count.
 * mask != NULL: only bits.Set(i) for mask-selected matching rows, returns -1 (count unknown).
 * ---------------------------------------------------------------------------------------- */
/* bytes.Compare */
static int bytes_compare(const uint8_t *p, uint32_t n, const uint8_t *v, uint32_t m) {
    uint32_t c = n < m ? n : m;
    int r = c ? memcmp(p, v, c) : 0;
    if (r) return r < 0 ? -1 : 1;
    return n == m ? 0 : (n < m ? -1 : 1);
}
/* one row of MatchEqual / NotEqual / Less / LessEqual / Greater / GreaterEqual / Between
 * (pkg/stringx/match.go:20-251) */
static int bytes_row_hit(int op, const uint8_t *p, uint32_t len, const uint8_t *val, uint32_t vlen, const uint8_t *val2, uint32_t vlen2) {
    switch (op) {
    case KO_EQ: return len == vlen && (len == 0 || memcmp(p, val, len) == 0);
    case KO_NE: return !(len == vlen && (len == 0 || memcmp(p, val, len) == 0));
    case KO_LT: return bytes_compare(p, len, val, vlen) < 0;
    case KO_LE: return bytes_compare(p, len, val, vlen) <= 0;
    case KO_GT: return bytes_compare(p, len, val, vlen) > 0;
    case KO_GE: return bytes_compare(p, len, val, vlen) >= 0;
    default: return bytes_compare(p, len, val, vlen) >= 0 && bytes_compare(p, len, val2, vlen2) <= 0;
    }
}
KO_EXPORT int64_t ko_bytes_match2(int op, const uint64_t *ptr, int64_t n, const uint8_t *buf, const uint8_t *val, int64_t vlen,
                                  const uint8_t *val2, int64_t vlen2, uint8_t *bits, const uint8_t *mask) {
    if (!(op == KO_EQ || op == KO_NE || op == KO_LT || op == KO_LE || op == KO_GT || op == KO_GE || op == KO_RG)) return -2;
    if (mask) {
        for (int64_t i = 0; i < n; i++) {
            if (!((mask[i >> 3] >> (i & 7)) & 1)) continue;
            uint32_t ofs = (uint32_t)(ptr[i] >> 32), len = (uint32_t)ptr[i];
            if (bytes_row_hit(op, buf + ofs, len, val, (uint32_t)vlen, val2, (uint32_t)vlen2)) bits[i >> 3] |= (uint8_t)(1 << (i & 7));
        }
        return -1;
    }
    int64_t cnt = 0;
    uint8_t b = 0;
    for (int64_t i = 0; i < n; i++) {
        uint32_t ofs = (uint32_t)(ptr[i] >> 32), len = (uint32_t)ptr[i];
        int k = (int)(i & 7);
        uint8_t bit = (uint8_t)bytes_row_hit(op, buf + ofs, len, val, (uint32_t)vlen, val2, (uint32_t)vlen2);
        b |= (uint8_t)(bit << k);
        cnt += bit;
        if (k == 7) { bits[i >> 3] = b; b = 0; }
    }
    if ((n & 7) > 0) bits[n >> 3] = b;
    return cnt;
}
/* bytesInSetMatcher / bytesNotInSetMatcher.MatchVector (operator/filter/match_bytes.go:408-470, :480-540):
 * a row matches when its bytes equal one of the set's values (the hash map and the ordered slice are
 * two lookup structures for the same equality); matching rows are bits.Set, only mask-selected rows
 * are tested when a mask is given.  The set: set_ptr[i] = off << 32 | len into set_buf. */
KO_EXPORT int ko_bytes_inset(int negate, const uint64_t *ptr, int64_t n, const uint8_t *buf, const uint64_t *set_ptr, int64_t n_set,
                             const uint8_t *set_buf, uint8_t *bits, const uint8_t *mask) {
    for (int64_t i = 0; i < n; i++) {
        if (mask && !((mask[i >> 3] >> (i & 7)) & 1)) continue;
        uint32_t ofs = (uint32_t)(ptr[i] >> 32), len = (uint32_t)ptr[i];
        int found = 0;
        for (int64_t k = 0; k < n_set && !found; k++) {
            uint32_t so = (uint32_t)(set_ptr[k] >> 32), sl = (uint32_t)set_ptr[k];
            found = sl == len && (len == 0 || memcmp(buf + ofs, set_buf + so, len) == 0);
        }
        if (found != (negate != 0)) bits[i >> 3] |= (uint8_t)(1 << (i & 7));
    }
    return 0;
}
KO_EXPORT int64_t ko_bytes_match(int op, const uint64_t *ptr, int64_t n, const uint8_t *buf, const uint8_t *val,
                                 int64_t vlen, uint8_t *bits, const uint8_t *mask) {
    if (op != KO_EQ && op != KO_NE) return -2;
    return ko_bytes_match2(op, ptr, n, buf, val, vlen, NULL, 0, bits, mask);
}

/* ------------------------------------------------------------------------------------------
 * Time-window aggregation ("next" row f1): pkg/series/series.go:192-246 feeds every selected row
 * to NativeBucket.Push (internal/reducer/bucket_native.go:104-159) with its timestamp truncated
 * to the window start (Interval.TruncateRelative, fixed-duration windows here):
 *     t = from + floor((ts - from) / step) * step
 * Push keeps one reducer per window that received a row: the typical case appends a reducer when
 * t >= next, the out-of-order case binary-searches the reducer list by time and inserts a missing
 * window in place (:120-141).  Output = (window time, Value()) per reducer in list order.
 * ---------------------------------------------------------------------------------------- */
#define KO_BUCKET(NAME, T)                                                                      \
    typedef struct { int64_t t; int seen; T v; int64_t n; double mean, m2; T j; int done; } NAME##_red; \
    /* join reducers (reducer.go:432-731): rows with the same raw timestamp as the previous streamed row \
     * (join, pkg/series/series.go:200-203) are summed into r.j before the reducer proper sees them;    \
     * `!r->seen` stands for r.t.IsZero().  Value() of the join reducers flushes r.j first.             \
     * The code below is the reference's, statement by statement, quirks included (MaxJoin / MinJoin    \
     * start from r.v = 0).  */                                                                         \
    static void NAME##_reduce_join(int func, NAME##_red *r, T x, int64_t t, int join) {         \
        if (func == KO_R_LAST_JOIN) { r->t = t; r->seen = 1; if (join) r->v = (T)(r->v + x); else r->v = x; return; } \
        if (!r->seen) { /* r.t.IsZero() */                                                      \
            r->t = t; r->seen = 1;                                                              \
            if (func == KO_R_FIRST_JOIN) r->v = x; else r->j = x;                               \
            return;                                                                             \
        }                                                                                       \
        switch (func) {                                                                         \
        case KO_R_FIRST_JOIN: if (join && !r->done) r->v = (T)(r->v + x); else r->done = 1; break; \
        case KO_R_MAX_JOIN: if (join) r->j = (T)(r->j + x); else { if (r->v < r->j) r->v = r->j; r->j = x; } break; \
        case KO_R_MIN_JOIN: if (join) r->j = (T)(r->j + x); else { if (r->j < r->v) r->v = r->j; r->j = x; } break; \
        default: /* mean / var / std join */                                                    \
            if (join) r->j = (T)(r->j + x);                                                     \
            else {                                                                              \
                r->n++;                                                                         \
                double d = (double)r->j - r->mean;                                              \
                r->mean += d / (double)r->n;                                                    \
                if (func != KO_R_MEAN_JOIN) r->m2 += d * ((double)r->j - r->mean);              \
                r->j = x;                                                                       \
            }                                                                                   \
        }                                                                                       \
    }                                                                                           \
    /* Reducer.Value(): returns seen, writes the value */                                       \
    static int NAME##_value(int func, NAME##_red *r, T *out) {                                  \
        switch (func) {                                                                         \
        case KO_R_MEAN: *out = (T)r->mean; return r->n > 0;                                     \
        case KO_R_VAR: *out = r->n < 2 ? (T)NAN : (T)(r->m2 / (double)(r->n - 1)); return r->n > 0; \
        case KO_R_STD: *out = r->n < 2 ? (T)NAN : (T)sqrt(r->m2 / (double)(r->n - 1)); return r->n > 0; \
        case KO_R_MAX_JOIN: if (r->seen) NAME##_reduce_join(func, r, (T)0, r->t, 0); *out = r->v; return r->seen; \
        case KO_R_MIN_JOIN: if (r->seen) NAME##_reduce_join(func, r, r->j, r->t, 0); *out = r->v; return r->seen; \
        case KO_R_MEAN_JOIN: if (r->seen) NAME##_reduce_join(func, r, (T)0, r->t, 0); *out = (T)r->mean; return r->n > 0; \
        case KO_R_VAR_JOIN: case KO_R_STD_JOIN: {                                               \
            if (r->seen) NAME##_reduce_join(func, r, (T)0, r->t, 0);                            \
            if (r->n < 2) *out = (T)NAN;                                                        \
            else *out = func == KO_R_VAR_JOIN ? (T)(r->m2 / (double)(r->n - 1)) : (T)sqrt(r->m2 / (double)(r->n - 1)); \
            return r->n > 0; }                                                                  \
        default: *out = r->v; return r->seen;                                                   \
        }                                                                                       \
    }                                                                                           \
    static void NAME##_reduce(int func, NAME##_red *r, T x, int64_t t) {                        \
        switch (func) {                                                                         \
        case KO_R_COUNT: if (!r->seen) r->t = t; r->v = (T)(r->v + 1); break;                   \
        case KO_R_SUM: if (!r->seen) r->t = t; r->v = (T)(r->v + x); break;                     \
        case KO_R_FIRST: if (!r->seen) { r->t = t; r->v = x; } break;                           \
        case KO_R_LAST: r->t = t; r->v = x; break;                                              \
        case KO_R_MAX: if (!r->seen || r->v < x) { r->t = t; r->v = x; } break;                 \
        case KO_R_MIN: if (!r->seen || r->v > x) { r->t = t; r->v = x; } break;                 \
        default: { /* mean / var / std (Welford) :318-429 */                                    \
            if (r->n == 0) r->t = t;                                                            \
            r->n++;                                                                             \
            double d = (double)x - r->mean;                                                     \
            r->mean += d / (double)r->n;                                                        \
            if (func != KO_R_MEAN) r->m2 += d * ((double)x - r->mean);                          \
            break; }                                                                            \
        }                                                                                       \
        r->seen = 1;                                                                            \
    }                                                                                           \
    static int64_t NAME##_run(int func, const int64_t *times, const T *vals, int64_t n, const uint8_t *bits, \
                              int64_t from, int64_t step, int64_t *out_t, T *out_v, int64_t cap) { \
        NAME##_red *reds = (NAME##_red *)calloc((size_t)cap + 1, sizeof(NAME##_red));           \
        int64_t nr = 0, last = 0, next = 0, prev_raw = 0;                                       \
        int have_next = 0, have_prev = 0; /* b.next / series' `last` are the zero time until the first row */ \
        for (int64_t i = 0; i < n; i++) {                                                       \
            if (!KO_SELECTED(i)) continue;                                                      \
            const int join = have_prev && prev_raw == times[i]; /* series.go:200-203 */          \
            prev_raw = times[i]; have_prev = 1;                                                 \
            int64_t d = times[i] - from, q = d / step;                                          \
            if ((d % step != 0) && ((d < 0) != (step < 0))) q--; /* floor division */           \
            const int64_t t = from + q * step;                                                  \
            int64_t target = nr - 1;                                                            \
            if (!have_next || t >= next) { /* typical case: window boundary crossed :113-121 */ \
                if (nr >= cap) { free(reds); return -1; }                                       \
                memset(&reds[nr], 0, sizeof(NAME##_red)); nr++;                                 \
                last = t; next = t + step; have_next = 1; target++;                             \
            } else if (t < last) { /* out of order :123-141 */                                  \
                int64_t idx = 0;                                                                \
                while (idx < nr && !(reds[idx].t > t)) idx++; /* first reducer after t */       \
                if (idx == 0 || reds[idx - 1].t + step <= t) {                                  \
                    if (nr >= cap) { free(reds); return -1; }                                   \
                    memmove(&reds[idx + 1], &reds[idx], (size_t)(nr - idx) * sizeof(NAME##_red)); \
                    memset(&reds[idx], 0, sizeof(NAME##_red)); nr++;                            \
                    target = idx;                                                               \
                } else target = idx - 1;                                                        \
            }                                                                                   \
            if (func >= KO_R_FIRST_JOIN) NAME##_reduce_join(func, &reds[target], vals[i], t, join); \
            else NAME##_reduce(func, &reds[target], vals[i], t);                                \
        }                                                                                       \
        for (int64_t k = 0; k < nr; k++) {                                                      \
            out_t[k] = reds[k].t;                                                               \
            NAME##_value(func, &reds[k], &out_v[k]);                                            \
        }                                                                                       \
        free(reds);                                                                             \
        return nr;                                                                              \
    }
KO_BUCKET(bk_i64, int64_t) KO_BUCKET(bk_i32, int32_t) KO_BUCKET(bk_i16, int16_t) KO_BUCKET(bk_i8, int8_t)
KO_BUCKET(bk_u64, uint64_t) KO_BUCKET(bk_u32, uint32_t) KO_BUCKET(bk_u16, uint16_t) KO_BUCKET(bk_u8, uint8_t)
KO_BUCKET(bk_f64, double) KO_BUCKET(bk_f32, float)

KO_EXPORT int64_t ko_buckets(int dtype, int func, const int64_t *times, const void *vals, int64_t n, const uint8_t *bits,
                             int64_t from, int64_t step, int64_t *out_t, void *out_v, int64_t cap) {
    if (step <= 0) return -2;
    switch (dtype) {
    case KO_I64: return bk_i64_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_I32: return bk_i32_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_I16: return bk_i16_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_I8:  return bk_i8_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_U64: return bk_u64_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_U32: return bk_u32_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_U16: return bk_u16_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_U8:  return bk_u8_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_F64: return bk_f64_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    case KO_F32: return bk_f32_run(func, times, vals, n, bits, from, step, out_t, out_v, cap);
    default: return -3;
    }
}

/* ------------------------------------------------------------------------------------------
 * Bool columns are bitsets; their matchers are copies / negations: internal/bitset/match.go:14-62
 * (MatchEqual .. MatchBetween).  `bits` arrives zeroed; a non-matching mode leaves it untouched.
 * ---------------------------------------------------------------------------------------- */
static void bool_eq(int val, const uint8_t *src, int64_t size, uint8_t *bits) {
    int64_t l = (size + 7) >> 3;
    memcpy(bits, src, (size_t)l);
    if (size > 0) bits[l - 1] &= bytemask(size);
    if (!val) ko_bits_neg(bits, size);
}
KO_EXPORT int ko_bool_match(int op, int a, int b, const uint8_t *src, int64_t size, uint8_t *bits) {
    switch (op) {
    case KO_EQ: bool_eq(a, src, size, bits); return 0;
    case KO_NE: bool_eq(!a, src, size, bits); return 0;
    case KO_LT: if (a) bool_eq(0, src, size, bits); return 0;
    case KO_LE: if (a) ko_bits_one(bits, size); else bool_eq(0, src, size, bits); return 0;
    case KO_GT: if (!a) bool_eq(1, src, size, bits); return 0;
    case KO_GE: if (!a) ko_bits_one(bits, size); else bool_eq(1, src, size, bits); return 0;
    case KO_RG:
        if (a && b) bool_eq(1, src, size, bits);
        else if (!a && b) ko_bits_one(bits, size);
        else if (!a && !b) bool_eq(0, src, size, bits);
        return 0;
    default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * Int128 / Int256 compare kernels over the stride layout (one limb slice per 64-bit limb,
 * X0 = most significant, signed):  internal/cmp/int128.go:14-471, int256.go:14-270;
 * ordering num.Int128.Lt/Le/Gt/Ge pkg/num/int128.go:541-555 (signed top limb, then unsigned
 * limbs), num.Int256.Lt pkg/num/int256.go:701-713 (sign test, then unsigned compare) -- the
 * same lexicographic order.  limbs[l][i] = limb l of row i; a / b most significant limb first.
 *
 * Mask handling restated as written: the unrolled i128 eq..ge loops store `b & m` per full byte
 * but add popcount(b) -- the UNMASKED byte -- to the returned count (int128.go:40-41 and
 * siblings); their tails, cmp_i128_bw and all cmp_i256_* count only mask-selected hits.
 * `quirk_count` selects the former so the returned count is the reference's in both families.
 * ---------------------------------------------------------------------------------------- */
static int wide_cmp3(const uint64_t *const *limbs, int nl, int64_t i, const uint64_t *v) {
    /* -1 / 0 / +1 : row i <=> v */
    int64_t x0 = (int64_t)limbs[0][i], v0 = (int64_t)v[0];
    if (x0 != v0) return x0 < v0 ? -1 : 1;
    for (int l = 1; l < nl; l++) {
        if (limbs[l][i] != v[l]) return limbs[l][i] < v[l] ? -1 : 1;
    }
    return 0;
}
static int wide_hit(int op, const uint64_t *const *limbs, int nl, int64_t i, const uint64_t *a, const uint64_t *b) {
    int c = wide_cmp3(limbs, nl, i, a);
    switch (op) {
    case KO_EQ: return c == 0;
    case KO_NE: return c != 0;
    case KO_LT: return c < 0;
    case KO_LE: return c <= 0;
    case KO_GT: return c > 0;
    case KO_GE: return c >= 0;
    default: return c >= 0 && wide_cmp3(limbs, nl, i, b) <= 0; /* a.Le(v) && b.Ge(v) */
    }
}
KO_EXPORT int64_t ko_cmp_wide(int n_limbs, int op, const uint64_t *const *limbs, int64_t n, const uint64_t *a, const uint64_t *b,
                              uint8_t *bits, const uint8_t *mask) {
    if ((n_limbs != 2 && n_limbs != 4) || !(op == KO_EQ || op == KO_NE || op == KO_LT || op == KO_LE || op == KO_GT || op == KO_GE || op == KO_RG))
        return -1;
    int64_t cnt = 0;
    const int unrolled = n_limbs == 2 && op != KO_RG; /* the byte-at-a-time i128 loops */
    if (unrolled) {
        int64_t nb = n / 8;
        for (int64_t i = 0; i < nb; i++) {
            if (mask && mask[i] == 0) continue;
            uint8_t byte = 0;
            for (int k = 0; k < 8; k++) byte |= (uint8_t)(wide_hit(op, limbs, n_limbs, i * 8 + k, a, b) << k);
            bits[i] = mask ? (uint8_t)(byte & mask[i]) : byte;
            cnt += __builtin_popcount(byte); /* unmasked, as written */
        }
        for (int64_t i = nb * 8; i < n; i++) {
            if (mask && !((mask[nb] >> (i & 7)) & 1)) continue;
            if (wide_hit(op, limbs, n_limbs, i, a, b)) { bits[nb] |= (uint8_t)(1u << (i & 7)); cnt++; }
        }
        return cnt;
    }
    for (int64_t i = 0; i < n; i++) {
        if (mask && !((mask[i >> 3] >> (i & 7)) & 1)) continue;
        if (wide_hit(op, limbs, n_limbs, i, a, b)) { bits[i >> 3] |= (uint8_t)(1u << (i & 7)); cnt++; }
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * IN / NOT IN set matchers on materialized blocks:
 *   integers  numInSetMatcher / numNotInSetMatcher.MatchVector  operator/filter/match_num.go:672-690, :789-808
 *             (membership of uint64(v) in a set built from uint64(T(x)), i.e. equality on T)
 *   floats    floatInSetMatcher / floatNotInSetMatcher.MatchVector  operator/filter/match_float.go:57-73
 *             (OrderedFloats.Contains = IEEE equality: NaN is never a member, -0 == +0)
 * Only bits.Set(i): matching rows are OR-ed into `bits`; with a mask only mask-selected rows are tested.
 * ---------------------------------------------------------------------------------------- */
#define KO_INSET(NAME, T)                                                                       \
    static void NAME(int negate, const T *src, int64_t n, const T *set, int64_t nset,           \
                     uint8_t *bits, const uint8_t *mask) {                                      \
        for (int64_t i = 0; i < n; i++) {                                                       \
            if (mask && !((mask[i >> 3] >> (i & 7)) & 1)) continue;                             \
            int found = 0;                                                                      \
            for (int64_t k = 0; k < nset && !found; k++) found = (set[k] == src[i]);            \
            if (found != negate) bits[i >> 3] |= (uint8_t)(1 << (i & 7));                       \
        }                                                                                       \
    }
KO_INSET(inset_i64, int64_t) KO_INSET(inset_i32, int32_t) KO_INSET(inset_i16, int16_t) KO_INSET(inset_i8, int8_t)
KO_INSET(inset_u64, uint64_t) KO_INSET(inset_u32, uint32_t) KO_INSET(inset_u16, uint16_t) KO_INSET(inset_u8, uint8_t)
KO_INSET(inset_f64, double) KO_INSET(inset_f32, float)

KO_EXPORT int ko_inset(int dtype, int negate, const void *src, int64_t n, const void *set, int64_t nset, uint8_t *bits,
                       const uint8_t *mask) {
    switch (dtype) {
    case KO_I64: inset_i64(negate, src, n, set, nset, bits, mask); return 0;
    case KO_I32: inset_i32(negate, src, n, set, nset, bits, mask); return 0;
    case KO_I16: inset_i16(negate, src, n, set, nset, bits, mask); return 0;
    case KO_I8:  inset_i8(negate, src, n, set, nset, bits, mask); return 0;
    case KO_U64: inset_u64(negate, src, n, set, nset, bits, mask); return 0;
    case KO_U32: inset_u32(negate, src, n, set, nset, bits, mask); return 0;
    case KO_U16: inset_u16(negate, src, n, set, nset, bits, mask); return 0;
    case KO_U8:  inset_u8(negate, src, n, set, nset, bits, mask); return 0;
    case KO_F64: inset_f64(negate, src, n, set, nset, bits, mask); return 0;
    case KO_F32: inset_f32(negate, src, n, set, nset, bits, mask); return 0;
    default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * Raw on-disk containers ("next" row f2): pkg/num/varint.go:85-197 (SQLite4-style prefix varint,
 * NOT LEB128) and internal/encode/int_raw.go:73-94, float_raw.go:73-89.
 *   integers: [TIntRaw = 7 ][uvarint number of VALUES][little-endian values]
 *   floats:   [TFloatRaw=15][uvarint number of BYTES ][little-endian values]
 * ---------------------------------------------------------------------------------------- */
KO_EXPORT int ko_put_uvarint(uint8_t *b, uint64_t x) { /* varint.go:85-159 */
    if (x <= 240) { b[0] = (uint8_t)x; return 1; }
    if (x <= 2287) { uint64_t y = x - 240; b[0] = (uint8_t)((y >> 8) + 241); b[1] = (uint8_t)y; return 2; }
    if (x <= 67823) { uint64_t y = x - 2288; b[0] = 249; b[1] = (uint8_t)(y >> 8); b[2] = (uint8_t)y; return 3; }
    if (x <= 0xffffff) { b[0] = 250; b[1] = (uint8_t)(x >> 16); b[2] = (uint8_t)(x >> 8); b[3] = (uint8_t)x; return 4; }
    uint64_t w = x >> 32;
    int hi = w == 0 ? 0 : w <= 0xff ? 1 : w <= 0xffff ? 2 : w <= 0xffffff ? 3 : 4; /* bytes of w */
    b[0] = (uint8_t)(251 + hi);
    for (int i = 0; i < hi; i++) b[1 + i] = (uint8_t)(w >> (8 * (hi - 1 - i)));
    b[1 + hi] = (uint8_t)(x >> 24); b[2 + hi] = (uint8_t)(x >> 16); b[3 + hi] = (uint8_t)(x >> 8); b[4 + hi] = (uint8_t)x;
    return 5 + hi;
}
KO_EXPORT int ko_uvarint(const uint8_t *b, uint64_t *out) { /* varint.go:161-197 */
    uint8_t b0 = b[0];
    if (b0 <= 240) { *out = b0; return 1; }
    if (b0 <= 248) { *out = 240 + ((uint64_t)(b0 - 241) << 8) + b[1]; return 2; }
    if (b0 == 249) { *out = 2288 + ((uint64_t)b[1] << 8) + b[2]; return 3; }
    if (b0 == 250) { *out = ((uint64_t)b[1] << 16) + ((uint64_t)b[2] << 8) + b[3]; return 4; }
    uint64_t x = ((uint64_t)b[1] << 24) + ((uint64_t)b[2] << 16) + ((uint64_t)b[3] << 8) + b[4];
    if (b0 == 251) { *out = x; return 5; }
    if (b0 == 252) { *out = (x << 8) + b[5]; return 6; }
    if (b0 == 253) { *out = (x << 16) + ((uint64_t)b[5] << 8) + b[6]; return 7; }
    if (b0 == 254) { *out = (x << 24) + ((uint64_t)b[5] << 16) + ((uint64_t)b[6] << 8) + b[7]; return 8; }
    *out = (x << 32) + ((uint64_t)b[5] << 24) + ((uint64_t)b[6] << 16) + ((uint64_t)b[7] << 8) + b[8];
    return 9;
}
static int ko_width(int dtype) {
    switch (dtype) {
    case KO_I64: case KO_U64: case KO_F64: return 8;
    case KO_I32: case KO_U32: case KO_F32: return 4;
    case KO_I16: case KO_U16: return 2;
    case KO_I8: case KO_U8: return 1;
    default: return 0;
    }
}
/* RawContainer.Store / FloatRawContainer.Store: returns bytes written */
KO_EXPORT int64_t ko_raw_container_store(int dtype, const void *values, int64_t n, uint8_t *dst) {
    int w = ko_width(dtype), isf = (dtype == KO_F64 || dtype == KO_F32);
    if (!w) return -1;
    int64_t o = 0;
    dst[o++] = isf ? 15 : 7;
    o += ko_put_uvarint(dst + o, isf ? (uint64_t)n * (uint64_t)w : (uint64_t)n);
    memcpy(dst + o, values, (size_t)n * (size_t)w);
    return o + n * w;
}
/* RawContainer.Load / FloatRawContainer.Load: returns the number of values, copies them to out */
KO_EXPORT int64_t ko_raw_container_load(int dtype, const uint8_t *buf, int64_t len, void *out, int64_t cap) {
    int w = ko_width(dtype), isf = (dtype == KO_F64 || dtype == KO_F32);
    if (!w || len < 2 || buf[0] != (isf ? 15 : 7)) return -1; /* ErrInvalidType */
    uint64_t v;
    int k = ko_uvarint(buf + 1, &v);
    int64_t n = isf ? (int64_t)(v / (uint64_t)w) : (int64_t)v;
    if (1 + k + n * w > len || n > cap) return -2;
    memcpy(out, buf + 1 + k, (size_t)n * (size_t)w);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Bit-packed integer container (TIntBitpacked = 4): internal/encode/int_bitpack.go:93-133 +
 * internal/encode/bitpack/encode.go:216-246 (generic horizontal packer), decode.go:24-74.
 *   [4][uvarint uint64(For)][uvarint log2][uvarint N][packed bits]
 * For = min value (frame of reference), log2 = bits.Len64(max - min) (types/number.go:163-169),
 * value i = T((stream >> i*log2) & mask) + For, stream = little-endian 64-bit words,
 * packed size = ((log2*N + 63) &^ 63) / 8 bytes (bitpack.go:9-11); log2 == 0 packs nothing.
 * ---------------------------------------------------------------------------------------- */
static uint64_t ko_raw_bits(int dtype, const void *values, int64_t i) { /* sign/zero extended like uint64(T) */
    switch (dtype) {
    case KO_I64: return (uint64_t)((const int64_t *)values)[i];
    case KO_I32: return (uint64_t)(int64_t)((const int32_t *)values)[i];
    case KO_I16: return (uint64_t)(int64_t)((const int16_t *)values)[i];
    case KO_I8:  return (uint64_t)(int64_t)((const int8_t *)values)[i];
    case KO_U64: return ((const uint64_t *)values)[i];
    case KO_U32: return ((const uint32_t *)values)[i];
    case KO_U16: return ((const uint16_t *)values)[i];
    default:     return ((const uint8_t *)values)[i];
    }
}
static int ko_is_signed(int dtype) { return dtype >= KO_I64 && dtype <= KO_I8; }
KO_EXPORT int64_t ko_bitpack_container_store(int dtype, const void *values, int64_t n, uint8_t *dst) {
    int w = ko_width(dtype);
    if (!w || dtype == KO_F64 || dtype == KO_F32 || n <= 0) return -1;
    uint64_t minv = ko_raw_bits(dtype, values, 0), maxv = minv;
    for (int64_t i = 1; i < n; i++) {
        uint64_t v = ko_raw_bits(dtype, values, i);
        if (ko_is_signed(dtype) ? ((int64_t)v < (int64_t)minv) : (v < minv)) minv = v;
        if (ko_is_signed(dtype) ? ((int64_t)v > (int64_t)maxv) : (v > maxv)) maxv = v;
    }
    uint64_t range = maxv - minv;
    int log2 = range ? 64 - __builtin_clzll(range) : 0;      /* bits.Len64 */
    uint64_t mask = log2 >= 64 ? ~0ull : ((1ull << log2) - 1);
    int64_t o = 0;
    dst[o++] = 4;
    o += ko_put_uvarint(dst + o, minv);                        /* uint64(c.For) */
    o += ko_put_uvarint(dst + o, (uint64_t)log2);
    o += ko_put_uvarint(dst + o, (uint64_t)n);
    int64_t sz = (((int64_t)log2 * n + 63) & ~(int64_t)63) / 8;
    memset(dst + o, 0, (size_t)sz);
    if (log2 > 0) { /* encode(): LSB-first stream */
        uint64_t *out = (uint64_t *)malloc((size_t)sz + 8), word = 0;
        int offset = 0;
        int64_t k = 0;
        for (int64_t i = 0; i < n; i++) {
            uint64_t x = (ko_raw_bits(dtype, values, i) - minv) & mask;
            word |= x << offset;
            offset += log2;
            if (offset >= 64) {
                out[k++] = word;
                offset -= 64;
                word = offset > 0 ? (x >> (log2 - offset)) : 0;
            }
        }
        if (offset > 0) out[k++] = word;
        memcpy(dst + o, out, (size_t)k * 8);
        free(out);
    }
    return o + sz;
}
KO_EXPORT int64_t ko_bitpack_container_load(int dtype, const uint8_t *buf, int64_t len, void *out, int64_t cap) {
    int w = ko_width(dtype);
    if (!w || len < 4 || buf[0] != 4) return -1;
    uint64_t minv, l2, n;
    int64_t o = 1;
    o += ko_uvarint(buf + o, &minv);
    o += ko_uvarint(buf + o, &l2);
    o += ko_uvarint(buf + o, &n);
    int log2 = (int)l2;
    int64_t sz = (((int64_t)log2 * (int64_t)n + 63) & ~(int64_t)63) / 8;
    if (o + sz > len || (int64_t)n > cap || log2 > 64) return -2;
    uint64_t mask = log2 >= 64 ? ~0ull : ((1ull << log2) - 1);
    int64_t nwords = sz / 8;
    for (int64_t i = 0; i < (int64_t)n; i++) { /* DecodeValue decode.go:56-74 */
        uint64_t v = 0;
        if (log2 > 0) {
            int64_t idx = i * log2, pos = idx >> 6;
            int shift = (int)(idx & 63);
            uint64_t word, nextw = 0;
            memcpy(&word, buf + o + pos * 8, 8);
            word >>= shift;
            if (64 - shift < log2 && pos + 1 < nwords) { memcpy(&nextw, buf + o + (pos + 1) * 8, 8); word |= nextw << (64 - shift); }
            v = word & mask;
        }
        v += minv; /* T(word&mask) + minv wraps in T */
        memcpy((uint8_t *)out + i * w, &v, (size_t)w);
    }
    return (int64_t)n;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic column generator (shared SPEC with the device generator in
 * knoxdb_amd/csrc/knox_kernels.hip: both sides implement DESIGN.md "synthetic data" from the
 * text, neither calls the other).  Counter based so any row can be produced independently:
 *   r(i) = mix64(i + mix64(seed)),  mix64 = splitmix64 finalizer with the golden-gamma add.
 * kinds: 0 uniform int in [lo, lo+span) via mulhi (span==0 -> raw r); 1 float64 grid
 * k*2^-10, k in [0,2^20); 2 float64 uniform [0,1) = (r>>11)*2^-53; 3 row id (+lo);
 * 4 float64 "semantic" mix: 2^-20 of rows are NaN/+Inf/-Inf/-0/+0, the rest uniform in
 * [-2^20, 2^20).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint64_t gen_u64(int kind, uint64_t seedmix, uint64_t i, int64_t lo, uint64_t span) {
    uint64_t r = mix64(i + seedmix);
    switch (kind) {
    case 0: return span ? (uint64_t)lo + (uint64_t)(((unsigned __int128)r * span) >> 64) : r;
    case 1: { double d = (double)(r >> 44) * 0x1p-10; uint64_t u; memcpy(&u, &d, 8); return u; }
    case 2: { double d = (double)(r >> 11) * 0x1p-53; uint64_t u; memcpy(&u, &d, 8); return u; }
    case 3: return i + (uint64_t)lo;
    case 4: {
        if ((r & 0xfffff) == 0) {
            static const uint64_t sp[5] = {0x7ff8000000000000ULL, 0x7ff0000000000000ULL, 0xfff0000000000000ULL,
                                           0x8000000000000000ULL, 0x0000000000000000ULL};
            return sp[(r >> 20) % 5];
        }
        double d = ((double)(int64_t)(r >> 11) * 0x1p-53 - 0.5) * 0x1p21;
        uint64_t u; memcpy(&u, &d, 8); return u;
    }
    default: return 0;
    }
}
KO_EXPORT int ko_generate(int dtype, int kind, uint64_t seed, int64_t first_row, int64_t n, int64_t lo,
                          uint64_t span, void *dst) {
    uint64_t sm = mix64(seed);
    for (int64_t i = 0; i < n; i++) {
        uint64_t u = gen_u64(kind, sm, (uint64_t)(first_row + i), lo, span);
        switch (dtype) {
        case KO_I64: case KO_U64: case KO_F64: ((uint64_t *)dst)[i] = u; break;
        case KO_I32: case KO_U32: ((uint32_t *)dst)[i] = (uint32_t)u; break;
        case KO_F32: { double d; memcpy(&d, &u, 8); ((float *)dst)[i] = (kind == 1 || kind == 2 || kind == 4) ? (float)d : (float)(int64_t)u; break; }
        case KO_I16: case KO_U16: ((uint16_t *)dst)[i] = (uint16_t)u; break;
        case KO_I8: case KO_U8: ((uint8_t *)dst)[i] = (uint8_t)u; break;
        default: return -1;
        }
    }
    return 0;
}

/* the same generator over `threads` host threads (counter based: every slice is independent) -- bench.py's full-run
 * checksum regenerates all 2^30 rows on the host */
typedef struct { int dtype, kind; uint64_t seed; int64_t first_row, n, lo; uint64_t span; void *dst; int rc; } ko_genjob;
static void *ko_gen_worker(void *arg) {
    ko_genjob *j = (ko_genjob *)arg;
    j->rc = ko_generate(j->dtype, j->kind, j->seed, j->first_row, j->n, j->lo, j->span, j->dst);
    return NULL;
}
KO_EXPORT int ko_generate_mt(int dtype, int kind, uint64_t seed, int64_t first_row, int64_t n, int64_t lo, uint64_t span, void *dst,
                             int threads) {
    static const int width[11] = {0, 8, 4, 2, 1, 8, 4, 2, 1, 8, 4};
    if (dtype < 1 || dtype > 10) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    ko_genjob jobs[256];
    for (int t = 0; t < threads; t++) {
        const int64_t r0 = n * t / threads, r1 = n * (t + 1) / threads;
        jobs[t] = (ko_genjob){dtype, kind, seed, first_row + r0, r1 - r0, lo, span, (char *)dst + r0 * width[dtype], 0};
        pthread_create(&tid[t], NULL, ko_gen_worker, &jobs[t]);
    }
    int rc = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(tid[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline leg ("port"): what the reference does for `Between + sum` on one materialized
 * int64 pack -- cmp kernel -> bitset (match_num.go:559-566 -> cmp.Int64Between), bitset ->
 * selection vector (reader.go:432 bits.Indexes), then one SumReducer.Reduce per selected row
 * (result.go:134-150 -> reducer.go:173).  Scalar, single thread, like the reference's
 * one-goroutine pack loop (pack/table/query.go:195-219).
 * ---------------------------------------------------------------------------------------- */
KO_EXPORT int64_t ko_scan_i64_bw_sum(const int64_t *src, int64_t n, int64_t pack_rows, int64_t a, int64_t b,
                                     uint8_t *bits, uint32_t *sel_scratch, int64_t *sum_out) {
    int64_t total = 0, sum = 0;
    for (int64_t p = 0; p < n; p += pack_rows) {
        int64_t len = (n - p < pack_rows) ? n - p : pack_rows;
        uint8_t *pb = bits + (p >> 3);
        memset(pb, 0, (size_t)bit_field_len(len));
        int64_t cnt = cmp_i64_bw(src + p, len, a, b, pb);
        if (cnt == 0) continue; /* reader.go:338 bits.None() -> skip pack */
        int64_t k = ko_bits_indexes(pb, len, sel_scratch);
        for (int64_t j = 0; j < k; j++) sum = (int64_t)((uint64_t)sum + (uint64_t)src[p + sel_scratch[j]]);
        total += cnt;
    }
    *sum_out = sum;
    return total;
}
