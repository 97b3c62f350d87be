// knox_finalize.h -- Reducer.Value() from a mergeable aggregate state, ONE implementation for the host
// (knox_agg_value) and the device (window_values_kernel: 8 B per window cross PCIe instead of the 144-byte state), so
// the two agree bit for bit by construction.
//
// The arithmetic that needs more than a double -- exact_sum / n for the mean, (Q - D*D/n) / (n-1) for var / std -- runs
// in double-double (~106 bits, error-free transformations on fma): results are the correctly rounded exact values
// unless the exact value lies within ~2^-104 (relative) of a rounding boundary without being on it.
// Compile with -ffp-contract=off (the Makefile does): the error-free transformations must not be fused.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
// (always inlined in device code: called with a state that lives in registers -- as an out-of-line call the 144-byte state would be
//  spilled to scratch memory to pass it by reference)
#define KNOX_HD __host__ __device__ __attribute__((always_inline))
#else
#define KNOX_HD
#endif

namespace knox_fin {

struct dd {
    double hi, lo;
};

KNOX_HD inline dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
KNOX_HD inline dd fast_two_sum(double a, double b) {  // |a| >= |b| (or a == 0)
    const double s = a + b;
    return {s, b - (s - a)};
}
KNOX_HD inline dd two_prod(double a, double b) {
    const double p = a * b;
    return {p, __builtin_fma(a, b, -p)};
}
KNOX_HD inline dd dd_add(dd a, dd b) {  // accurate (IEEE-style) double-double addition
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return fast_two_sum(s.hi, s.lo);
}
KNOX_HD inline dd dd_neg(dd a) { return {-a.hi, -a.lo}; }
KNOX_HD inline dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return fast_two_sum(p.hi, p.lo);
}
KNOX_HD inline dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.hi, b);
    p.lo += a.lo * b;
    return fast_two_sum(p.hi, p.lo);
}
// a / n by long division: three quotient digits, the first two folded exactly
KNOX_HD inline dd dd_div_d(dd a, double n) {
    const double q1 = a.hi / n;
    dd r = dd_add(a, dd_neg(dd_mul_d({q1, 0.0}, n)));
    const double q2 = r.hi / n;
    r = dd_add(r, dd_neg(dd_mul_d({q2, 0.0}, n)));
    const double q3 = r.hi / n;
    dd q = fast_two_sum(q1, q2);
    q.lo += q3;
    return fast_two_sum(q.hi, q.lo);
}
// exact 128-bit integer sum (|S| < 2^106, true for any column that fits one GPU) -> double-double with hi + lo == S
KNOX_HD inline void int128_to_dd(uint64_t xlo, int64_t xhi, double &hi, double &lo) {
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
KNOX_HD inline bool finite_(double x) { return fabs(x) <= 1.7976931348623157e308; }

KNOX_HD inline int width_of(int dtype) {  // knox_dtype -> bytes per value (numeric types)
    switch (dtype) {
    case 1: case 5: case 9: return 8;
    case 2: case 6: case 10: return 4;
    case 3: case 7: return 2;
    case 4: case 8: return 1;
    default: return 0;
    }
}
KNOX_HD inline bool is_float(int dtype) { return dtype == 9 || dtype == 10; }
KNOX_HD inline bool is_signed(int dtype) { return dtype >= 1 && dtype <= 4; }

KNOX_HD inline double bits_f64(uint64_t b) {
    double d;
    memcpy(&d, &b, 8);
    return d;
}
KNOX_HD inline uint64_t f64_bits(double d) {
    uint64_t b;
    memcpy(&b, &d, 8);
    return b;
}
KNOX_HD inline uint64_t f32_bits(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return b;
}
KNOX_HD inline double value_as_f64(int dtype, uint64_t bits) {
    if (dtype == 9) return bits_f64(bits);
    float f;
    const uint32_t u = (uint32_t)bits;
    memcpy(&f, &u, 4);
    return (double)f;
}
// a float64 result in the column's float type, as raw bits
KNOX_HD inline uint64_t float_bits(int dtype, double d) { return dtype == 9 ? f64_bits(d) : f32_bits((float)d); }

// exact sum of a float column, rounded once: hi + lo IS the correctly rounded sum of the two doubles
KNOX_HD inline double sum_value(double hi, double lo) { return finite_(hi) ? hi + lo : hi; }

// exact_sum / n rounded once (reference: sequential Welford)
KNOX_HD inline double mean_value(double hi, double lo, int64_t count) {
    if (!finite_(hi)) return hi;
    const dd q = dd_div_d({hi, lo}, (double)count);
    return q.hi + q.lo;
}

// VarReducer.Value (reducer.go:376-381): m2 / (n-1) with m2 = Q - D*D/n from the second-pass moments
KNOX_HD inline double var_value(double q_hi, double q_lo, double d_hi, double d_lo, int64_t count) {
    if (count < 2) return NAN;
    if (!finite_(q_hi)) return q_hi + q_lo;  // non-finite input: NaN like Welford
    const double n = (double)count;
    const dd D = {d_hi, d_lo};
    dd m2 = dd_add({q_hi, q_lo}, dd_neg(dd_div_d(dd_mul(D, D), n)));
    if (m2.hi < 0) m2 = {0.0, 0.0};  // cannot happen mathematically; guards the last-bit residue
    const dd v = dd_div_d(m2, n - 1.0);
    return v.hi + v.lo;
}

// T(float64) for an integer T: truncation toward zero; NaN / out-of-range results are implementation-specific in Go
// ("the behavior is implementation-dependent") -> 0 here
KNOX_HD inline uint64_t int_from_f64(int dtype, double v) {
    const int w = width_of(dtype);
    const double lim = ldexp(1.0, 8 * w - (is_signed(dtype) ? 1 : 0));
    if (!(v == v) || v >= lim || v <= -lim) return 0;
    if (is_signed(dtype)) return (uint64_t)(int64_t)v;
    return v > 0 ? (uint64_t)v : 0;
}

// S: knox_agg_state (host) or AggStateDev (device) -- the same layout.  Returns 0 ok, -1 invalid, -5 unsupported;
// *bits = the value's raw bits zero-extended (the caller stores width_of(dtype) bytes), *seen = Value()'s ok,
// *row = the row whose timestamp Time() would report.
template <typename S>
KNOX_HD inline int agg_value_bits(const S &st, int func, uint64_t *bits, int *seen, int64_t *row) {
    // EVERY field is read here, unconditionally, and the branches below work on these copies: fields read inside the switch's cases
    // are merged by the compiler into loads through a selected ADDRESS, which keeps a state that lives in registers --
    // bucket_finalize_kernel's -- in scratch memory (found by tests/test_scan_shapes.py: ScratchSize 152)
    const int dtype = st.dtype;
    const uint32_t flags = st.flags;
    const int64_t count = st.count;
    const uint64_t isum = st.isum;
    const double fsum_hi = st.fsum_hi, fsum_lo = st.fsum_lo;
    const uint64_t min_bits = st.min_bits, max_bits = st.max_bits, first_bits = st.first_bits, last_bits = st.last_bits;
    const int64_t min_row = st.min_row, max_row = st.max_row, first_row = st.first_row, last_row = st.last_row;
    const double m_dev_hi = st.m_dev_hi, m_dev_lo = st.m_dev_lo, m_sq_hi = st.m_sq_hi, m_sq_lo = st.m_sq_lo;
    const int w = width_of(dtype);
    if (w == 0) return -1;
    const bool isf = is_float(dtype);
    const bool any = count > 0;
    *seen = any ? 1 : 0;
    *row = any ? first_row : -1;
    *bits = 0;
    if (!any) return 0;
    const uint64_t m = w == 8 ? ~0ull : ((1ull << (8 * w)) - 1);
    switch (func) {
    case 8:  // KNOX_R_COUNT: CountReducer counts in T (reducer.go:138-148)
        *bits = isf ? float_bits(dtype, dtype == 9 ? (double)count : (double)(float)count) : ((uint64_t)count & m);
        break;
    case 0:  // KNOX_R_SUM
        *bits = isf ? float_bits(dtype, sum_value(fsum_hi, fsum_lo)) : (isum & m);  // wraps in T like Go
        break;
    case 5: {  // KNOX_R_MEAN
        const double mean = mean_value(fsum_hi, fsum_lo, count);
        if (isf) *bits = float_bits(dtype, mean);
        else if (is_signed(dtype)) *bits = (uint64_t)(int64_t)mean & m;  // T(r.mean) truncates toward zero
        else *bits = (uint64_t)mean & m;
        break;
    }
    case 1:  // KNOX_R_FIRST
        *bits = first_bits & m;
        *row = first_row;
        break;
    case 2:  // KNOX_R_LAST
        *bits = last_bits & m;
        *row = last_row;
        break;
    case 3:    // KNOX_R_MIN
    case 4: {  // KNOX_R_MAX
        // a NaN first value is adopted and then sticks (reducer.go:256-296): r.v < NaN and r.v > NaN are false
        const double f = isf ? value_as_f64(dtype, first_bits) : 0.0;
        const uint64_t xb = func == 3 ? min_bits : max_bits;
        const int64_t xr = func == 3 ? min_row : max_row;
        if ((isf && f != f) || xr < 0) {
            *bits = first_bits & m;
            *row = first_row;
            break;
        }
        *bits = xb & m;
        *row = xr;
        break;
    }
    case 6:    // KNOX_R_VAR
    case 7: {  // KNOX_R_STD: VarReducer / StdReducer.Value (reducer.go:376-381, 416-421), NaN for n < 2
        if (!(flags & 2u)) return -1;  // state carries no moments (knox_reduce_moments)
        double v = var_value(m_sq_hi, m_sq_lo, m_dev_hi, m_dev_lo, count);
        if (func == 7 && count >= 2) v = sqrt(v);
        *bits = isf ? float_bits(dtype, v) : (int_from_f64(dtype, v) & m);
        break;
    }
    default:
        return -5;
    }
    return 0;
}

}  // namespace knox_fin
