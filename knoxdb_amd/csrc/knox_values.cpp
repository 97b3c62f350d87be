// knox_values.cpp -- Reducer.Value() from a merged aggregate state and the merge of states across row ranges / GPUs:
// pure host code, no device needed.
#include "knox_internal.h"

// ---- Reducer.Value() from a merged state (pure host code) ------------------------------------
static void store_value(int dtype, void *out, uint64_t bits) { memcpy(out, &bits, (size_t)dtype_width(dtype)); }

// Reducer.Value(): the arithmetic lives in knox_finalize.h, shared with the device's per-window finaliser
KNOX_API int knox_agg_value(const knox_agg_state *st, int func, void *value_out, int *seen, int64_t *row) {
    if (!st || !value_out) return KNOX_E_INVALID;
    uint64_t bits = 0;
    int sn = 0;
    int64_t r = -1;
    const int rc = knox_fin::agg_value_bits(*st, func, &bits, &sn, &r);
    if (rc == -5) return KNOX_E_UNSUPPORTED;
    if (rc) return KNOX_E_INVALID;
    store_value(st->dtype, value_out, bits);
    if (seen) *seen = sn;
    if (row) *row = r;
    return KNOX_OK;
}


KNOX_API int knox_agg_combine(const knox_agg_state *states, int n, knox_agg_state *out) {
    if (!states || !out || n <= 0) return KNOX_E_INVALID;
    knox_agg_state r = states[0];
    // central moments: re-centre every part on the first non-empty part's shift K --
    //   sum (v-K) = D_i + n_i*d,  sum (v-K)^2 = Q_i + 2*d*D_i + n_i*d^2,  d = K_i - K  (binary128)
    bool mom_ok = true, mom_any = false;
    double mom_k = 0.0;
    __float128 mom_d = 0, mom_q = 0;
    for (int i = 0; i < n; i++) {
        const knox_agg_state &s = states[i];
        if (s.count <= 0) continue;
        if (!(s.flags & 2u)) { mom_ok = false; break; }
        if (!mom_any) { mom_k = s.m_shift; mom_any = true; }
        const __float128 d = (__float128)s.m_shift - (__float128)mom_k, ni = (__float128)s.count;
        const __float128 Di = (__float128)s.m_dev_hi + (__float128)s.m_dev_lo, Qi = (__float128)s.m_sq_hi + (__float128)s.m_sq_lo;
        mom_d += Di + ni * d;
        mom_q += Qi + 2 * d * Di + ni * d * d;
    }
    for (int i = 1; i < n; i++) {
        const knox_agg_state &s = states[i];
        if (s.dtype != r.dtype) return KNOX_E_INVALID;
        r.flags |= s.flags & 1u;
        r.count += s.count;
        r.isum += s.isum;
        {   // double-double add, same form as the device merge
            double hi = r.fsum_hi, lo = r.fsum_lo;
            double sum = hi + s.fsum_hi;
            double bb = sum - hi;
            double e = (hi - (sum - bb)) + (s.fsum_hi - bb);
            if (!(fabs(sum) <= 1.7976931348623157e308)) e = 0.0;
            r.fsum_lo = lo + s.fsum_lo + e;
            r.fsum_hi = sum;
        }
        if (s.min_row >= 0) {
            const uint64_t ks = order_key(r.dtype, s.min_bits), kr = order_key(r.dtype, r.min_bits);
            if (r.min_row < 0 || ks < kr || (ks == kr && s.min_row < r.min_row)) { r.min_bits = s.min_bits; r.min_row = s.min_row; }
        }
        if (s.max_row >= 0) {
            const uint64_t ks = order_key(r.dtype, s.max_bits), kr = order_key(r.dtype, r.max_bits);
            if (r.max_row < 0 || ks > kr || (ks == kr && s.max_row < r.max_row)) { r.max_bits = s.max_bits; r.max_row = s.max_row; }
        }
        if (s.first_row >= 0 && (r.first_row < 0 || s.first_row < r.first_row)) { r.first_row = s.first_row; r.first_bits = s.first_bits; }
        if (s.last_row >= 0 && (r.last_row < 0 || s.last_row > r.last_row)) { r.last_row = s.last_row; r.last_bits = s.last_bits; }
    }
    r.flags &= ~2u;
    r.m_shift = r.m_dev_hi = r.m_dev_lo = r.m_sq_hi = r.m_sq_lo = 0.0;
    if (mom_ok && mom_any) {
        r.flags |= 2u;
        r.m_shift = mom_k;
        r.m_dev_hi = (double)mom_d;
        r.m_dev_lo = (double)(mom_d - (__float128)r.m_dev_hi);
        r.m_sq_hi = (double)mom_q;
        r.m_sq_lo = (double)(mom_q - (__float128)r.m_sq_hi);
        if (!(fabs(r.m_dev_hi) <= 1.7976931348623157e308)) r.m_dev_lo = 0.0;
        if (!(fabs(r.m_sq_hi) <= 1.7976931348623157e308)) r.m_sq_lo = 0.0;
    }
    *out = r;
    return KNOX_OK;
}

