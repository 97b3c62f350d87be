"""Host-side (no GPU) checks of the var/std finalisation and merge in the C ABI:
knox_agg_value(KNOX_R_VAR / KNOX_R_STD) and knox_agg_combine over hand-built states whose
moments are computed here with exact rationals (reducer.go:352-429 semantics)."""
import math
from fractions import Fraction

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd._lib import AggState
from knoxdb_amd.types import BlockType, ReducerFunc as R


def _split(fr):
    hi = float(fr)
    return hi, float(fr - Fraction(hi))


def make_state(xs, dtype=BlockType.Float64, shift=None):
    st = AggState()
    st.dtype = int(dtype)
    st.flags = 3
    st.count = len(xs)
    fr = [Fraction(float(v)) for v in xs]
    st.fsum_hi, st.fsum_lo = _split(sum(fr)) if xs.size else (0.0, 0.0)
    k = (st.fsum_hi / st.count if st.count else 0.0) if shift is None else shift
    st.m_shift = k
    st.m_dev_hi, st.m_dev_lo = _split(sum(f - Fraction(k) for f in fr)) if xs.size else (0.0, 0.0)
    st.m_sq_hi, st.m_sq_lo = _split(sum((f - Fraction(k)) ** 2 for f in fr)) if xs.size else (0.0, 0.0)
    st.first_row = st.last_row = st.min_row = st.max_row = 0 if xs.size else -1
    return st


def exact_var(xs):
    n = len(xs)
    fr = [Fraction(float(v)) for v in xs]
    s1, s2 = sum(fr), sum(f * f for f in fr)
    return float((s2 - s1 * s1 / n) / (n - 1))


def test_value_var_std_float64():
    rng = np.random.default_rng(1)
    xs = rng.standard_normal(500) * 7 + 1e9
    st = make_state(xs)
    v, seen, _ = K.agg_value(st, R.Var)
    s, _, _ = K.agg_value(st, R.Std)
    assert seen and v == exact_var(xs) and s == math.sqrt(exact_var(xs))
    # a deliberately bad shift is corrected by the D*D/n term
    st2 = make_state(xs, shift=1e9 + 3.0)
    assert K.agg_value(st2, R.Var)[0] == exact_var(xs)


def test_value_small_counts():
    one = make_state(np.array([5.0]))
    v, seen, _ = K.agg_value(one, R.Var)
    assert seen and math.isnan(v)
    none = make_state(np.array([]))
    v, seen, _ = K.agg_value(none, R.Std)
    assert not seen and v == 0.0


def test_value_integer_truncates():
    xs = np.array([1, 2, 4, 8, 16], dtype=np.int64)
    st = make_state(xs.astype(np.float64), dtype=BlockType.Int64)
    want = exact_var(xs.astype(np.float64))  # 37.2
    assert int(K.agg_value(st, R.Var)[0]) == int(want) == 37
    assert int(K.agg_value(st, R.Std)[0]) == int(math.sqrt(want)) == 6
    u8 = make_state(np.array([0.0, 255.0] * 10), dtype=BlockType.Uint8)
    assert int(K.agg_value(u8, R.Std)[0]) == int(math.sqrt(exact_var(np.array([0.0, 255.0] * 10))))
    # var = 17113.4 does not fit uint8: implementation-specific in Go, 0 here
    assert int(K.agg_value(u8, R.Var)[0]) == 0


def test_combine_recentres_parts():
    rng = np.random.default_rng(2)
    parts_x = [rng.standard_normal(300) + 1e6, rng.standard_normal(10) * 100 - 5e5, np.array([]), rng.standard_normal(77)]
    states = [make_state(x) for x in parts_x]
    merged = K.agg_combine(states)
    allx = np.concatenate(parts_x)
    assert merged.count == allx.size and (merged.flags & 2)
    assert K.agg_value(merged, R.Var)[0] == exact_var(allx)
    # merge order = row order; a different grouping gives the same value
    m2 = K.agg_combine([K.agg_combine(states[:2]), K.agg_combine(states[2:])])
    assert K.agg_value(m2, R.Var)[0] == exact_var(allx)


def test_missing_moments_is_refused():
    st = make_state(np.array([1.0, 2.0, 3.0]))
    st.flags = 1
    with pytest.raises(K.KnoxError):
        K.agg_value(st, R.Var)
    good = make_state(np.array([4.0, 5.0]))
    assert not (K.agg_combine([good, st]).flags & 2)


def test_mean_and_var_are_the_correctly_rounded_exact_values_on_random_states():
    """knox_finalize.h evaluates exact_sum / n and (Q - D*D/n) / (n-1) in double-double; the same code finalises the
    per-window values on the device.  Against exact rationals, rounded once, over many random states incl. huge counts,
    cancellation (D far from 0) and sums whose low word matters."""
    rng = np.random.default_rng(20260)
    st = AggState()
    st.dtype = int(BlockType.Float64)
    st.flags = 3
    st.first_row = st.last_row = st.min_row = st.max_row = 0
    for _ in range(4000):
        n = int(rng.choice([1, 2, 3, 7, 10, 1000, 65536, 10 ** 6 + 3, 2 ** 30, 2 ** 31 - 1, int(rng.integers(1, 2 ** 32))]))
        hi = float(rng.standard_normal() * 10.0 ** int(rng.integers(-5, 15)))
        lo = float(np.spacing(hi) * rng.uniform(-0.5, 0.5))
        st.count, st.fsum_hi, st.fsum_lo = n, hi, lo
        want = float((Fraction(hi) + Fraction(lo)) / n)
        assert K.agg_value(st, R.Mean)[0] == want, (hi, lo, n)
        assert K.agg_value(st, R.Sum)[0] == float(Fraction(hi) + Fraction(lo))
        if n >= 2:
            q = abs(hi) * float(rng.uniform(0.5, 2.0)) + 1.0
            ql = float(np.spacing(q) * rng.uniform(-0.5, 0.5))
            d = float(rng.uniform(-1, 1)) * math.sqrt(q * n) * 0.7          # D*D/n <= Q/2: m2 stays positive
            dl = float(np.spacing(d) * rng.uniform(-0.5, 0.5))
            st.m_sq_hi, st.m_sq_lo, st.m_dev_hi, st.m_dev_lo = q, ql, d, dl
            D, Q = Fraction(d) + Fraction(dl), Fraction(q) + Fraction(ql)
            want_v = float((Q - D * D / n) / (n - 1))
            got = K.agg_value(st, R.Var)[0]
            assert got == want_v, (q, ql, d, dl, n)
            assert K.agg_value(st, R.Std)[0] == math.sqrt(want_v)
