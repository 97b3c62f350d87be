"""Reducer restatement (internal/reducer/reducer.go:138-429).  The reference has NO reducer
tests ("parity unpinned"): these are hand-derived cases plus an exact-arithmetic cross-check."""
import math
from fractions import Fraction

import numpy as np

from oracle import oracle as O


def test_int_reducers_hand_cases():
    v = np.array([5, -3, 9, 9, -3, 2], dtype=np.int64)
    assert O.reduce(O.I64, O.R_SUM, v) == (19, True, 0)
    assert O.reduce(O.I64, O.R_COUNT, v) == (6, True, 0)
    assert O.reduce(O.I64, O.R_MIN, v) == (-3, True, 1)   # first occurrence wins (strict >)
    assert O.reduce(O.I64, O.R_MAX, v) == (9, True, 2)
    assert O.reduce(O.I64, O.R_FIRST, v) == (5, True, 0)
    assert O.reduce(O.I64, O.R_LAST, v) == (2, True, 5)
    # Welford mean of ints, truncated toward zero by T(mean): mean = 19/6 = 3.1666 -> 3
    assert O.reduce(O.I64, O.R_MEAN, v)[0] == 3
    # selection bitset: rows 1,2,4 -> -3, 9, -3
    bits = np.array([0b00010110], dtype=np.uint8)
    assert O.reduce(O.I64, O.R_SUM, v, bits) == (3, True, 1)
    assert O.reduce(O.I64, O.R_MIN, v, bits) == (-3, True, 1)
    assert O.reduce(O.I64, O.R_LAST, v, bits) == (-3, True, 4)
    # nothing selected -> unseen
    assert O.reduce(O.I64, O.R_SUM, v, np.zeros(1, np.uint8)) == (0, False, -1)


def test_int_sum_wraps():
    v = np.array([2**63 - 1, 1, 5], dtype=np.int64)
    assert O.reduce(O.I64, O.R_SUM, v)[0] == -(2**63) + 5
    v8 = np.array([100, 100, 100], dtype=np.int8)
    assert O.reduce(O.I8, O.R_SUM, v8)[0] == np.int8(300 - 256)
    # CountReducer counts in T (reducer.go:138-148): 300 rows of uint8 wrap to 44
    assert O.reduce(O.U8, O.R_COUNT, np.zeros(300, np.uint8))[0] == 44


def test_float_min_max_nan_and_zero_semantics():
    nan = math.nan
    # NaN adopted only as the very first value, then sticky (reducer.go:291-296)
    v, seen, row = O.reduce(O.F64, O.R_MIN, np.array([nan, 1.0, -5.0]))
    assert math.isnan(v) and seen and row == 0
    assert O.reduce(O.F64, O.R_MIN, np.array([3.0, nan, 1.0])) == (1.0, True, 2)
    assert O.reduce(O.F64, O.R_MAX, np.array([3.0, nan, 7.0, nan])) == (7.0, True, 2)
    # ties keep the first occurrence: min(+0, -0) keeps +0, min(-0, +0) keeps -0
    v, _, row = O.reduce(O.F64, O.R_MIN, np.array([0.0, -0.0]))
    assert math.copysign(1, v) == 1 and row == 0
    v, _, row = O.reduce(O.F64, O.R_MIN, np.array([-0.0, 0.0]))
    assert math.copysign(1, v) == -1 and row == 0


def test_float_sum_mean_sequential_order():
    # left-to-right IEEE adds: (1e16 + 1) + 1 == 1e16 but 1 + 1 + 1e16 == 1e16 + 2
    assert O.reduce(O.F64, O.R_SUM, np.array([1e16, 1.0, 1.0]))[0] == 1e16
    assert O.reduce(O.F64, O.R_SUM, np.array([1.0, 1.0, 1e16]))[0] == 1e16 + 2
    # Welford by hand: n=1 mean=2; n=2 delta=2 mean=3; n=3 delta=6 mean=5
    assert O.reduce(O.F64, O.R_MEAN, np.array([2.0, 4.0, 9.0]))[0] == 5.0
    # var / std (m2 / (n-1)); [2,4,9]: m2 = 0 + 2*1 + 6*4 = 26 -> var 13
    assert O.reduce(O.F64, O.R_VAR, np.array([2.0, 4.0, 9.0]))[0] == 13.0
    assert O.reduce(O.F64, O.R_STD, np.array([2.0, 4.0, 9.0]))[0] == math.sqrt(13.0)
    assert math.isnan(O.reduce(O.F64, O.R_VAR, np.array([2.0]))[0])


def test_exact_sum_matches_fsum_and_fraction():
    rng = np.random.default_rng(11)
    for scale in (1.0, 1e280, 1e-300):
        x = (rng.standard_normal(5000) * scale * np.exp(rng.uniform(-30, 30, 5000))).astype(np.float64)
        s, m, cnt = O.exact_sum_f64(x)
        assert cnt == x.size
        assert s == math.fsum(x.tolist())
        fr = sum((Fraction(float(v)) for v in x), Fraction(0))
        assert m == float(fr / x.size)
    # cancellation + denormals
    x = np.array([1e308, 1e308, -1e308, -1e308, 5e-324, 5e-324, 5e-324], dtype=np.float64)
    s, m, _ = O.exact_sum_f64(x)
    assert s == 1.5e-323 and m == float(Fraction(1.5e-323) / 7)
    # specials
    assert math.isnan(O.exact_sum_f64(np.array([1.0, math.inf, -math.inf]))[0])
    assert O.exact_sum_f64(np.array([1.0, math.inf]))[0] == math.inf
    # selection mask
    bits = np.array([0b101], dtype=np.uint8)
    assert O.exact_sum_f64(np.array([1.0, 10.0, 100.0]), bits) == (101.0, 50.5, 2)


def test_exact_dataset_is_order_independent():
    # values k*2^-10 with k < 2^20 sum exactly in any order
    x = O.generate(O.F64, 1, 3, 0, 1 << 16)
    assert ((x * 1024) == np.floor(x * 1024)).all() and x.max() < 1024
    seq = O.reduce(O.F64, O.R_SUM, x)[0]
    assert seq == O.exact_sum_f64(x)[0] == float(np.sum(x[::-1]))


def test_bytes_match():
    vals = [b"alpha", b"beta", b"alpha", b"", b"alphA", b"alpha", b"x", b"beta", b"alpha", b"zz"]
    buf = np.frombuffer(b"".join(vals), dtype=np.uint8)
    off, ptr = 0, []
    for s in vals:
        ptr.append((off << 32) | len(s))
        off += len(s)
    bits, cnt = O.bytes_match(O.EQ, ptr, buf, b"alpha")
    assert cnt == 4 and bits.tolist() == [0b00100101, 0b01]
    bits, cnt = O.bytes_match(O.NE, ptr, buf, b"alpha")
    assert cnt == 6 and bits.tolist() == [0b11011010, 0b10]
    bits, cnt = O.bytes_match(O.EQ, ptr, buf, b"")
    assert cnt == 1 and bits.tolist() == [0b00001000, 0]
    # mask path: only mask-selected rows are set, count unknown (-1)
    mask = np.array([0b00000101, 0b11], dtype=np.uint8)
    bits, cnt = O.bytes_match(O.EQ, ptr, buf, b"alpha", mask=mask)
    assert cnt == -1 and bits.tolist() == [0b00000101, 0b01]


def test_generator_spec():
    a = O.generate(O.I64, 0, 2, 0, 4096, lo=-(1 << 40), span=1 << 41)
    assert a.min() >= -(1 << 40) and a.max() < (1 << 40)
    # counter based: any slice reproduces
    b = O.generate(O.I64, 0, 2, 1000, 96, lo=-(1 << 40), span=1 << 41)
    assert (a[1000:1096] == b).all()
    u = O.generate(O.F64, 2, 4, 0, 4096)
    assert u.min() >= 0 and u.max() < 1
    assert (O.generate(O.U64, 3, 0, 10, 5) == np.arange(10, 15, dtype=np.uint64)).all()


def test_cpu_baseline_leg_matches_pieces():
    x = O.generate(O.I64, 0, 2, 0, 3 * 4096 + 77, lo=-(1 << 40), span=1 << 41)
    bits, cnt, s = O.scan_i64_bw_sum(x, 4096, -(1 << 38), 1 << 38)
    sel = (x >= -(1 << 38)) & (x <= (1 << 38))
    assert cnt == int(sel.sum()) and s == int(x[sel].sum())


def test_inset_semantics():
    x = np.array([5, -3, 7, 5, 0, -128, 127], dtype=np.int8)
    assert O.inset(O.I8, False, x, [5, -128]).tolist() == [0b00101001]
    assert O.inset(O.I8, True, x, [5, -128]).tolist() == [0b01010110]
    # mask: only mask-selected rows are tested, result OR-ed into bits (bits.Set)
    assert O.inset(O.I8, False, x, [5, -128], mask=np.array([0b00100001], np.uint8), bits=np.array([0b01000000], np.uint8)).tolist() == [0b01100001]
    f = np.array([0.0, -0.0, np.nan, 1.5, np.inf])
    assert O.inset(O.F64, False, f, [np.nan, 0.0]).tolist() == [0b00011 & 0b11111]
    assert O.inset(O.F64, True, f, [np.nan, 0.0]).tolist() == [0b11100]
    assert O.inset(O.F64, False, f, []).tolist() == [0]


def test_simd_cpu_baseline_equals_literal_port():
    x = O.generate(O.I64, 0, 2, 0, 5 * 4096 + 100, lo=-(1 << 40), span=1 << 41)
    b0, c0, s0 = O.scan_i64_bw_sum(x, 4096, -(1 << 38), 1 << 38)
    for threads in (1, 3, 8):
        for level in (0, 1, 2, 3):   # best available / scalar / AVX2 / AVX-512 (a level the CPU lacks runs its best)
            b1, c1, s1 = O.simd_scan_i64_bw_sum(x, 4096, -(1 << 38), 1 << 38, threads, level=level)
            assert c1 == c0 and s1 == s0 and (b1 == b0).all(), (threads, level)
    # ragged lengths and the wrapped range (a > b) the reference's cmp_bw accepts (internal/cmp/number.go:211-243)
    for m in (1, 7, 8, 9, 63, 64, 65, 1000, 4097):
        for a, b in ((-(1 << 38), 1 << 38), (5, -5), (0, 0)):
            w = O.scan_i64_bw_sum(x[:m], 4096, a, b)
            for level in (2, 3):
                r = O.simd_scan_i64_bw_sum(x[:m], 4096, a, b, 1, level=level)
                assert r[1] == w[1] and r[2] == w[2] and (r[0] == w[0]).all(), (m, a, b, level)
