"""GPU parity for VarReducer / StdReducer (internal/reducer/reducer.go:352-429).

The reference runs Welford's recurrence row by row (unpinned by its own tests); the
device uses the corrected two-pass form with double-double sums.  Checked three ways:
  * against the EXACT variance (Python Fractions over float64(v)), correctly rounded: <= 1 ulp;
  * against the oracle's literal Welford restatement: relative 1e-9 on well-conditioned data
    (the tolerance covers Welford's own rounding, not ours);
  * merge of row-range states == the whole (knox_agg_combine).
"""
import math
from fractions import Fraction

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import NP_DTYPE, BlockType, FilterMode, ReducerFunc
from oracle import oracle as O
from helpers import oracle_match

pytestmark = pytest.mark.gpu
R = ReducerFunc
NUMERIC = [BlockType.Int64, BlockType.Uint64, BlockType.Float64, BlockType.Int32, BlockType.Uint32, BlockType.Float32,
           BlockType.Int16, BlockType.Uint16, BlockType.Int8, BlockType.Uint8]


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def exact_var(sel_f64):
    """correctly rounded (sum x^2 - (sum x)^2/n)/(n-1) over exact rationals"""
    n = len(sel_f64)
    if n < 2:
        return math.nan
    fr = [Fraction(float(v)) for v in sel_f64]
    s1 = sum(fr)
    s2 = sum(f * f for f in fr)
    return float((s2 - s1 * s1 / n) / (n - 1))


def ulps(a, b):
    if math.isnan(a) and math.isnan(b):
        return 0
    ia, ib = np.array([a, b], dtype=np.float64).view(np.int64)
    return abs(int(ia) - int(ib))


def make_col(bt, n, rng):
    npdt = NP_DTYPE[bt]
    if npdt in (np.float64, np.float32):
        return (rng.standard_normal(n) * 1000 + 123.0).astype(npdt)
    info = np.iinfo(npdt)
    if info.bits == 64:
        lo, hi = (-(1 << 62), 1 << 62) if info.min < 0 else (0, 1 << 63)
        return rng.integers(lo, hi, size=n, dtype=np.int64 if info.min < 0 else np.uint64).astype(npdt)
    return rng.integers(info.min, int(info.max) + 1, size=n).astype(npdt)


@pytest.mark.parametrize("bt", NUMERIC)
@pytest.mark.parametrize("filtered", [False, True])
def test_var_std_all_types(ctx, bt, filtered):
    rng = np.random.default_rng(100 + int(bt) + (50 if filtered else 0))
    n = 12 * 1024 + 37  # full tiles for every width + a ragged tail
    x = make_col(bt, n, rng)
    col = ctx.upload(x, 4096)
    if filtered:
        srt = np.sort(x)
        a, b = srt[n // 5].item(), srt[4 * n // 5].item()
        tree = And(Leaf(0, FilterMode.Range, a, b))
        sel = np.unpackbits(oracle_match(tree, [x], n), bitorder="little")[:n].astype(bool)
    else:
        tree = And()
        sel = np.ones(n, bool)
    res = ctx.scan(tree, [col], aggs=[(R.Var, 0), (R.Std, 0), (R.Mean, 0), (R.Count, 0)])
    assert res.count == int(sel.sum())
    xs = x[sel].astype(np.float64)  # float64(v) as the reducers convert
    want = exact_var(xs)
    gv = res.value(0)[0]
    gs = res.value(1)[0]
    dt = int(bt)
    selbits = np.packbits(sel, bitorder="little")
    ov = O.reduce(dt, O.R_VAR, x, selbits)[0]
    os_ = O.reduce(dt, O.R_STD, x, selbits)[0]
    npdt = NP_DTYPE[bt]
    if npdt == np.float64:
        assert ulps(float(gv), want) <= 1
        assert ulps(float(gs), math.sqrt(want)) <= 1
        assert abs(float(gv) - float(ov)) <= 1e-9 * abs(want)
        assert abs(float(gs) - float(os_)) <= 1e-9 * math.sqrt(want)
    elif npdt == np.float32:
        # T(m2/(n-1)): the float64 result rounded to float32
        assert abs(int(np.float32(gv).view(np.int32)) - int(np.float32(want).view(np.int32))) <= 1
        assert abs(int(np.float32(gs).view(np.int32)) - int(np.float32(math.sqrt(want)).view(np.int32))) <= 1
        assert abs(float(gv) - float(ov)) <= 1e-6 * abs(want)
    else:
        info = np.iinfo(npdt)
        # T(float64) truncates; values outside T are implementation-specific in Go -> compare in range only
        if want < float(info.max):
            assert abs(int(gv) - int(want)) <= max(1, int(2.3e-16 * want)), (gv, want)
            assert abs(int(gv) - int(ov)) <= max(1, int(1e-9 * want))
        if math.sqrt(want) < float(info.max):
            assert abs(int(gs) - int(math.sqrt(want))) <= max(1, int(2.3e-16 * math.sqrt(want)))
            assert abs(int(gs) - int(os_)) <= max(1, int(1e-9 * math.sqrt(want)))
    col.free()


def test_var_ill_conditioned_timestamps(ctx):
    """nanosecond timestamps ~1.7e18 with ~1e6 spread: mean^2/var ~ 1e24.  The textbook
    sum-of-squares form loses everything here; the centred second pass must stay exact."""
    rng = np.random.default_rng(7)
    n = 40000
    base = 1_700_000_000_000_000_000
    x = (base + rng.integers(-1_000_000, 1_000_000, size=n) * 256).astype(np.int64)  # multiples of 256: exact in float64
    col = ctx.upload(x, 8192)
    res = ctx.scan(And(), [col], aggs=[(R.Var, 0), (R.Std, 0)])
    want = exact_var(x.astype(np.float64))
    assert abs(int(res.value(0)[0]) - int(want)) <= 1
    assert abs(int(res.value(1)[0]) - int(math.sqrt(want))) <= 1
    # float64 view of the same problem: exact to the ulp
    xf = (1.0e15 + rng.standard_normal(n)).astype(np.float64)
    colf = ctx.upload(xf, 8192)
    resf = ctx.scan(And(), [colf], aggs=[(R.Var, 0), (R.Std, 0)])
    wantf = exact_var(xf)
    assert ulps(float(resf.value(0)[0]), wantf) <= 1
    assert ulps(float(resf.value(1)[0]), math.sqrt(wantf)) <= 1
    col.free()
    colf.free()


def test_var_small_counts_and_nonfinite(ctx):
    x = np.array([2.0, 4.0, 9.0, 1.5, -3.0], dtype=np.float64)
    col = ctx.upload(x, 8)
    # n == 0: not seen; n == 1: NaN, seen (reducer.go:376-379); n == 3 -> hand value 13
    for a, b, cnt in [(100.0, 200.0, 0), (9.0, 9.0, 1), (2.0, 9.0, 3)]:
        res = ctx.scan(And(Leaf(0, FilterMode.Range, a, b)), [col], aggs=[(R.Var, 0), (R.Std, 0)])
        v, seen, _ = res.value(0)
        s, seen2, _ = res.value(1)
        assert res.count == cnt
        assert seen == (cnt > 0) and seen2 == (cnt > 0)
        if cnt == 1:
            assert math.isnan(v) and math.isnan(s)
        if cnt == 3:
            assert v == 13.0 and s == math.sqrt(13.0)
    col.free()
    # any selected non-finite value makes Welford's m2 NaN (Inf - Inf); unselected ones do not matter
    y = np.array([1.0, 2.0, np.inf, 4.0, np.nan, 6.0, 7.0, 8.0], dtype=np.float64)
    coly = ctx.upload(y, 8)
    for tree in (And(), And(Leaf(0, FilterMode.Le, 3.0))):  # second: NaN dropped, -inf..3 keeps 1, 2
        res = ctx.scan(tree, [coly], aggs=[(R.Var, 0)])
        ov = O.reduce(O.F64, O.R_VAR, y, oracle_match(tree, [y], y.size))[0]
        gv = res.value(0)[0]
        assert (math.isnan(gv) and math.isnan(ov)) or gv == ov
    coly.free()


def test_var_masked_reduce_and_combine(ctx):
    """knox_reduce + knox_reduce_moments over three row ranges, merged with knox_agg_combine, equals
    the single-pass result and the exact value (different shifts K per part)."""
    rng = np.random.default_rng(11)
    n = 3 * 20000
    x = np.concatenate([rng.standard_normal(20000) * 3 + 1e6, rng.standard_normal(20000) * 50 - 4e5, rng.standard_normal(20000)]).astype(np.float64)
    maskbits = rng.integers(0, 2, size=n).astype(bool)
    parts = []
    for i in range(3):
        xi = x[i * 20000:(i + 1) * 20000]
        ci = ctx.upload(xi, 4096)
        mi = ctx.bits(20000)
        mi.upload(np.packbits(maskbits[i * 20000:(i + 1) * 20000], bitorder="little"))
        parts.append(ctx.reduce(ci, mi, row_base=i * 20000, moments=True))
        ci.free()
        mi.free()
    merged = K.agg_combine(parts)
    assert merged.flags & 2
    want = exact_var(x[maskbits])
    assert ulps(float(K.agg_value(merged, R.Var)[0]), want) <= 1
    assert ulps(float(K.agg_value(merged, R.Std)[0]), math.sqrt(want)) <= 1
    # a part without moments poisons the merge: VAR on it is refused, not silently wrong
    c0 = ctx.upload(x[:20000], 4096)
    nomom = ctx.reduce(c0)
    c0.free()
    m2 = K.agg_combine([parts[0], nomom])
    assert not (m2.flags & 2)
    with pytest.raises(K.KnoxError):
        K.agg_value(m2, R.Var)


def test_var_two_columns_and_multipass(ctx):
    """var on a column that is not the filter column (different width -> separate reduce pass)"""
    rng = np.random.default_rng(13)
    n = 50000
    a = rng.integers(0, 1000, size=n).astype(np.int32)
    b = (rng.standard_normal(n) * 10).astype(np.float64)
    ca, cb = ctx.upload(a, 8192), ctx.upload(b, 8192)
    tree = And(Leaf(0, FilterMode.Lt, 500))
    res = ctx.scan(tree, [ca, cb], aggs=[(R.Var, 1), (R.Sum, 0), (R.Std, 0)])
    sel = a < 500
    assert ulps(float(res.value(0)[0]), exact_var(b[sel])) <= 1
    assert int(res.value(1)[0]) == int(a[sel].astype(np.int64).sum())
    assert abs(int(res.value(2)[0]) - int(math.sqrt(exact_var(a[sel].astype(np.float64))))) <= 1
    ca.free()
    cb.free()


def test_var_per_time_window(ctx):
    """var/std per window (knox_reduce_buckets flags bit1): every window centred on its own mean;
    windows spanning several work items (> 65536 rows), tiny windows and empty windows."""
    rng = np.random.default_rng(17)
    n = 400000
    dt = rng.integers(0, 3, size=n)
    dt[100000:300000] = 0  # one window holding 200k rows -> several work items
    dt[rng.integers(0, n, size=50)] += 5000  # empty windows
    t = (1_700_000_000 + np.cumsum(dt)).astype(np.int64)
    step = 1000
    t_from = int(t[0])
    nb = int((t[-1] - t_from) // step) + 1
    tcol = ctx.upload(t, 4096)
    sel = rng.random(n) < 0.7
    mask = ctx.bits(n).upload(np.packbits(sel, bitorder="little"))
    for npdt in (np.float64, np.int32, np.uint8):
        if npdt == np.float64:
            v = (rng.standard_normal(n) * 5 + 1e7 + np.arange(n) * 0.01).astype(npdt)
        else:
            info = np.iinfo(npdt)
            v = rng.integers(info.min, int(info.max) + 1, size=n).astype(npdt)
        vcol = ctx.upload(v, 4096)
        states = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=mask, moments=True)
        win = (t - t_from) // step
        checked = 0
        for k in list(range(0, nb, max(1, nb // 40))) + [int(win[150000])]:
            xs = v[(win == k) & sel].astype(np.float64)
            st = states[k]
            assert st.count == xs.size
            if xs.size == 0:
                continue
            assert st.flags & 2
            gv, seen, _ = K.agg_value(st, R.Var)
            assert seen
            if xs.size < 2:
                if npdt == np.float64:
                    assert math.isnan(gv)
                continue
            want = exact_var(xs)
            if npdt == np.float64:
                assert ulps(float(gv), want) <= 1, k
                assert ulps(float(K.agg_value(st, R.Std)[0]), math.sqrt(want)) <= 1
            elif want < float(np.iinfo(npdt).max):
                assert abs(int(gv) - int(want)) <= 1
            checked += 1
        assert checked > 10
        # windows merge like row ranges: all windows combined == the global variance
        allst = K.agg_combine([s for s in states])
        wantall = exact_var(v[sel & (win >= 0) & (win < nb)].astype(np.float64))
        if npdt == np.float64:
            assert ulps(float(K.agg_value(allst, R.Var)[0]), wantall) <= 1
        vcol.free()
    tcol.free()
    mask.free()
