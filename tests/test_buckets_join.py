"""'Next' row f1: the reference's join reducers (internal/reducer/reducer.go:432-731) and the literal
per-window mode.  The reference holds no tests for these reducers, so the oracle is pinned by hand-worked
cases and by an independent Python model written from the same source lines; the GPU
(knox_reduce_buckets_sequential, one lane per window) must equal the oracle bit for bit for EVERY reducer."""
import math

import numpy as np
import pytest

from oracle import oracle as O

JOIN = [O.R_FIRST_JOIN, O.R_LAST_JOIN, O.R_MIN_JOIN, O.R_MAX_JOIN, O.R_MEAN_JOIN, O.R_VAR_JOIN, O.R_STD_JOIN]
PLAIN = [O.R_SUM, O.R_FIRST, O.R_LAST, O.R_MIN, O.R_MAX, O.R_MEAN, O.R_VAR, O.R_STD, O.R_COUNT]


def model_window(func, times, vals, npdt):
    """One window's join reducer, modelled on groups: rows with equal raw timestamps form one group whose
    members are summed (in T) before the reducer proper runs."""
    T = np.dtype(npdt).type
    groups, prev = [], None
    with np.errstate(over="ignore"):
        for t, x in zip(times, vals):
            if prev is not None and t == prev:
                groups[-1] = T(groups[-1] + T(x))
            else:
                groups.append(T(x))
            prev = t
    if not groups:
        return None
    if func == O.R_FIRST_JOIN:
        return groups[0]
    if func == O.R_LAST_JOIN:
        return groups[-1]
    if func == O.R_MAX_JOIN:   # r.v starts at the zero value (reducer.go:497-518): max(0, groups)
        return max([T(0)] + groups)
    if func == O.R_MIN_JOIN:   # likewise min(0, groups)
        return min([T(0)] + groups)
    n, mean, m2 = 0, 0.0, 0.0
    for g in groups:           # Welford over the group sums, in float64
        n += 1
        d = float(g) - mean
        mean += d / n
        m2 += d * (float(g) - mean)
    def conv(x):   # Go's T(float64): out-of-range and NaN -> integer conversions are platform defined, not compared
        if np.dtype(npdt).kind == "f":
            return T(x)
        info = np.iinfo(npdt)
        if not math.isfinite(x) or not (info.min <= math.trunc(x) <= info.max) or abs(x) >= 2.0 ** 62:
            return None
        return T(math.trunc(x))

    if func == O.R_MEAN_JOIN:
        return conv(mean)
    if n < 2:
        return conv(math.nan)
    var = m2 / (n - 1)
    return conv(var) if func == O.R_VAR_JOIN else conv(math.sqrt(var))


def test_oracle_join_hand_cases():
    # raw times: two rows at 100, one at 103, three at 107 | next window: 110, 110, 115
    t = np.array([100, 100, 103, 107, 107, 107, 110, 110, 115], dtype=np.int64)
    v = np.array([5, 7, -2, 1, 1, 1, 10, -30, 4], dtype=np.int64)
    want = {
        O.R_FIRST_JOIN: [12, -20],   # first group summed
        O.R_LAST_JOIN: [3, 4],       # last group summed
        O.R_MAX_JOIN: [12, 4],       # groups 12,-2,3 | -20,4
        O.R_MIN_JOIN: [-2, -20],     # min(0, ...)
        O.R_MEAN_JOIN: [4, -8],      # (12-2+3)/3 = 4.33 -> 4 ; (-20+4)/2 = -8
    }
    for f, w in want.items():
        bt, bv = O.buckets(O.I64, f, t, v, None, 100, 10)
        assert bt.tolist() == [100, 110] and bv.tolist() == w, f
    # MaxJoin over negative groups only: the zero start value wins (the reference's behaviour, kept)
    bt, bv = O.buckets(O.I64, O.R_MAX_JOIN, np.array([1, 2], np.int64), np.array([-5, -7], np.int64), None, 0, 10)
    assert bv.tolist() == [0]
    # var / std of the group sums, float
    tf = np.array([1, 1, 2, 3, 3], dtype=np.int64)
    vf = np.array([1.0, 2.0, 5.0, 0.5, 0.5], dtype=np.float64)   # groups 3, 5, 1 -> mean 3, var 4, std 2
    assert O.buckets(O.F64, O.R_VAR_JOIN, tf, vf, None, 0, 10)[1].tolist() == [4.0]
    assert O.buckets(O.F64, O.R_STD_JOIN, tf, vf, None, 0, 10)[1].tolist() == [2.0]
    assert O.buckets(O.F64, O.R_MEAN_JOIN, tf, vf, None, 0, 10)[1].tolist() == [3.0]
    # one group only: variance undefined -> NaN
    assert math.isnan(O.buckets(O.F64, O.R_VAR_JOIN, tf[:2], vf[:2], None, 0, 10)[1][0])
    # a deselected row between two equal timestamps does not break the join (series.go compares streamed rows)
    bits = np.array([0b101], np.uint8)
    bt, bv = O.buckets(O.I64, O.R_FIRST_JOIN, np.array([4, 4, 4], np.int64), np.array([1, 100, 2], np.int64), bits, 0, 10)
    assert bv.tolist() == [3]


@pytest.mark.parametrize("o,npdt", [(O.I64, np.int64), (O.I16, np.int16), (O.U8, np.uint8), (O.F64, np.float64), (O.F32, np.float32)])
def test_oracle_join_matches_group_model(o, npdt):
    rng = np.random.default_rng(o * 7 + 3)
    n, step = 4000, 50
    t = (1000 + np.cumsum(rng.integers(0, 3, size=n))).astype(np.int64)   # many repeated timestamps
    if np.dtype(npdt).kind == "f":
        v = (rng.integers(-1 << 12, 1 << 12, n) / 8.0).astype(npdt)
    else:
        info = np.iinfo(npdt)
        v = rng.integers(info.min, info.max, n, endpoint=True).astype(npdt)
    sel = rng.random(n) < 0.7
    bits = np.packbits(sel, bitorder="little")
    t_from = 990
    win = (t - t_from) // step
    for f in JOIN:
        bt, bv = O.buckets(o, f, t, v, bits, t_from, step)
        ks = sorted(set(win[sel].tolist()))
        assert bt.tolist() == [t_from + k * step for k in ks]
        for got, k in zip(bv, ks):
            rows = sel & (win == k)
            want = model_window(f, t[rows], v[rows], npdt)
            if want is None:
                continue
            assert np.array([got], npdt).tobytes() == np.array([want], npdt).tobytes() or (got != got and want != want), (f, k, got, want)


def representable(func, t, v, bits, t_from, step, ks, npdt):
    """Range guide for the integer mean / var / std reducers: which windows (of ks) have a float result that
    converts to T without leaving its range (approximate float64 statistics; margins keep clear of the edge)."""
    n = t.size
    if ks.size == 0:
        return np.zeros(0, bool)
    sel = np.unpackbits(bits, bitorder="little")[:n].astype(bool) if bits is not None else np.ones(n, bool)
    ts, vs = t[sel], v[sel]
    if func >= O.R_FIRST_JOIN:   # group sums wrap in T
        starts = np.flatnonzero(np.r_[True, ts[1:] != ts[:-1]])
        with np.errstate(over="ignore"):
            vs = np.add.reduceat(vs, starts, dtype=npdt)
        ts = ts[starts]
    k = (ts - t_from) // step
    x = vs.astype(np.float64)
    nb = int(ks.max()) + 1
    cnt = np.bincount(k, minlength=nb).astype(np.float64)
    with np.errstate(all="ignore"):
        mean = np.bincount(k, weights=x, minlength=nb) / cnt
        var = np.bincount(k, weights=(x - mean[k]) ** 2, minlength=nb) / (cnt - 1)
    info = np.iinfo(npdt)
    lim = min(float(info.max), 2.0 ** 62) * 0.99
    if func in (O.R_MEAN, O.R_MEAN_JOIN):
        res, need = mean, 1
    else:
        res, need = (var if func in (O.R_VAR, O.R_VAR_JOIN) else np.sqrt(var)), 2
    ok = (cnt >= need) & np.isfinite(res) & (np.abs(res) < lim)
    return ok[ks]


@pytest.mark.gpu
def test_gpu_sequential_windows_are_bit_exact_for_every_reducer():
    import knoxdb_amd as K
    from knoxdb_amd.types import BlockType, ReducerFunc as R
    rng = np.random.default_rng(97)
    with K.Context(0) as ctx:
        for n, step, gaps in ((20000, 500, False), (150000, 33, True), (777, 5, True), (1, 10, False)):
            dt = rng.integers(0, 3, size=n)
            if gaps and n > 10:
                dt[rng.integers(0, n, size=max(1, n // 300))] += step * 4
            t = (1_700_000_000 + np.cumsum(dt)).astype(np.int64)
            t_from = int(t[0]) - 7
            nb = int((t[-1] - t_from) // step) + 1
            tcol = ctx.upload(t, 4096)
            bits_h = np.packbits(rng.random(n) < 0.65, bitorder="little")
            mask = ctx.bits(n).upload(bits_h)
            for bt_k, o, npdt in ((BlockType.Int64, O.I64, np.int64), (BlockType.Int32, O.I32, np.int32), (BlockType.Int8, O.I8, np.int8),
                                  (BlockType.Uint64, O.U64, np.uint64), (BlockType.Uint16, O.U16, np.uint16),
                                  (BlockType.Float64, O.F64, np.float64), (BlockType.Float32, O.F32, np.float32)):
                if np.dtype(npdt).kind == "f":
                    v = (rng.standard_normal(n) * 1e3).astype(npdt)   # arbitrary floats: the order of operations must match
                    v[rng.integers(0, n, 3)] = -0.0
                else:
                    info = np.iinfo(npdt)
                    v = rng.integers(info.min, info.max, n, endpoint=True, dtype=npdt)
                vcol = ctx.upload(v, 4096)
                isf = np.dtype(npdt).kind == "f"
                for m, mh in ((None, None), (mask, bits_h)):
                    for f in PLAIN + JOIN:
                        want_t, want_v0 = O.buckets(o, f, t, v, mh, t_from, step)
                        # both forms of the kernel: one lane per window, one wave per window (the default picks by window count)
                        for form in ("lane", "wave"):
                            want_v = want_v0
                            vals, seen = ctx.reduce_buckets_sequential(tcol, vcol, R(f), t_from, step, nb, mask=m, form=form)
                            ks = ((want_t - t_from) // step).astype(np.int64)
                            # every window the reference created a reducer for is seen, and no other
                            assert np.flatnonzero(seen).tolist() == ks.tolist(), (n, bt_k, f, form)
                            got = vals[ks]
                            if not isf and f in (O.R_VAR, O.R_STD, O.R_VAR_JOIN, O.R_STD_JOIN, O.R_MEAN, O.R_MEAN_JOIN):
                                # T(NaN) / out-of-range float->int conversions are platform defined in Go: compare the
                                # windows whose float result is representable in T
                                ok = representable(f, t, v, mh, t_from, step, ks, npdt)
                                got, want_v = got[ok], want_v[ok]
                            assert got.tobytes() == want_v.tobytes(), (n, bt_k, f, form, got[:4], want_v[:4])
                vcol.free()
            tcol.free(); mask.free()
        # out-of-order timestamps are refused here too
        t = np.array([5, 6, 7, 3, 9, 10] * 100, dtype=np.int64)
        tcol, vcol = ctx.upload(t), ctx.upload(np.arange(600, dtype=np.int64))
        with pytest.raises(K.KnoxError) as e:
            ctx.reduce_buckets_sequential(tcol, vcol, R.MaxJoin, 0, 4, 3)
        assert e.value.code == -5
        tcol.free(); vcol.free()
