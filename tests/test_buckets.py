"""'Next' row f1: time-window aggregation.  Oracle = literal NativeBucket.Push
(internal/reducer/bucket_native.go:104-159) incl. the out-of-order insert path; GPU =
knox_reduce_buckets over a non-decreasing time column (out-of-order input is refused)."""
import math

import numpy as np
import pytest

from oracle import oracle as O

FUNCS = [O.R_SUM, O.R_COUNT, O.R_MIN, O.R_MAX, O.R_FIRST, O.R_LAST]


def test_oracle_bucket_push_hand_cases():
    t = np.array([100, 105, 109, 110, 125, 131, 139, 150], dtype=np.int64)
    v = np.arange(1, 9, dtype=np.int64)
    bt, bv = O.buckets(O.I64, O.R_SUM, t, v, None, 100, 10)
    assert bt.tolist() == [100, 110, 120, 130, 150] and bv.tolist() == [6, 4, 5, 13, 8]
    # out-of-order rows: reducers are inserted / found by time (bucket_native.go:123-141)
    t2 = np.array([100, 125, 105, 131, 110, 112], dtype=np.int64)
    bt, bv = O.buckets(O.I64, O.R_SUM, t2, v[:6], None, 100, 10)
    assert bt.tolist() == [100, 110, 120, 130] and bv.tolist() == [4, 11, 2, 4]
    # windows are relative to `from`, not to zero (Interval.TruncateRelative)
    bt, bv = O.buckets(O.I64, O.R_MAX, t, v, None, 97, 10)
    assert bt.tolist() == [97, 107, 117, 127, 137, 147] and bv.tolist() == [2, 4, 5, 6, 7, 8]
    # selection bitset
    bt, bv = O.buckets(O.I64, O.R_COUNT, t, v, np.array([0b10101010], np.uint8), 100, 10)
    assert bt.tolist() == [100, 110, 130, 150] and bv.tolist() == [1, 1, 1, 1]


@pytest.mark.gpu
def test_gpu_buckets_match_native_bucket_semantics():
    import knoxdb_amd as K
    from knoxdb_amd.types import BlockType, ReducerFunc as R
    rng = np.random.default_rng(61)
    KF = {O.R_SUM: R.Sum, O.R_COUNT: R.Count, O.R_MIN: R.Min, O.R_MAX: R.Max, O.R_FIRST: R.First, O.R_LAST: R.Last}
    with K.Context(0) as ctx:
        for n, step, gaps in ((50000, 1000, False), (300000, 77, True), (1 << 20, 4096, False), (999, 5, True)):
            dt = rng.integers(0, 3, size=n)
            if gaps:
                dt[rng.integers(0, n, size=max(1, n // 500))] += step * 3   # empty windows
            t = (1_700_000_000 + np.cumsum(dt)).astype(np.int64)
            t_from = int(t[0]) - 13
            nb = int((t[-1] - t_from) // step) + 1
            tcol = ctx.upload(t, 4096)
            bits_h = np.packbits(rng.random(n) < 0.6, bitorder="little")
            mask = ctx.bits(n).upload(bits_h)
            for bt_k, o, npdt in ((BlockType.Int64, O.I64, np.int64), (BlockType.Float64, O.F64, np.float64), (BlockType.Int16, O.I16, np.int16),
                                  (BlockType.Uint8, O.U8, np.uint8), (BlockType.Float32, O.F32, np.float32)):
                if np.dtype(npdt).kind == "f":
                    v = (rng.integers(0, 1 << 20, n) / 1024.0).astype(npdt)   # exact-sum grid
                    v[rng.integers(0, n, 5)] = -0.0
                else:
                    info = np.iinfo(npdt)
                    v = rng.integers(info.min, info.max, n, endpoint=True).astype(npdt)
                vcol = ctx.upload(v, 4096)
                for m, mh in ((None, None), (mask, bits_h)):
                    states = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=m)
                    for f in FUNCS:
                        want_t, want_v = O.buckets(o, f, t, v, mh, t_from, step)
                        got = [(t_from + k * step, K.agg_value(st, KF[f])[0]) for k, st in enumerate(states) if st.count > 0]
                        assert [g[0] for g in got] == want_t.tolist(), (n, f)
                        gv = np.array([g[1] for g in got], dtype=npdt)
                        if npdt == np.float32 and f == O.R_SUM:
                            # float32 sums: the reference rounds after every add (r.v += v in float32); the GPU
                            # returns the exactly rounded sum -> compare with the exact checker
                            sel = np.unpackbits(mh, bitorder="little")[:n].astype(bool) if mh is not None else np.ones(n, bool)
                            k_of = (t - t_from) // step
                            exact = [np.float32(O.exact_sum_f64(v[sel & (k_of == (wt - t_from) // step)].astype(np.float64))[0]) for wt in want_t]
                            assert gv.tobytes() == np.array(exact, dtype=np.float32).tobytes(), (n, bt_k, f)
                            continue
                        assert gv.tobytes() == want_v.tobytes(), (n, bt_k, f)
                    # mean: exact per window vs the exact-arithmetic checker
                    if np.dtype(npdt).kind == "f":
                        sel = np.unpackbits(mh, bitorder="little")[:n].astype(bool) if mh is not None else np.ones(n, bool)
                        k_of = (t - t_from) // step
                        for k in (0, nb // 2, nb - 1):
                            rows = sel & (k_of == k)
                            if rows.any():
                                _, exact_mean, _ = O.exact_sum_f64(v[rows].astype(np.float64))
                                assert K.agg_value(states[k], R.Mean)[0] == npdt(exact_mean)
                vcol.free()
            tcol.free(); mask.free()
        # out-of-order timestamps: the reference finds / inserts the window's reducer by time
        # (bucket_native.go:123-141); the device sorts the selected rows by window (stable) and reduces
        for n, step, frac in ((600, 4, 0.5), (200000, 50, 0.02), (1 << 20, 1000, 0.3), (5000, 7, 1.0)):
            base = 1_700_000_000 + np.cumsum(rng.integers(0, 3, size=n))
            t = base.copy()
            k = max(1, int(n * frac))
            where = rng.integers(0, n, size=k)
            t[where] = base[rng.integers(0, n, size=k)]          # rows carrying some other row's timestamp
            t = t.astype(np.int64)
            t_from = int(t.min()) + 5                             # some rows before the first window
            nb = int((t.max() - t_from) // step)                  # and some after the last
            nb = max(nb, 1)
            inside = (t >= t_from) & (t < t_from + nb * step)
            tcol = ctx.upload(t, 4096)
            bits_h = np.packbits(rng.random(n) < 0.7, bitorder="little")
            mask = ctx.bits(n).upload(bits_h)
            for bt_k, o, npdt in ((BlockType.Int64, O.I64, np.int64), (BlockType.Float64, O.F64, np.float64), (BlockType.Uint16, O.U16, np.uint16),
                                  (BlockType.Int8, O.I8, np.int8), (BlockType.Float32, O.F32, np.float32)):
                if np.dtype(npdt).kind == "f":
                    v = (rng.integers(0, 1 << 20, n) / 1024.0).astype(npdt)
                    v[rng.integers(0, n, 5)] = np.nan
                else:
                    info = np.iinfo(npdt)
                    v = rng.integers(info.min, info.max, n, endpoint=True, dtype=npdt)
                vcol = ctx.upload(v, 4096)
                for m, mh in ((None, None), (mask, bits_h)):
                    states = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=m, row_base=1000, moments=True)
                    sel = np.unpackbits(mh, bitorder="little")[:n].astype(bool) if mh is not None else np.ones(n, bool)
                    sel_in = np.packbits(sel & inside, bitorder="little")   # the oracle has no upper window limit: cut there
                    for f in FUNCS:
                        if npdt == np.float32 and f == O.R_SUM:
                            continue
                        if np.dtype(npdt).kind == "f" and f == O.R_SUM:
                            vv = np.where(np.isnan(v), 0, v)               # NaN sums compared separately below
                            if np.isnan(v[sel & inside]).any():
                                continue
                        want_t, want_v = O.buckets(o, f, t, v, sel_in, t_from, step)
                        order = np.argsort(want_t, kind="stable")          # Emit walks the windows in time order
                        got = [(t_from + k * step, K.agg_value(st, KF[f])[0]) for k, st in enumerate(states) if st.count > 0]
                        assert [g[0] for g in got] == want_t[order].tolist(), (n, f)
                        gv = np.array([g[1] for g in got], dtype=npdt)
                        assert gv.tobytes() == want_v[order].tobytes(), (n, bt_k, f)
                    # row ids inside the states are the caller's rows (+ row_base)
                    win = (t - t_from) // step
                    for k in (0, nb // 3, nb - 1):
                        rows = np.flatnonzero(sel & inside & (win == k))
                        if rows.size:
                            assert states[k].first_row == rows[0] + 1000 and states[k].last_row == rows[-1] + 1000
                            assert states[k].count == rows.size
                            if npdt == np.int8 and rows.size > 1:       # the var/std second pass runs on the gathered rows too
                                want_var = np.var(v[rows].astype(np.float64), ddof=1)
                                if want_var < 127:
                                    assert K.agg_value(states[k], R.Var)[0] == np.int8(want_var)
                vcol.free()
            tcol.free(); mask.free()
        # the literal per-window mode needs ordered time
        t = np.array([5, 6, 7, 3, 9, 10] * 100, dtype=np.int64)
        tcol, vcol = ctx.upload(t), ctx.upload(np.arange(600, dtype=np.int64))
        with pytest.raises(K.KnoxError) as e:
            ctx.reduce_buckets_sequential(tcol, vcol, R.Sum, 0, 4, 3)
        assert e.value.code == -5
        tcol.free(); vcol.free()
