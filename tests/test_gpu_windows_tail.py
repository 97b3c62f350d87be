"""The host-facing tails of the window / selection path:

  * knox_reduce_buckets_values: Reducer.Value() per window finalised on the device == knox_agg_value over the states
    knox_reduce_buckets returns (same code, knox_finalize.h), every reducer, every width, ordered and out-of-order time;
  * the hand-written stable radix sort behind out-of-order time (knox_sort.hip): 1, 2, 3 and 4 digit passes, against a
    stable numpy argsort -- First / Last observe the stream order inside a window, which is exactly what stability means;
  * knox_bits_indexes_view == knox_bits_indexes; the async flag / count ring shared with knox_scan_async."""
import math
import os

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

pytestmark = pytest.mark.gpu

FUNCS = [R.Sum, R.First, R.Last, R.Min, R.Max, R.Mean, R.Var, R.Std, R.Count]
NP = {B.Int64: np.int64, B.Float64: np.float64, B.Float32: np.float32, B.Int16: np.int16, B.Uint8: np.uint8, B.Uint32: np.uint32}


def _values(rng, bt, n):
    dt = NP[bt]
    if np.dtype(dt).kind == "f":
        v = (rng.integers(-(1 << 18), 1 << 18, n) / 64.0).astype(dt)
        v[rng.integers(0, n, 7)] = np.nan
        v[rng.integers(0, n, 7)] = -0.0
        return v
    info = np.iinfo(dt)
    return rng.integers(info.min, info.max, n, endpoint=True, dtype=dt)


def test_window_values_equal_the_host_finalised_states():
    rng = np.random.default_rng(88)
    with K.Context(0) as ctx:
        for ordered in (True, False):
            n, step = 150_000, 37
            t = 1_000_000 + np.cumsum(rng.integers(0, 3, n))
            if not ordered:
                t = rng.permutation(t)
            t = t.astype(np.int64)
            t_from = int(t.min()) + 11
            nb = int((t.max() - t_from) // step) - 2         # rows before the first and after the last window
            tcol = ctx.upload(t, 4096)
            bits_h = np.packbits(rng.random(n) < 0.55, bitorder="little")
            mask = ctx.bits(n).upload(bits_h)
            for bt in (B.Int64, B.Float64, B.Float32, B.Int16, B.Uint8):
                v = _values(rng, bt, n)
                vcol = ctx.upload(v, 4096)
                for m in (None, mask):
                    states = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=m, moments=True)
                    view = ctx.reduce_buckets_view(tcol, vcol, t_from, step, nb, mask=m, moments=True)
                    assert bytes(view) == bytes(states), (ordered, bt)        # the page-locked view holds the same states
                    for f in FUNCS:
                        vals, seen = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                        vals, seen = vals.copy(), seen.copy()
                        want = np.zeros(nb, dtype=NP[bt])
                        want_seen = np.zeros(nb, dtype=np.uint8)
                        for k in range(nb):
                            val, sn, _ = K.agg_value(states[k], f)
                            want[k], want_seen[k] = val, sn
                        assert (seen == want_seen).all(), (ordered, bt, f)
                        if np.dtype(NP[bt]).kind == "f":   # NaN payloads are not part of Value()'s contract; signed zeros are
                            assert np.array_equal(vals, want, equal_nan=True) and (np.signbit(vals) == np.signbit(want))[~np.isnan(want)].all(), (ordered, bt, f)
                        else:
                            assert vals.tobytes() == want.tobytes(), (ordered, bt, f)
                vcol.free()
            tcol.free(); mask.free()


@pytest.mark.parametrize("nb,n", [(200, 300_000), (60_000, 1 << 20), (70_000, 1 << 20), ((1 << 24) + 5, 1 << 21), (1, 5000), (256, 70_001), (257, 4097)])
def test_out_of_order_windows_sort_is_stable_for_every_pass_count(nb, n):
    """keys < 2^8: 1 pass, < 2^16: 2, < 2^24: 3, above: 4 digit passes"""
    rng = np.random.default_rng(nb)
    step, t_from = 10, 5000
    key = rng.integers(-2, nb + 2, n)                      # some rows before / after every window
    t = (t_from + key * step + rng.integers(0, step, n)).astype(np.int64)
    sel = rng.random(n) < 0.7
    inside = sel & (key >= 0) & (key < nb)
    with K.Context(0) as ctx:
        tcol = ctx.upload(t, 4096)
        mask = ctx.bits(n).upload(np.packbits(sel, bitorder="little"))
        rows = np.flatnonzero(inside)
        order = rows[np.argsort(key[rows], kind="stable")]    # window order, stream order inside a window
        ks = key[order]
        first_pos = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        last_pos = np.r_[first_pos[1:] - 1, ks.size - 1]
        present = ks[first_pos]
        for bt in (B.Int64, B.Uint32, B.Int16, B.Uint8):
            dt = NP[bt]
            info = np.iinfo(dt)
            v = rng.integers(info.min, info.max, n, endpoint=True, dtype=dt)
            vcol = ctx.upload(v, 4096)
            for direct in ("0", None):        # through the sort, and (few windows) folded where the rows lie: the same oracle
                keep = os.environ.pop("KNOX_WINDOW_DIRECT", None)
                if direct is not None:
                    os.environ["KNOX_WINDOW_DIRECT"] = direct
                got = {}
                try:
                    for f in (R.First, R.Last, R.Count, R.Sum, R.Max):
                        vals, seen = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=mask)
                        got[f] = (vals.copy(), seen.copy())
                        if direct == "0":
                            assert ctx.windows_last_path() in (2, 3), (nb, f)
                        elif nb <= 4096:
                            assert ctx.windows_last_path() == 4, (nb, f)      # (more windows: several ranges, or the sort)
                finally:
                    os.environ.pop("KNOX_WINDOW_DIRECT", None)
                    if keep is not None:
                        os.environ["KNOX_WINDOW_DIRECT"] = keep
                seen = got[R.First][1]
                assert np.flatnonzero(seen).tolist() == present.tolist()
                assert (got[R.First][0][present] == v[order[first_pos]]).all(), bt
                assert (got[R.Last][0][present] == v[order[last_pos]]).all(), bt
                cnt = np.bincount(ks, minlength=nb)[present]
                assert (got[R.Count][0][present] == cnt.astype(dt)).all(), bt          # CountReducer counts in T
                sums = np.zeros(nb, dtype=np.uint64)
                np.add.at(sums, ks, v[order].astype(np.int64).astype(np.uint64))
                udt = np.dtype("u%d" % np.dtype(dt).itemsize)                            # SumReducer wraps in T
                assert (got[R.Sum][0].view(udt)[present] == sums[present].astype(udt)).all(), bt
                mx = np.full(nb, info.min, dtype=np.int64)
                np.maximum.at(mx, ks, v[order].astype(np.int64))
                assert (got[R.Max][0][present].astype(np.int64) == mx[present]).all(), bt
                # windows without rows: Value() is the zero value, seen 0
                assert not got[R.Sum][0][seen == 0].any()
            vcol.free()
        tcol.free(); mask.free()


def test_indexes_view_is_indexes_without_a_copy():
    rng = np.random.default_rng(5)
    with K.Context(0) as ctx:
        for n, p in ((1 << 22, 0.3), (100_003, 0.9), (64, 0.5), (1 << 20, 0.0), (1 << 20, 1.0), (1, 1.0)):
            sel = rng.random(n) < p if 0 < p < 1 else np.full(n, bool(p))
            bits = ctx.bits(n).upload(np.packbits(sel, bitorder="little"))
            view = bits.indexes_view()
            want = np.flatnonzero(sel).astype(np.uint32)
            assert view.dtype == np.uint32 and view.shape == want.shape and (view == want).all()
            assert (bits.indexes() == want).all()
            buf = np.zeros(want.size + 3, dtype=np.uint32)
            assert (bits.indexes(out=buf) == want).all()
            bits.free()


def test_async_flags_share_the_result_ring_with_the_scans():
    rng = np.random.default_rng(9)
    n = 1 << 18
    with K.Context(0) as ctx:
        ha, hb = rng.random(n) < 0.5, rng.random(n) < 0.5
        a = ctx.bits(n).upload(np.packbits(ha, bitorder="little"))
        b = ctx.bits(n).upload(np.packbits(hb, bitorder="little"))
        z = ctx.bits(n)
        col = ctx.upload(np.arange(n, dtype=np.int64), 4096)
        plan = ctx.plan(And(Leaf(0, F.Lt, 1000)), [col])
        a.and_flag_async(b)                  # a &= b           -> any True, all False
        a.count_async()                      #                  -> popcount(a & b)
        ctx.scan_async(plan)                 # a scan in between
        z.or_flag_async(z)                   # 0 | 0            -> any False, all False
        with pytest.raises(K.KnoxError):     # 5th uncollected result: busy
            a.count_async()
        assert ctx.bits_collect() == (True, False, None)
        assert ctx.bits_collect() == (None, None, int((ha & hb).sum()))
        with pytest.raises(K.KnoxError):     # the oldest is the scan's: wrong collector, nothing consumed
            ctx.bits_collect()
        assert ctx.scan_collect(plan).count == 1000
        assert ctx.bits_collect() == (False, False, None)
        with pytest.raises(K.KnoxError):     # nothing in flight
            ctx.bits_collect()
        # the blocking calls keep working and agree
        z.one()
        assert z.or_flag(a) == (True, True)
        assert a.count() == int((ha & hb).sum())
        for o in (a, b, z, col):
            o.free()


def test_out_of_order_windows_with_nothing_selected_or_nothing_inside():
    rng = np.random.default_rng(3)
    n, nb, step, t_from = 100_000, 300, 10, 1000
    t = rng.permutation(t_from + rng.integers(0, nb * step, n)).astype(np.int64)
    v = rng.integers(-100, 100, n).astype(np.int64)
    with K.Context(0) as ctx:
        tcol, vcol = ctx.upload(t, 4096), ctx.upload(v, 4096)
        none = ctx.bits(n)                                     # nothing selected
        vals, seen = ctx.reduce_buckets_values(tcol, vcol, R.Sum, t_from, step, nb, mask=none)
        assert not seen.any() and not vals.any()
        states = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=none)
        assert all(st.count == 0 for st in states)
        # every row outside the windows (all timestamps before t_from + ... shifted far away)
        vals, seen = ctx.reduce_buckets_values(tcol, vcol, R.Count, t_from + 10 * nb * step, step, nb)
        assert not seen.any()
        # and exactly one selected row
        one = ctx.bits(n).upload(np.packbits(np.arange(n) == 777, bitorder="little"))
        vals, seen = ctx.reduce_buckets_values(tcol, vcol, R.First, t_from, step, nb, mask=one)
        k = (int(t[777]) - t_from) // step
        assert seen.sum() == 1 and seen[k] == 1 and vals[k] == v[777]
        for o in (tcol, vcol, none, one):
            o.free()


def test_many_small_windows_three_launch_forms_agree_and_match_the_oracle():
    """The reduce step has three launch forms (knox_windows.cpp reduce_windows): a workgroup per work item, a wave per item, and
    -- for more windows than the chip has wave slots -- a 16-lane group per item, four windows per wave (knox_window.hip
    bucket_reduce_group_kernel).  KNOX_WINDOW_FORM pins one; all three must return the same states, byte for byte, for every
    width, with and without a selection, with empty windows, windows of 1 ... a few thousand rows and ragged ends; the default
    choice (the group form at 40 000 windows) is checked against the oracle's literal NativeBucket.Push
    (internal/reducer/bucket_native.go:104-159)."""
    import os
    from oracle import oracle as O
    rng = np.random.default_rng(404)
    OT = {B.Int64: O.I64, B.Float64: O.F64, B.Int16: O.I16, B.Uint8: O.U8, B.Uint32: O.U32}
    OF = {R.Sum: O.R_SUM, R.Min: O.R_MIN, R.Max: O.R_MAX, R.First: O.R_FIRST, R.Last: O.R_LAST, R.Count: O.R_COUNT}
    old = os.environ.pop("KNOX_WINDOW_FORM", None)
    try:
        with K.Context(0) as ctx:
            for n, mean_rows in ((1 << 20, 26), (700_001, 5), (1 << 21, 900)):
                dt = rng.integers(0, 3, n)
                dt[rng.integers(0, n, n // 300)] += 50                      # gaps: runs of empty windows
                t = (5_000_000 + np.cumsum(dt)).astype(np.int64)
                step = mean_rows
                t_from = int(t[0]) - 3
                nb = int((t[-1] - t_from) // step) + 1
                tcol = ctx.upload(t, 4096)
                bits_h = np.packbits(rng.random(n) < 0.5, bitorder="little")
                mask = ctx.bits(n).upload(bits_h)
                for bt in (B.Int64, B.Float64, B.Float32, B.Int16, B.Uint8, B.Uint32):
                    v = _values(rng, bt, n)
                    if np.dtype(NP[bt]).kind == "f":
                        v = np.where(np.isnan(v), 0.5, v).astype(NP[bt])   # the oracle comparison below wants comparable sums
                    vcol = ctx.upload(v, 4096)
                    for m, mh in ((None, None), (mask, bits_h)):
                        got = {}
                        for form in ("block", "wave", "group", None):
                            if form is None:
                                os.environ.pop("KNOX_WINDOW_FORM", None)
                            else:
                                os.environ["KNOX_WINDOW_FORM"] = form
                            got[form] = bytes(ctx.reduce_buckets_view(tcol, vcol, t_from, step, nb, mask=m))
                            if bt in (B.Int16, B.Float64):
                                # the var / std second pass has a workgroup and a group form: its double-double partial sums meet in
                                # another order, which Value() must not show
                                for f in (R.Var, R.Std):
                                    vals, seen = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                                    got[(form, f)] = (vals.tobytes(), seen.tobytes())
                        # values of a known reducer: the group form then keeps only the sums (knox_reduce_buckets_values, Sum / Count / Mean /
                        # Var / Std) -- the same values as the other forms, which always carry every field
                        for form in ("block", "group"):
                            os.environ["KNOX_WINDOW_FORM"] = form
                            for f in (R.Sum, R.Count, R.Mean, R.Min, R.Last):
                                vals, seen = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                                got[(form, "v", f)] = (vals.tobytes(), seen.tobytes())
                        for f in (R.Sum, R.Count, R.Mean, R.Min, R.Last):
                            assert got[("group", "v", f)] == got[("block", "v", f)], (n, bt, f)
                        assert got["wave"] == got["block"], (n, bt, "wave")
                        assert got["group"] == got["block"], (n, bt, "group")
                        assert got[None] == got["block"], (n, bt, "default")
                        if bt in (B.Int16, B.Float64):
                            for f in (R.Var, R.Std):
                                assert got[("group", f)] == got[("block", f)] == got[(None, f)], (n, bt, f)
                    if mean_rows == 26 and bt in OT:
                        states = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=mask)
                        for f, of in OF.items():
                            want_t, want_v = O.buckets(OT[bt], of, t, v, bits_h, t_from, step)
                            gt = [t_from + k * step for k, st in enumerate(states) if st.count > 0]
                            assert gt == want_t.tolist(), (bt, f)
                            gv = np.array([K.agg_value(st, f)[0] for st in states if st.count > 0], dtype=want_v.dtype)
                            assert gv.tobytes() == want_v.tobytes(), (bt, f)
                    vcol.free()
                tcol.free(); mask.free()
    finally:
        os.environ.pop("KNOX_WINDOW_FORM", None)
        if old is not None:
            os.environ["KNOX_WINDOW_FORM"] = old


def test_out_of_order_windows_behind_a_selection_compaction_forms_agree():
    """Out-of-order time behind a selection bitset: up to a third selected, the selected rows' (window, [row id], value) are
    compacted in row order by a streaming pass before the sort (knox_sort.hip window_compact_kernel: the vector form -- 512 rows per
    round, dense and thinly selected units -- and the general form for ragged ends and columns without the alignment).
    KNOX_WINDOW_COMPACT=0 / 1 pins the path, KNOX_COMPACT_FORM=rows the general form: states (row ids travel) and values (they do
    not) must be byte-identical in every form, and First / Last must observe the stream order."""
    import os
    rng = np.random.default_rng(4242)
    n, nb, step, t_from = 1_300_037, 700, 16, 90_000          # ragged: not a multiple of 512, several units per wave slot
    key = rng.integers(-3, nb + 3, n)
    t = (t_from + key * step + rng.integers(0, step, n)).astype(np.int64)
    dens = np.where(np.arange(n) < n // 2, 0.02, 0.45)        # thinly selected units first (value loads wait for the bits), dense ones behind
    old = {k: os.environ.pop(k, None) for k in ("KNOX_WINDOW_COMPACT", "KNOX_COMPACT_FORM")}
    try:
        with K.Context(0) as ctx:
            tcol = ctx.upload(t, 4096)
            for sel in (rng.random(n) < dens, rng.random(n) < 0.3, rng.random(n) < 0.8):
                mask = ctx.bits(n).upload(np.packbits(sel, bitorder="little"))
                inside = sel & (key >= 0) & (key < nb)
                rows = np.flatnonzero(inside)
                order = rows[np.argsort(key[rows], kind="stable")]
                ks = key[order]
                first_pos = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
                present = ks[first_pos]
                for bt in (B.Int64, B.Uint32, B.Int16, B.Uint8, B.Float64):
                    v = _values(rng, bt, n)
                    col = ctx.upload(v, 4096)      # (columns are 16-byte aligned, knox_col_wrap refuses others: the general form is reached
                    want = None                    #  through the knob and at ragged ends only)
                    for compact, form in ((None, None), ("0", None), ("1", None), ("1", "rows")):
                        for k, val in (("KNOX_WINDOW_COMPACT", compact), ("KNOX_COMPACT_FORM", form)):
                            os.environ.pop(k, None)
                            if val is not None:
                                os.environ[k] = val
                        states = bytes(ctx.reduce_buckets_view(tcol, col, t_from, step, nb, mask=mask, moments=True))
                        got = [states]
                        for f in (R.First, R.Last, R.Sum, R.Min, R.Count):
                            vals, seen = ctx.reduce_buckets_values(tcol, col, f, t_from, step, nb, mask=mask)
                            got.append((vals.tobytes(), seen.tobytes()))
                        if want is None:
                            want = got
                            firsts = np.frombuffer(got[1][0], dtype=NP[bt])
                            assert np.flatnonzero(np.frombuffer(got[1][1], dtype=np.uint8)).tolist() == present.tolist()
                            assert firsts[present].tobytes() == v[order[first_pos]].tobytes(), bt
                        else:
                            assert got == want, (bt, compact, form)
                    col.free()
                mask.free()
            tcol.free()
    finally:
        for k, val in old.items():
            os.environ.pop(k, None)
            if val is not None:
                os.environ[k] = val


def test_out_of_order_windows_order_free_values_without_the_sort():
    """Value() per window of the reducers with an order-free form (count, first, last, min, max; over integers sum and mean) over
    out-of-order time is folded row by row into per-workgroup LDS tables (window_direct_kernel) instead of being sorted first: byte-identical to the sorted path (KNOX_WINDOW_DIRECT=0), which the
    tests above hold to the oracle -- every integer width, values at the ends of their range (the exact sum behind Mean leaves 64
    bits), Count over a float column, no selection / thin / dense selections, window counts around the sizes at which the tables stop
    fitting LDS (Mean over 8-byte values 7987, Sum ~19 Ki, Count ~39 Ki; above them the sort)."""
    import os
    rng = np.random.default_rng(777)
    old = os.environ.pop("KNOX_WINDOW_DIRECT", None)
    try:
        with K.Context(0) as ctx:
            for n, nb, step in ((1_000_003, 700, 9), (70_001, 4096, 3), (300, 1, 50), (200_000, 7987, 2), (200_000, 7988, 2), (511, 33, 1 << 33), (400_000, 19_000, 2),
                                (400_000, 39_000, 2), (400_000, 70_000, 2)):
                t_from = -5_000_000_000
                key = rng.integers(-2, nb + 2, n)
                t = (t_from + key * step + rng.integers(0, step, n)).astype(np.int64)
                tcol = ctx.upload(t, 4096)
                masks = [None] + [ctx.bits(n).upload(np.packbits(rng.random(n) < p, bitorder="little")) for p in (0.04, 0.6)]
                sels = [np.ones(n, dtype=bool)] + [np.unpackbits(m.bytes(), bitorder="little")[:n].astype(bool) for m in masks[1:]]
                for bt in (B.Int64, B.Uint32, B.Int16, B.Uint8, B.Float64, B.Float32):
                    v = _values(rng, bt, n)
                    if bt == B.Int64:
                        v[: n // 2] = np.iinfo(np.int64).max - rng.integers(0, 5, n // 2)      # sums far beyond 64 bits
                    if bt in (B.Float64, B.Float32):      # zeros of both signs and NaNs everywhere: first occurrence, NaN-first rule
                        v[rng.random(n) < 0.3] = 0.0
                        v[rng.random(n) < 0.2] = -0.0
                        v[rng.random(n) < 0.1] = np.nan
                    vcol = ctx.upload(v, 4096)
                    for m, sel in zip(masks, sels):
                        for f in ((R.Count, R.First, R.Last, R.Min, R.Max) if bt in (B.Float64, B.Float32) else (R.Sum, R.Count, R.Mean, R.Min, R.Max, R.First, R.Last)):
                            os.environ["KNOX_WINDOW_DIRECT"] = "0"
                            a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                            want = (a.tobytes(), b.tobytes())
                            os.environ.pop("KNOX_WINDOW_DIRECT")
                            a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                            if nb <= 4096:
                                assert ctx.windows_last_path() == 4          # (no sort)
                            assert (a.tobytes(), b.tobytes()) == want, (n, nb, bt, f, m is not None)
                            if f == R.Count:
                                inside = sel & (key >= 0) & (key < nb)
                                cnt = np.bincount(key[inside], minlength=nb)
                                assert (a.astype(np.float64) == cnt.astype(NP[bt]).astype(np.float64)).all() and (b == (cnt > 0)).all(), (n, nb, bt)
                            if f in (R.First, R.Last):      # the value of the lowest / highest selected row of the window
                                inside = sel & (key >= 0) & (key < nb)
                                rows = np.flatnonzero(inside)
                                pick = np.full(nb, -1, dtype=np.int64)
                                if f == R.First:
                                    pick[key[rows[::-1]]] = rows[::-1]
                                else:
                                    pick[key[rows]] = rows
                                have = pick >= 0
                                assert (b.astype(bool) == have).all() and a[have].tobytes() == v[pick[have]].tobytes(), (n, nb, bt, f)
                    vcol.free()
                for o in [tcol] + masks[1:]:
                    o.free()
    finally:
        os.environ.pop("KNOX_WINDOW_DIRECT", None)
        if old is not None:
            os.environ["KNOX_WINDOW_DIRECT"] = old


def test_out_of_order_windows_whole_states_without_the_sort():
    """knox_reduce_buckets over out-of-order time, integer columns, up to a few thousand windows: every state field folded by LDS
    atomics where the rows lie (window_state_kernel; 8-byte values take a second walk for the rows of min / max) -- byte-identical
    to the sorted path's states (KNOX_WINDOW_DIRECT=0), with ties for min / max (first occurrence wins), values at the ends of their
    range, no / thin / dense selections, a row base, window counts around the LDS limits (3630 / 3993 / 4437) and the empty windows."""
    import os
    rng = np.random.default_rng(909)
    old = os.environ.pop("KNOX_WINDOW_DIRECT", None)
    try:
        with K.Context(0) as ctx:
            for n, nb, step in ((900_001, 700, 9), (300, 1, 50), (150_000, 3630, 2), (150_000, 3631, 2), (150_000, 3993, 2), (150_000, 4437, 2), (150_000, 4438, 2),
                                (2_000, 3000, 1 << 30)):
                t_from = 7_000_000_000
                key = rng.integers(-2, nb + 2, n)
                t = (t_from + key * step + rng.integers(0, step, n)).astype(np.int64)
                tcol = ctx.upload(t, 4096)
                masks = [None] + [ctx.bits(n).upload(np.packbits(rng.random(n) < p, bitorder="little")) for p in (0.04, 0.6)]
                for bt in (B.Int64, B.Uint32, B.Int16, B.Uint8):
                    for narrow in (False, True):
                        v = _values(rng, bt, n)
                        if narrow:
                            v = (v % 5).astype(v.dtype)           # many rows share a window's min / max: the lowest row must win
                        elif bt == B.Int64:
                            v[: n // 2] = np.iinfo(np.int64).min + rng.integers(0, 5, n // 2)
                        vcol = ctx.upload(v, 4096)
                        for m in masks:
                            for base in (0, 1 << 33):
                                os.environ["KNOX_WINDOW_DIRECT"] = "0"
                                want = bytes(ctx.reduce_buckets_view(tcol, vcol, t_from, step, nb, mask=m, row_base=base))
                                os.environ.pop("KNOX_WINDOW_DIRECT")
                                got = bytes(ctx.reduce_buckets_view(tcol, vcol, t_from, step, nb, mask=m, row_base=base))
                                if nb <= 3630:
                                    assert ctx.windows_last_path() == 5      # (the sort-free states ran)
                                assert got == want, (n, nb, bt, narrow, m is not None, base)
                        vcol.free()
                for o in [tcol] + masks[1:]:
                    o.free()
    finally:
        os.environ.pop("KNOX_WINDOW_DIRECT", None)
        if old is not None:
            os.environ["KNOX_WINDOW_DIRECT"] = old


def test_min_max_rows_are_first_occurrences_in_every_launch_form():
    """MinReducer / MaxReducer adopt a value only when it is strictly better (reducer.go:256-296): among equal values the FIRST row
    wins.  Windows that do not start on a tile boundary, few distinct values (the smallest stands in a window's first, unaligned rows
    and again further down): min_row / max_row against numpy for the workgroup, wave and 16-lane forms of the reduce step.  (Until
    late in round 4 the workgroup form walked a window's unaligned head AFTER its aligned middle and reported the later row.)"""
    import os
    rng = np.random.default_rng(5150)
    n, nb, step, t_from = 700_003, 257, 11, 1000
    key = np.sort(rng.integers(-1, nb + 1, n))
    t = (t_from + key * step + np.sort(rng.integers(0, step, n)) * 0).astype(np.int64)
    old = os.environ.pop("KNOX_WINDOW_FORM", None)
    try:
        with K.Context(0) as ctx:
            tcol = ctx.upload(t, 4096)
            mask = ctx.bits(n).upload(np.packbits(rng.random(n) < 0.7, bitorder="little"))
            sel = np.unpackbits(mask.bytes(), bitorder="little")[:n].astype(bool)
            for bt in (B.Int64, B.Uint32, B.Int16, B.Uint8, B.Float64):
                v = (rng.integers(0, 4, n) - 2).astype(NP[bt]) if bt != B.Float64 else (rng.integers(0, 4, n) / 2.0 - 1.0)
                if np.dtype(NP[bt]).kind == "i":
                    v[rng.random(n) < 0.2] = np.iinfo(NP[bt]).min
                vcol = ctx.upload(v.astype(NP[bt]), 4096)
                for m, s in ((None, np.ones(n, dtype=bool)), (mask, sel)):
                    for form in ("block", "wave", "group"):
                        os.environ["KNOX_WINDOW_FORM"] = form
                        st = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=m)
                        for k in range(nb):
                            rows = np.flatnonzero((key == k) & s)
                            if rows.size == 0:
                                assert st[k].count == 0
                                continue
                            vv = v[rows].astype(np.float64)
                            assert st[k].min_row == rows[np.argmin(vv)] and st[k].max_row == rows[np.argmax(vv)], (bt, form, k, m is not None)
                vcol.free()
            tcol.free(); mask.free()
    finally:
        os.environ.pop("KNOX_WINDOW_FORM", None)
        if old is not None:
            os.environ["KNOX_WINDOW_FORM"] = old


def test_out_of_order_windows_float_sums_are_exact_without_the_sort():
    """Sum / Mean of a float column per window over out-of-order time: every finite value enters a 160-bit fixed-point adder per
    window (window_fsum_kernel: LDS adds of 32-bit digits), so the fold is exact whatever the order -- against math.fsum (the
    correctly rounded exact sum) and against the sorted path's double-double sums (KNOX_WINDOW_DIRECT=0), float64 and float32, values
    spread over ~2^70 in magnitude; infinities of both signs, NaNs and negative zeros follow IEEE (inf - inf and NaN -> NaN, a sum of
    nothing but -0 is -0); a column whose exponents span more than the adders hold (107 binades) takes the sort."""
    import math, os
    rng = np.random.default_rng(31337)
    old = os.environ.pop("KNOX_WINDOW_DIRECT", None)
    try:
        with K.Context(0) as ctx:
            for n, nb, step in ((400_003, 300, 7), (50_000, 3500, 2), (50_000, 8193, 2), (50_000, 32769, 2)):
                t_from = 10_000
                key = rng.integers(-1, nb + 1, n)
                t = (t_from + key * step + rng.integers(0, step, n)).astype(np.int64)
                tcol = ctx.upload(t, 4096)
                mask = ctx.bits(n).upload(np.packbits(rng.random(n) < 0.5, bitorder="little"))
                sel = np.unpackbits(mask.bytes(), bitorder="little")[:n].astype(bool)
                for bt in (B.Float64, B.Float32):
                    for kind in ("wide", "specials", "too wide"):
                        v = ((1.0 + rng.random(n)) * rng.choice([-1.0, 1.0], n) * 10.0 ** rng.integers(-6, 15, n)).astype(NP[bt])      # ~2^70 between the smallest and the largest
                        if kind == "specials":
                            v[rng.random(n) < 0.3] = -0.0
                            idx = rng.integers(0, n, 40)
                            v[idx[:10]] = np.inf
                            v[idx[10:20]] = -np.inf
                            v[idx[20:25]] = np.nan
                            v[key == 3] = -0.0                       # a window of nothing but negative zeros
                        if kind == "too wide" and bt == B.Float64:
                            v[rng.integers(0, n, 50)] = 1e-250
                            v[rng.integers(0, n, 50)] = 1e250
                        vcol = ctx.upload(v, 4096)
                        for m, s in ((None, np.ones(n, dtype=bool)), (mask, sel)):
                            for f in (R.Sum, R.Mean):
                                os.environ["KNOX_WINDOW_DIRECT"] = "0"
                                a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                                want, want_seen = a.copy(), b.copy()
                                os.environ.pop("KNOX_WINDOW_DIRECT")
                                a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                                # 6 = the exact adders ran; 2 / 3 = the sort (more windows than LDS holds, exponents wider than the adders)
                                sorts = nb > 32768 or (kind == "too wide" and bt == B.Float64)
                                if sorts or nb <= 3500 or m is None:      # (8193 windows: three ranges of tables without a selection, the sort behind one)
                                    assert ctx.windows_last_path() in ((2, 3) if sorts else (6,)), (nb, kind, bt, ctx.windows_last_path())
                                assert (b == want_seen).all(), (n, nb, bt, kind, f)
                                assert np.array_equal(a, want, equal_nan=True) and (np.signbit(a) == np.signbit(want))[~np.isnan(want)].all(), (n, nb, bt, kind, f)
                            if bt == B.Float64 and kind == "wide" and n > 100_000:
                                a, b = ctx.reduce_buckets_values(tcol, vcol, R.Sum, t_from, step, nb, mask=m)
                                for k in range(nb):
                                    rows = np.flatnonzero((key == k) & s)
                                    assert bool(b[k]) == (rows.size > 0)
                                    if rows.size:
                                        assert a[k] == math.fsum(v[rows].tolist()), (k, m is not None)
                        vcol.free()
                tcol.free(); mask.free()
            # cancellation far beyond a double-double: every window holds pairs (x, -x) of magnitude ~1e15 and a few values near 1e-3 --
            # the exact adders return math.fsum's answer (negative totals and two's-complement carries across all digits included)
            n, nb, step, t_from = 200_000, 500, 5, 0
            half = n // 2
            key = rng.integers(0, nb, half)
            big = rng.standard_normal(half) * 1e15
            small = rng.standard_normal(half) * 1e-3
            v = np.concatenate([big, -big + 0.0]).astype(np.float64)
            v[rng.integers(0, n, n // 10)] = small[: n // 10]
            keys = np.concatenate([key, key])
            perm = rng.permutation(n)
            v, keys = v[perm], keys[perm]
            t = (t_from + keys * step + rng.integers(0, step, n)).astype(np.int64)
            tcol, vcol = ctx.upload(t, 4096), ctx.upload(v, 4096)
            a, b = ctx.reduce_buckets_values(tcol, vcol, R.Sum, t_from, step, nb)
            assert ctx.windows_last_path() == 6
            for k in range(nb):
                rows = np.flatnonzero(keys == k)
                assert a[k] == math.fsum(v[rows].tolist()), k
            tcol.free(); vcol.free()
    finally:
        os.environ.pop("KNOX_WINDOW_DIRECT", None)
        if old is not None:
            os.environ["KNOX_WINDOW_DIRECT"] = old


def test_out_of_order_windows_float_states_without_the_sort():
    """knox_reduce_buckets over out-of-order time and a FLOAT column (window_fstate_kernel): every field against the sorted path's state
    -- rows and bits of min / max with the reference's rules (a NaN is never a candidate, -0 == +0 and the first of them wins, windows
    of nothing but NaNs have no min), first / last, count; the sum's value hi + lo is the same correctly rounded number on both paths and the
    pairs agree to ~2^-95 of it (the sorted path's is a running double-double, this one's the exact sum split once)."""
    rng = np.random.default_rng(2718)
    old = os.environ.pop("KNOX_WINDOW_DIRECT", None)
    try:
        with K.Context(0) as ctx:
            for n, nb, step in ((500_001, 400, 6), (60_000, 2000, 2), (300, 1, 40)):
                t_from = -123_456
                key = rng.integers(-1, nb + 1, n)
                t = (t_from + key * step + rng.integers(0, step, n)).astype(np.int64)
                tcol = ctx.upload(t, 4096)
                mask = ctx.bits(n).upload(np.packbits(rng.random(n) < 0.4, bitorder="little"))
                for bt in (B.Float64, B.Float32):
                    for kind in ("ties and specials", "wide"):
                        if kind == "wide":
                            v = ((1.0 + rng.random(n)) * rng.choice([-1.0, 1.0], n) * 10.0 ** rng.integers(-6, 15, n)).astype(NP[bt])
                        else:
                            v = (rng.integers(-3, 4, n) / 2.0).astype(NP[bt])
                            v[rng.random(n) < 0.2] = -0.0
                            v[rng.random(n) < 0.1] = np.nan
                            idx = rng.integers(0, n, 30)
                            v[idx[:10]] = np.inf
                            v[idx[10:20]] = -np.inf
                            v[key == 2] = np.nan                      # a window of nothing but NaNs
                        vcol = ctx.upload(v, 4096)
                        for m in (None, mask):
                            for base in (0, 1 << 34):
                                os.environ["KNOX_WINDOW_DIRECT"] = "0"
                                want = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=m, row_base=base)
                                assert ctx.windows_last_path() in (2, 3)
                                os.environ.pop("KNOX_WINDOW_DIRECT")
                                got = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=m, row_base=base)
                                assert ctx.windows_last_path() == 5
                                for k in range(nb):
                                    a, b = got[k], want[k]
                                    for f in ("dtype", "flags", "count", "isum", "min_bits", "max_bits", "first_bits", "last_bits", "min_row", "max_row",
                                              "first_row", "last_row"):
                                        assert getattr(a, f) == getattr(b, f), (n, nb, bt, kind, k, f, getattr(a, f), getattr(b, f))
                                    if math.isnan(b.fsum_hi) or math.isinf(b.fsum_hi):
                                        assert (math.isnan(a.fsum_hi) and math.isnan(b.fsum_hi)) or a.fsum_hi == b.fsum_hi, (k, a.fsum_hi, b.fsum_hi)
                                    else:
                                        # (the sorted path's pair is not normalised -- its hi need not be the rounded sum --: the VALUE hi + lo is what
                                        #  both must agree on, and the pairs as a whole to ~2^-95)
                                        sa, sb = a.fsum_hi + a.fsum_lo, b.fsum_hi + b.fsum_lo
                                        assert sa == sb and math.copysign(1.0, sa) == math.copysign(1.0, sb), (bt, kind, k, sa, sb)
                                        assert abs((a.fsum_hi - b.fsum_hi) + (a.fsum_lo - b.fsum_lo)) <= abs(sb) * 2.0 ** -95, (bt, kind, k)   # (a few ulps of the lo words themselves)
                        vcol.free()
                tcol.free(); mask.free()
    finally:
        os.environ.pop("KNOX_WINDOW_DIRECT", None)
        if old is not None:
            os.environ["KNOX_WINDOW_DIRECT"] = old


def test_out_of_order_windows_float_var_std_are_exact_without_the_sort():
    """Var / Std of a float column per window over out-of-order time (window_fvar_kernel): exact fixed-point sums of the values and of
    their squares, n * S2 - S1^2 as an integer -- against the exact rational variance (fractions) rounded once, and against the
    sorted path's centred second pass; a mean 10^12 times the spread (what a naive sum of squares loses) included."""
    from fractions import Fraction
    rng = np.random.default_rng(1618)
    old = os.environ.pop("KNOX_WINDOW_DIRECT", None)
    try:
        with K.Context(0) as ctx:
            for n, nb, step in ((120_001, 200, 6), (30_000, 1500, 2)):
                t_from = 77
                key = rng.integers(-1, nb + 1, n)
                t = (t_from + key * step + rng.integers(0, step, n)).astype(np.int64)
                tcol = ctx.upload(t, 4096)
                mask = ctx.bits(n).upload(np.packbits(rng.random(n) < 0.6, bitorder="little"))
                sel = np.unpackbits(mask.bytes(), bitorder="little")[:n].astype(bool)
                for bt in (B.Float64, B.Float32):
                    for kind in ("wide", "large mean", "specials"):
                        if kind == "wide":
                            v = ((1.0 + rng.random(n)) * rng.choice([-1.0, 1.0], n) * 10.0 ** rng.integers(-4, 9, n)).astype(NP[bt])
                        elif kind == "large mean":
                            v = (1e9 + rng.standard_normal(n) * 1e-3).astype(NP[bt]) if bt == B.Float64 else (1e4 + rng.standard_normal(n)).astype(NP[bt])
                        else:
                            v = (rng.integers(-3, 4, n) / 2.0).astype(NP[bt])
                            v[rng.integers(0, n, 20)] = np.inf
                            v[rng.integers(0, n, 20)] = np.nan
                            v[rng.random(n) < 0.2] = -0.0
                        vcol = ctx.upload(v, 4096)
                        for m, s in ((None, np.ones(n, dtype=bool)), (mask, sel)):
                            for f in (R.Var, R.Std):
                                os.environ["KNOX_WINDOW_DIRECT"] = "0"
                                a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                                want, want_seen = a.copy(), b.copy()
                                assert ctx.windows_last_path() in (2, 3)
                                os.environ.pop("KNOX_WINDOW_DIRECT")
                                a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                                assert ctx.windows_last_path() == 6
                                assert (b == want_seen).all(), (n, nb, bt, kind, f)
                                assert np.array_equal(a, want, equal_nan=True), (n, nb, bt, kind, f, np.flatnonzero(~((a == want) | (np.isnan(a) & np.isnan(want))))[:5])
                            if bt == B.Float64 and kind != "specials" and n > 100_000:
                                a, b = ctx.reduce_buckets_values(tcol, vcol, R.Var, t_from, step, nb, mask=m)
                                for k in range(0, nb, 7):
                                    rows = np.flatnonzero((key == k) & s)
                                    if rows.size < 2:
                                        continue
                                    fr = [Fraction(float(x)) for x in v[rows]]
                                    mean = sum(fr) / len(fr)
                                    exact = sum((x - mean) ** 2 for x in fr) / (len(fr) - 1)
                                    assert a[k] == float(exact), (kind, k, a[k], float(exact))
                        vcol.free()
                # integer columns take the same adders (up to 4 bytes as doubles, 8-byte ones as they are); Value() truncates toward zero in T
                for bt in (B.Int64, B.Uint32, B.Int16, B.Uint8):
                    v = _values(rng, bt, n)
                    if bt == B.Int64:
                        v[rng.integers(0, n, 50)] = np.iinfo(np.int64).min
                        v[: n // 3] = 10 ** 15 + rng.integers(-5, 6, n // 3)      # a mean 10^15 times the spread
                    vcol = ctx.upload(v, 4096)
                    for m in (None, mask):
                        for f in (R.Var, R.Std):
                            os.environ["KNOX_WINDOW_DIRECT"] = "0"
                            a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                            want = (a.tobytes(), b.tobytes())
                            os.environ.pop("KNOX_WINDOW_DIRECT")
                            a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                            assert ctx.windows_last_path() == 6
                            assert (a.tobytes(), b.tobytes()) == want, (n, nb, bt, f, m is not None)
                    vcol.free()
                tcol.free(); mask.free()
    finally:
        os.environ.pop("KNOX_WINDOW_DIRECT", None)
        if old is not None:
            os.environ["KNOX_WINDOW_DIRECT"] = old
