"""Integer mean / var inputs: the device keeps the EXACT sum of the selected integers as a 128-bit value and hands
it over as a double-double (hi + lo == sum).  Checked against Python big-integer arithmetic on data whose 64-bit
sums overflow many times over, for every integer type, through the streaming scan, the masked reduction, the tail
path (ragged sizes) and the per-window reduction."""
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TYPES = ["Int64", "Uint64", "Int32", "Uint32", "Int16", "Uint16", "Int8", "Uint8"]


def exact(st):
    return int(Fraction(st.fsum_hi) + Fraction(st.fsum_lo))


@pytest.mark.parametrize("name", TYPES)
def test_exact_integer_sums(name):
    import knoxdb_amd as K
    from knoxdb_amd import And, Leaf
    from knoxdb_amd.types import BlockType, FilterMode, NP_DTYPE, ReducerFunc as R
    bt = BlockType[name]
    npdt = NP_DTYPE[bt]
    info = np.iinfo(npdt)
    rng = np.random.default_rng(len(name) * 7 + info.bits)
    with K.Context(0) as ctx:
        for n, kind in ((1 << 21, "extremes"), ((1 << 20) + 12345, "random"), (777, "min"), (100000, "max")):
            if kind == "extremes":
                v = rng.choice(np.array([info.min, info.max, info.min + 1, info.max - 1, 0], dtype=npdt), size=n)
            elif kind == "random":
                v = rng.integers(info.min, info.max, n, endpoint=True, dtype=npdt)
            else:
                v = np.full(n, info.min if kind == "min" else info.max, dtype=npdt)
            col = ctx.upload(v, 4096)
            sel = rng.random(n) < 0.6
            mask = ctx.bits(n).upload(np.packbits(sel, bitorder="little"))
            want_all = int(v.astype(object).sum())
            want_sel = int(v[sel].astype(object).sum())
            # masked reduction
            st = ctx.reduce(col)
            assert exact(st) == want_all and st.count == n, (name, kind)
            st = ctx.reduce(col, mask)
            assert exact(st) == want_sel and st.count == int(sel.sum()), (name, kind)
            # streaming scan (TRUE leaf = every row; mean requested so the exact sum is built)
            res = ctx.scan(And(Leaf(0, FilterMode.True_, 0)), [col], aggs=[(R.Mean, 0), (R.Sum, 0)])
            assert exact(res.states[0]) == want_all, (name, kind)
            got_sum = K.agg_value(res.states[1], R.Sum)[0]
            assert int(got_sum) == int(v.sum(dtype=npdt)), (name, kind)          # SumReducer wraps in T
            mean = K.agg_value(res.states[0], R.Mean)[0]
            want_mean = Fraction(want_all, n)
            if abs(want_mean) < 2 ** 52:
                assert int(mean) == int(want_mean) or abs(Fraction(int(mean)) - want_mean) <= 1, (name, kind, mean, float(want_mean))
            # per-window reduction: 8 equal windows over a synthetic clock
            t = np.arange(n, dtype=np.int64)
            tcol = ctx.upload(t, 4096)
            step = (n + 7) // 8
            states = ctx.reduce_buckets(tcol, col, 0, step, 8, mask=mask)
            for k, s in enumerate(states):
                rows = sel & (t // step == k)
                assert exact(s) == int(v[rows].astype(object).sum()) and s.count == int(rows.sum()), (name, kind, k)
            col.free(); mask.free(); tcol.free()
