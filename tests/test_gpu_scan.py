"""GPU parity for the fused scan (filter.Match over resident packs + reducers) through the
C ABI, against the oracle's MatchAnd / MatchOr + sequential reducers."""
import math

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf, Or
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
from oracle import oracle as O
from helpers import oracle_match, rand_col

pytestmark = pytest.mark.gpu

R = ReducerFunc
R2O = {R.Sum: O.R_SUM, R.First: O.R_FIRST, R.Last: O.R_LAST, R.Min: O.R_MIN, R.Max: O.R_MAX, R.Mean: O.R_MEAN, R.Count: O.R_COUNT}


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def _same(a, b):
    return a == b or (isinstance(a, (float, np.floating)) and math.isnan(a) and math.isnan(b))


def _ulp_dist(a, b):
    ia, ib = np.array([a, b], dtype=np.float64).view(np.int64)
    ia = ia if ia >= 0 else np.int64(-(2**63)) - ia
    ib = ib if ib >= 0 else np.int64(-(2**63)) - ib
    return abs(int(ia) - int(ib))


def check_scan(ctx, tree, host_cols, aggs, pack_rows, n):
    dev = [ctx.upload(c, pack_rows) if not isinstance(c, tuple) else ctx.upload_bytes(c[0], c[1], pack_rows) for c in host_cols]
    for d in dev:
        d.pack_rows = pack_rows
    bits = ctx.bits(n)
    res = ctx.scan(tree, dev, aggs=aggs, bits_out=bits, pack_counts=True)
    want = oracle_match(tree if not isinstance(tree, Leaf) else And(tree), host_cols, n)
    got = bits.bytes(poison=32)
    assert (got == want).all(), "bitset mismatch"
    assert res.count == O.popcount(want, n)
    # per-pack counts == popcount of each pack's slice (Bitset.ResetCount semantics)
    for p, c in enumerate(res.pack_counts):
        lo = p * pack_rows
        ln = min(pack_rows, n - lo)
        assert c == O.popcount(want[lo // 8:lo // 8 + (ln + 7) // 8], ln), p
    for i, (f, ci) in enumerate(aggs or []):
        col = host_cols[ci]
        dt = int({np.dtype(v): k for k, v in K.types.NP_DTYPE.items()}[col.dtype])
        gv, gseen, grow = res.value(i)
        if f == R.Mean or (f == R.Sum and col.dtype.kind == "f"):
            xs = col.astype(np.float64)
            if col.dtype.kind == "f":
                selm = np.unpackbits(want, bitorder="little")[:n].astype(bool)
                big = np.abs(xs[selm & np.isfinite(xs)]) > 1e300
                if big.sum() > 1:
                    # partial sums leave the float64 range: the result is order dependent in the
                    # reference too (sequential `r.v += v` overflows to Inf at some prefix)
                    continue
            s, m, cnt = O.exact_sum_f64(xs, want)
            exact = (m if f == R.Mean else s)
            if col.dtype == np.float32 or col.dtype.kind != "f":
                exact = col.dtype.type(exact) if not (col.dtype.kind != "f" and math.isnan(exact)) else exact
            assert gseen == (cnt > 0)
            if cnt:
                if col.dtype.kind == "f":
                    assert _same(gv, exact) or _ulp_dist(float(gv), float(exact)) <= 1, (f, gv, exact)
                else:
                    assert int(gv) == int(exact), (f, gv, exact)  # T(mean) truncation of the exact mean
            continue
        wv, wseen, wrow = O.reduce(dt, R2O[f], col, want)
        assert gseen == wseen, (f, gseen, wseen)
        assert _same(gv, wv), (f, gv, wv)
        if f in (R.Min, R.Max, R.First, R.Last) and wseen:
            assert grow == wrow, (f, grow, wrow)
    for d in dev:
        d.free()
    bits.free()
    return res


SIZES = [100, 2048, 2049, 4096 + 77, 3 * 65536 + 11]


@pytest.mark.parametrize("n", SIZES)
def test_int64_between_sum(ctx, n):
    rng = np.random.default_rng(n)
    x = O.generate(O.I64, 0, 2, 0, n, lo=-(1 << 40), span=1 << 41)
    tree = And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38))
    check_scan(ctx, tree, [x], [(R.Sum, 0), (R.Count, 0), (R.Min, 0), (R.Max, 0), (R.First, 0), (R.Last, 0), (R.Mean, 0)], 1024, n)


@pytest.mark.parametrize("bt", [BlockType.Int64, BlockType.Uint64, BlockType.Float64, BlockType.Int32, BlockType.Uint32,
                                BlockType.Float32, BlockType.Int16, BlockType.Uint16, BlockType.Int8, BlockType.Uint8])
def test_single_leaf_all_types_all_aggs(ctx, bt):
    rng = np.random.default_rng(int(bt))
    for n in (777, 16384 + 100, 70000):
        x = rand_col(bt, n, rng)
        for op in (FilterMode.Le, FilterMode.NotEqual, FilterMode.Range):
            a, b = x[3], x[5]
            if op == FilterMode.Range and x.dtype.kind == "f" and not (a <= b):
                a, b = b, a
            tree = And(Leaf(0, op, a, b))
            aggs = [(R.Sum, 0), (R.Count, 0), (R.Min, 0), (R.Max, 0), (R.First, 0), (R.Last, 0)]
            if x.dtype.kind != "f" or True:
                aggs.append((R.Mean, 0))
            if x.dtype.kind == "f":
                # sums containing Inf/NaN follow IEEE: keep them, exact checker handles specials
                pass
            check_scan(ctx, tree, [x], aggs, 256, n)


def test_two_float_predicates_anded_min_max_mean(ctx):
    # BASELINE config 3 shape: A >= 256 AND B < 0.5, min/max/mean(A) + count
    n = 5 * 65536 + 333
    a = O.generate(O.F64, 1, 3, 0, n)
    b = O.generate(O.F64, 2, 4, 0, n)
    tree = And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5))
    res = check_scan(ctx, tree, [a, b], [(R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.Count, 0), (R.Sum, 0)], 65536, n)
    # exact-sum dataset: the float sum must be bit-identical to the literal sequential sum
    want = oracle_match(tree, [a, b], n)
    assert res.value(4)[0] == O.reduce(O.F64, O.R_SUM, a, want)[0]


def test_semantic_float_set_nan_inf_zero(ctx):
    n = 40000
    a = O.generate(O.F64, 4, 7, 0, n)
    a[0] = np.nan  # first selected value NaN -> sticky for min/max
    tree = And(Leaf(0, FilterMode.NotEqual, 1.0))
    check_scan(ctx, tree, [a], [(R.Min, 0), (R.Max, 0), (R.First, 0), (R.Last, 0), (R.Count, 0)], 4096, n)
    a[0] = 3.0
    check_scan(ctx, tree, [a], [(R.Min, 0), (R.Max, 0)], 4096, n)
    z = np.array([0.0, -0.0, 5.0, -0.0, 0.0] * 1000)
    check_scan(ctx, And(Leaf(0, FilterMode.Le, 0.0)), [z], [(R.Min, 0), (R.Max, 0), (R.Last, 0)], 512, z.size)
    z = -z
    check_scan(ctx, And(Leaf(0, FilterMode.Ge, 0.0)), [z], [(R.Min, 0), (R.Max, 0)], 512, z.size)


def test_trees_fused_up_to_four_leaves(ctx):
    rng = np.random.default_rng(21)
    n = 2 * 65536 + 77
    cols = [rand_col(BlockType.Int64, n, rng), rand_col(BlockType.Uint64, n, rng), rand_col(BlockType.Float64, n, rng),
            rand_col(BlockType.Int64, n, rng)]
    L = Leaf
    trees = [
        And(L(0, FilterMode.Gt, 0), L(1, FilterMode.Lt, 30)),
        And(Or(L(0, FilterMode.Equal, 5), L(1, FilterMode.Equal, 5))),
        And(L(0, FilterMode.Ge, -10), Or(L(1, FilterMode.Le, 10), L(2, FilterMode.Gt, 50.0)), L(3, FilterMode.NotEqual, 0)),
        And(Or(And(L(0, FilterMode.Lt, 0), L(2, FilterMode.Range, -100.0, 100.0)), L(3, FilterMode.Range, -5, 5))),
        And(L(2, FilterMode.NotEqual, float("nan")), L(2, FilterMode.Le, float("inf"))),
        And(L(0, FilterMode.True_, 0), L(1, FilterMode.Ge, 0)),
        And(L(0, FilterMode.False_, 0)),
        And(Or()),  # empty OR = no match; empty AND = all
        And(),
    ]
    for t in trees:
        check_scan(ctx, t, cols, [(R.Sum, 0), (R.Max, 3), (R.Count, 1)], 16384, n)


def test_trees_multipass_many_leaves_and_mixed_widths(ctx):
    rng = np.random.default_rng(22)
    n = 65536 + 4097
    cols = [rand_col(BlockType.Int64, n, rng), rand_col(BlockType.Int32, n, rng), rand_col(BlockType.Uint8, n, rng),
            rand_col(BlockType.Float32, n, rng), rand_col(BlockType.Int16, n, rng)]
    L = Leaf
    t1 = And(L(0, FilterMode.Gt, -20), L(1, FilterMode.Lt, 30), L(2, FilterMode.NotEqual, 7))
    t2 = And(Or(L(0, FilterMode.Equal, 5), L(0, FilterMode.Equal, 6), L(0, FilterMode.Equal, 7), L(0, FilterMode.Equal, 8),
                L(0, FilterMode.Equal, 9), L(0, FilterMode.Lt, -40)), L(0, FilterMode.NotEqual, 9))
    t3 = And(Or(L(3, FilterMode.Gt, 10.0), And(L(4, FilterMode.Range, -10, 10), L(1, FilterMode.Ge, 0))), L(2, FilterMode.Le, 200))
    for t in (t1, t2, t3):
        check_scan(ctx, t, cols, [(R.Sum, 1), (R.Min, 3), (R.Max, 4), (R.Count, 0), (R.Mean, 2)], 4096, n)


def test_bytes_column_eq_ne_and_mixed_tree(ctx):
    rng = np.random.default_rng(23)
    n = 20000 + 13
    keys = [bytes(rng.integers(0, 256, size=32, dtype=np.uint8)) for _ in range(50)] + [b"", b"short", b"x" * 100]
    pick = rng.integers(0, len(keys), size=n)
    vals = [keys[i] for i in pick]
    buf = np.frombuffer(b"".join(vals), dtype=np.uint8)
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ptr = (offs << np.uint64(32)) | lens
    ids = np.arange(n, dtype=np.uint64)
    for key in (keys[0], keys[7], b"", b"short", b"absent-key"):
        for op in (FilterMode.Equal, FilterMode.NotEqual):
            check_scan(ctx, And(Leaf(0, op, key)), [(ptr, buf), ids], [(R.Count, 1), (R.Sum, 1)], 1024, n)
    check_scan(ctx, And(Leaf(0, FilterMode.Equal, keys[3]), Leaf(1, FilterMode.Lt, 10000)), [(ptr, buf), ids], [(R.Max, 1)], 1024, n)


def test_bytes_column_ordered_modes(ctx):
    """StringPool.MatchLess / LessEqual / Greater / GreaterEqual / Between (pkg/stringx/match.go:86-251):
    bytes.Compare order -- byte-lexicographic, a proper prefix sorts first, empty string lowest."""
    rng = np.random.default_rng(29)
    n = 9000 + 5
    keys = [b"", b"a", b"ab", b"abc", b"abd", b"b", b"\x00", b"\x00\x00", b"\xff", b"\xff\x00", b"abcdefgh", b"abcdefgi", b"abcdefghi",
            b"abcdefghabcdefgh", b"abcdefghabcdefgi", b"abcdefgh\x80", b"abcdefgh\x7f"]
    keys += [bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8)) for _ in range(60)]
    pick = rng.integers(0, len(keys), size=n)
    vals = [keys[i] for i in pick]
    buf = np.frombuffer(b"".join(vals) + b"\x00" * 8, dtype=np.uint8)
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ptr = (offs << np.uint64(32)) | lens
    ids = np.arange(n, dtype=np.uint64)
    # python bytes order == bytes.Compare order: cross-check the oracle once, independently
    want_lt = np.packbits(np.array([v < b"abcdefgh" for v in vals]), bitorder="little")
    assert (O.bytes_match(O.LT, ptr, buf, b"abcdefgh")[0] == want_lt).all()
    for key in (b"", b"abc", b"abcdefgh", b"abcdefgh\x80", b"\xff", keys[20], b"zzzz-absent"):
        for op in (FilterMode.Lt, FilterMode.Le, FilterMode.Gt, FilterMode.Ge):
            check_scan(ctx, And(Leaf(0, op, key)), [(ptr, buf), ids], [(R.Count, 1)], 1024, n)
    for lo, hi in ((b"a", b"abd"), (b"", b"\x00\x00"), (b"abcdefgh", b"abcdefghi"), (b"b", b"a"), (keys[25], keys[25])):
        check_scan(ctx, And(Leaf(0, FilterMode.Range, lo, hi)), [(ptr, buf), ids], [(R.Sum, 1)], 1024, n)
    # inside a mixed tree (multi-pass plan)
    check_scan(ctx, Or(Leaf(0, FilterMode.Lt, b"abc"), And(Leaf(0, FilterMode.Ge, b"\xff"), Leaf(1, FilterMode.Lt, 5000))),
               [(ptr, buf), ids], [(R.Max, 1)], 1024, n)


def test_bytes_column_in_not_in(ctx):
    """bytesInSetMatcher / bytesNotInSetMatcher (match_bytes.go:300-540): equality membership, incl. the
    empty string, duplicates in the set, an empty set, values that share long prefixes"""
    rng = np.random.default_rng(31)
    n = 7000 + 3
    keys = [b"", b"a", b"ab", b"abcdefgh", b"abcdefghi", b"abcdefgh\x00", b"\xff" * 9] + \
           [bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8)) for _ in range(80)]
    vals = [keys[i] for i in rng.integers(0, len(keys), size=n)]
    buf = np.frombuffer(b"".join(vals) + b"\x00" * 8, dtype=np.uint8)
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ptr = (offs << np.uint64(32)) | lens
    ids = np.arange(n, dtype=np.uint64)
    sets = [[keys[3], keys[10], b"absent", keys[3], b""], [b"ab"], [], keys[:50], [b"zz-absent-1", b"zz-absent-2"]]
    for st in sets:
        for mode in (FilterMode.In, FilterMode.NotIn):
            check_scan(ctx, And(Leaf(0, mode, st)), [(ptr, buf), ids], [(R.Count, 1)], 1024, n)
    # independent check of the oracle against Python set membership
    want = np.packbits(np.array([v in set(sets[0]) for v in vals]), bitorder="little")
    assert (O.bytes_inset(False, ptr, buf, sets[0]) == want).all()
    # inside a mixed tree
    check_scan(ctx, And(Leaf(0, FilterMode.In, keys[:30]), Leaf(1, FilterMode.Ge, 1000)), [(ptr, buf), ids], [(R.Sum, 1)], 1024, n)


@pytest.mark.parametrize("bt", [BlockType.Int64, BlockType.Float64, BlockType.Int32, BlockType.Uint16, BlockType.Int8])
def test_pack_counts_from_the_scan_kernel(ctx, bt):
    """whole packs, no ragged tail: the per-pack counts come from the scan kernel's per-wave-tile
    popcounts (no second pass over the bitset) -- must equal the popcount of every pack's slice,
    also with pruned packs (compact live-pack walk) and for a multi-leaf tree"""
    rng = np.random.default_rng(300 + int(bt))
    pack, n_packs = 16384, 24
    n = pack * n_packs
    x = rand_col(bt, n, rng)
    y = rand_col(bt, n, rng)
    cx, cy = ctx.upload(x, pack), ctx.upload(y, pack)
    skip = np.zeros(n_packs, dtype=np.uint8)
    for tree, cols, hcols in ((And(Leaf(0, FilterMode.Ge, 3)), [cx], [x]),
                              (Or(Leaf(0, FilterMode.Lt, 2), And(Leaf(1, FilterMode.Ge, 1), Leaf(0, FilterMode.NotEqual, 5))), [cx, cy], [x, y])):
        want = oracle_match(tree, hcols, n)
        per_pack = [O.popcount(want[p * pack // 8:(p + 1) * pack // 8], pack) for p in range(n_packs)]
        bits = ctx.bits(n)
        res = ctx.scan(tree, cols, aggs=[(R.Count, 0)], bits_out=bits, pack_counts=True)
        assert (bits.bytes() == want).all()
        assert res.pack_counts.tolist() == per_pack and res.count == sum(per_pack)
        # prune the packs that hold no match (what a zone map / bloom may do) plus nothing else
        skip[:] = [1 if c == 0 else 0 for c in per_pack]
        skip[5] = 1 if per_pack[5] == 0 else skip[5]
        res2 = ctx.scan(tree, cols, aggs=[(R.Count, 0)], bits_out=bits, pack_counts=True, pack_skip=skip)
        assert (bits.bytes() == want).all() and res2.pack_counts.tolist() == per_pack
        bits.free()
    # all-zero packs in the data so that pruning really skips something
    z = x.copy()
    z[3 * pack:7 * pack] = 0
    cz = ctx.upload(z, pack)
    tree = And(Leaf(0, FilterMode.Gt, 0) if z.dtype.kind != "f" else Leaf(0, FilterMode.Gt, 0.0))
    want = oracle_match(tree, [z], n)
    per_pack = [O.popcount(want[p * pack // 8:(p + 1) * pack // 8], pack) for p in range(n_packs)]
    assert per_pack[3:7] == [0, 0, 0, 0]
    skip[:] = 0
    skip[3:7] = 1
    bits = ctx.bits(n)
    bits.one()  # stale ones in the pruned range must be cleared by the scan
    res3 = ctx.scan(tree, [cz], aggs=[(R.Count, 0)], bits_out=bits, pack_counts=True, pack_skip=skip)
    assert (bits.bytes() == want).all() and res3.pack_counts.tolist() == per_pack and res3.rows_scanned == n - 4 * pack
    for o in (cx, cy, cz, bits):
        o.free()


def test_pack_skip_is_an_optimisation_only(ctx):
    # packs pruned by a bloom / zone map (no false negatives) produce the same bitset, fewer rows read
    n, pack = 8 * 4096, 4096
    x = np.arange(n, dtype=np.int64)
    col = ctx.upload(x, pack)
    col.pack_rows = pack
    bits = ctx.bits(n)
    tree = And(Leaf(0, FilterMode.Range, 4096 * 3 + 5, 4096 * 3 + 900))
    skip = np.ones(8, dtype=np.uint8)
    skip[3] = 0
    res = ctx.scan(tree, [col], aggs=[(R.Sum, 0)], bits_out=bits, pack_counts=True, pack_skip=skip)
    want, cnt = O.cmp(O.I64, O.RG, x, 4096 * 3 + 5, 4096 * 3 + 900)
    assert (bits.bytes() == want).all() and res.count == cnt == 896
    assert res.rows_scanned == 4096 and res.pack_counts.tolist() == [0, 0, 0, 896, 0, 0, 0, 0]
    assert res.value(0)[0] == int(x[(x >= 4096 * 3 + 5) & (x <= 4096 * 3 + 900)].sum())
    col.free()
    bits.free()


def test_row_base_offsets_row_ids(ctx):
    x = np.array([5, 1, 9, 1, 9], dtype=np.int64).repeat(1000)
    col = ctx.upload(x, 1024)
    res = ctx.scan(And(Leaf(0, FilterMode.Ge, 0)), [col], aggs=[(R.Min, 0), (R.Max, 0), (R.Last, 0)], row_base=1 << 33)
    assert res.value(0)[2] == (1 << 33) + 1000 and res.value(1)[2] == (1 << 33) + 2000 and res.value(2)[2] == (1 << 33) + 4999
    col.free()


def test_reduce_sequential_is_literal_reference_fold(ctx):
    # slow single-lane mode reproduces reducer.go bit for bit (incl. Welford mean / var / std)
    rng = np.random.default_rng(31)
    x = (rng.standard_normal(5000) * 1e3).astype(np.float64)
    bits_h = np.packbits(rng.random(5000) < 0.5, bitorder="little")
    col = ctx.upload(x)
    m = ctx.bits(5000).upload(bits_h)
    for f in range(9):
        got = ctx.reduce_sequential(col, f, m)
        want = O.reduce(O.F64, f, x, bits_h)
        assert _same(got[0], want[0]) and got[1] == want[1] and got[2] == want[2], f
    xi = rng.integers(-1000, 1000, 3000).astype(np.int32)
    ci = ctx.upload(xi)
    for f in range(9):
        if f in (6, 7):
            continue  # T(NaN) for ints is implementation defined
        got, want = ctx.reduce_sequential(ci, f), O.reduce(O.I32, f, xi)
        assert got == want, f
    col.free(); ci.free(); m.free()


def test_masked_reduce_matches_oracle(ctx):
    rng = np.random.default_rng(32)
    n = 3 * 16384 + 5
    for bt in (BlockType.Int64, BlockType.Float64, BlockType.Int16, BlockType.Uint8, BlockType.Float32):
        x = rand_col(bt, n, rng)
        if x.dtype.kind == "f":
            # keep partial sums inside the float range: overflowing sums are order dependent
            x[np.isinf(x) | np.isnan(x) | (np.abs(x) > 1e30)] = 1.0
        bits_h = np.packbits(rng.random(n) < 0.4, bitorder="little")
        col = ctx.upload(x)
        m = ctx.bits(n).upload(bits_h)
        st = ctx.reduce(col, m)
        dt = int(bt)
        for fk, fo in ((R.Count, O.R_COUNT), (R.Min, O.R_MIN), (R.Max, O.R_MAX), (R.First, O.R_FIRST), (R.Last, O.R_LAST)):
            got, want = K.agg_value(st, fk), O.reduce(dt, fo, x, bits_h)
            assert _same(got[0], want[0]) and got[1] == want[1], (bt, fk, got, want)
        if x.dtype.kind != "f":
            assert K.agg_value(st, R.Sum)[0] == O.reduce(dt, O.R_SUM, x, bits_h)[0]
        else:
            s, mean, cnt = O.exact_sum_f64(x.astype(np.float64), bits_h)
            assert _ulp_dist(float(K.agg_value(st, R.Sum)[0]), float(x.dtype.type(s))) <= 1
        st_all = ctx.reduce(col, None)
        assert st_all.count == n
        col.free(); m.free()


def test_in_and_not_in_sets(ctx):
    # numInSetMatcher / numNotInSetMatcher / floatInSetMatcher (match_num.go:593-808, match_float.go)
    rng = np.random.default_rng(41)
    n = 3 * 4096 + 19
    for bt in (BlockType.Int64, BlockType.Int32, BlockType.Uint16, BlockType.Int8, BlockType.Uint64, BlockType.Float64, BlockType.Float32):
        x = rand_col(bt, n, rng)
        sets = [x[rng.integers(0, n, size=5)], x[:1], x[rng.integers(0, n, size=300)], np.zeros(0, dtype=x.dtype)]
        if x.dtype.kind == "f":
            sets.append(np.array([np.nan, 0.0, 5.0], dtype=x.dtype))   # NaN is never a member; 0.0 also matches -0.0
            sets.append(np.array([-0.0, np.inf], dtype=x.dtype))
        else:
            info = np.iinfo(x.dtype)
            sets.append(np.array([info.min, info.max, 0], dtype=x.dtype))
        for s in sets:
            for mode in (FilterMode.In, FilterMode.NotIn):
                check_scan(ctx, And(Leaf(0, mode, s)), [x], [(R.Count, 0), (R.Min, 0)], 1024, n)
    a, b = rand_col(BlockType.Int64, n, rng), rand_col(BlockType.Float64, n, rng)
    tree = And(Leaf(0, FilterMode.In, np.array([5, 6, 7, -3, 0], dtype=np.int64)), Or(Leaf(1, FilterMode.Gt, 0.0), Leaf(0, FilterMode.NotIn, np.array([5], dtype=np.int64))))
    check_scan(ctx, tree, [a, b], [(R.Sum, 0), (R.Max, 1)], 2048, n)


@pytest.mark.parametrize("bt", [BlockType.Int64, BlockType.Uint64, BlockType.Int32, BlockType.Float64, BlockType.Float32, BlockType.Int16, BlockType.Uint8])
def test_in_sets_of_every_lookup_tier(ctx, bt, monkeypatch):
    """the streaming IN / NOT IN kernel (knox_bitset.hip inset_tile_kernel) picks its lookup by the set: a bitmap in LDS for 1- / 2-byte
    columns, <= 8 keys from scalar registers, <= 4096 keys by a branch-free binary search in LDS, larger sets through the L2 -- set sizes
    on both sides of every threshold, members at the type's extremes, a ragged tail behind the last whole tile, and the
    row-per-lane kernel (KNOX_INSET_FORM=rows) on the same input: same bits, both equal to the oracle"""
    rng = np.random.default_rng(4100 + int(bt))
    n = 9 * 8192 + 777
    npdt = {BlockType.Int64: np.int64, BlockType.Uint64: np.uint64, BlockType.Int32: np.int32, BlockType.Float64: np.float64,
            BlockType.Float32: np.float32, BlockType.Int16: np.int16, BlockType.Uint8: np.uint8}[bt]
    if np.dtype(npdt).kind == "f":
        x = (rng.integers(-20000, 20000, size=n) / 4.0).astype(npdt)
        x[rng.integers(0, n, 20)] = np.nan
        x[rng.integers(0, n, 20)] = -0.0
        x[rng.integers(0, n, 5)] = np.inf
        pool = np.unique(x[~np.isnan(x)])
    else:
        info = np.iinfo(npdt)
        span = min(40000, int(info.max) - int(info.min))
        x = rng.integers(int(info.min), int(info.min) + span, size=n, endpoint=True).astype(npdt)
        x[:3] = [info.min, info.max, 0]
        pool = np.unique(x)
    col = ctx.upload(x, 8192)
    for nk in (1, 8, 9, 100, 4096, 4097, 9000):
        members = rng.choice(pool, size=min(nk, pool.size), replace=False)
        if nk >= 9:
            members = np.concatenate([members, members[:3]])                  # duplicates in the set
        if np.dtype(npdt).kind == "f":
            members = np.concatenate([members, np.array([np.nan, 12345.5], dtype=npdt)]).astype(npdt)
        else:
            members = np.concatenate([members, np.array([np.iinfo(npdt).max], dtype=npdt)]).astype(npdt)
        for mode in (FilterMode.In, FilterMode.NotIn):
            want = oracle_match(And(Leaf(0, mode, members)), [x], n)
            got = {}
            for form in ("", "rows"):
                if form:
                    monkeypatch.setenv("KNOX_INSET_FORM", form)
                else:
                    monkeypatch.delenv("KNOX_INSET_FORM", raising=False)
                bits = ctx.bits(n)
                res = ctx.scan(And(Leaf(0, mode, members)), [col], aggs=[(R.Count, 0)], bits_out=bits)
                got[form] = bits.bytes(poison=32)
                assert res.count == O.popcount(want, n), (bt, nk, mode, form)
                bits.free()
            assert (got[""] == want).all() and (got["rows"] == want).all(), (bt, nk, mode)
    monkeypatch.delenv("KNOX_INSET_FORM", raising=False)
    col.free()


def test_bytes_in_sets_prefix_search_and_general_form_agree(ctx, monkeypatch):
    """bytes IN / NOT IN searches the members' 8-byte prefixes in LDS and compares in full only the members that share the row's
    prefix (knox_bytes.hip bytes_inset_prefix_kernel); KNOX_BYTES_IN_FORM=general keeps the full-compare binary search.  Members
    and rows that share prefixes of 8 and more bytes, rows shorter than 8 bytes (their first word is cut to their length), the empty
    string, zero bytes behind a prefix, a set beyond 4096 members (general form either way)"""
    rng = np.random.default_rng(77)
    n = 5 * 1024 + 123
    stems = [b"", b"\x00", b"\x00\x00", b"a", b"ab", b"abcdefg", b"abcdefgh", b"abcdefgh\x00", b"abcdefghi", b"abcdefghij" * 3, b"\xff" * 8, b"\xff" * 9]
    keys = stems + [bytes(rng.integers(0, 4, size=int(rng.integers(0, 20)), dtype=np.uint8)) for _ in range(400)]
    vals = [keys[i] for i in rng.integers(0, len(keys), size=n)]
    buf = np.frombuffer(b"".join(vals) + b"\x00" * 8, dtype=np.uint8)
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ptr = (offs << np.uint64(32)) | lens
    col = ctx.upload_bytes(ptr, buf, 1024)
    big = [bytes(rng.integers(0, 256, size=12, dtype=np.uint8)) for _ in range(5000)] + stems[:6]
    for st in (stems, stems[:1], keys[5:200], [b"abcdefgh"], [b"abcdefghX", b"\x00\x00\x00"], big):
        for mode in (FilterMode.In, FilterMode.NotIn):
            want = oracle_match(And(Leaf(0, mode, st)), [(ptr, buf)], n)
            for form in ("", "general"):
                if form:
                    monkeypatch.setenv("KNOX_BYTES_IN_FORM", form)
                else:
                    monkeypatch.delenv("KNOX_BYTES_IN_FORM", raising=False)
                bits = ctx.bits(n)
                ctx.scan(And(Leaf(0, mode, st)), [col], bits_out=bits)
                assert (bits.bytes(poison=32) == want).all(), (len(st), mode, form)
                bits.free()
    monkeypatch.delenv("KNOX_BYTES_IN_FORM", raising=False)
    col.free()


def test_bytes_ordered_modes_prefix_form_and_general_form_agree(ctx, monkeypatch):
    """LT / LE / GT / GE / RANGE on bytes columns decide a row by its zero-padded 8-byte prefix and compare in full only on a tie
    (knox_bytes.hip bytes_ord_kernel); KNOX_BYTES_ORD_FORM=general keeps the one-group-at-a-time kernel.  Values and rows that share
    prefixes of exactly 8 and more bytes, rows that are proper prefixes of the value (and the other way round), zero bytes behind a
    prefix, the empty string, rows shorter than 8 bytes"""
    rng = np.random.default_rng(78)
    n = 6 * 1024 + 77
    stems = [b"", b"\x00", b"\x00\x00", b"a", b"ab", b"abcdefg", b"abcdefgh", b"abcdefgh\x00", b"abcdefgh\x00\x00", b"abcdefghi", b"abcdefghij" * 3,
             b"\xff" * 7, b"\xff" * 8, b"\xff" * 9]
    keys = stems + [bytes(rng.integers(0, 3, size=int(rng.integers(0, 14)), dtype=np.uint8)) for _ in range(300)]
    vals = [keys[i] for i in rng.integers(0, len(keys), size=n)]
    buf = np.frombuffer(b"".join(vals) + b"\x00" * 8, dtype=np.uint8)
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ptr = (offs << np.uint64(32)) | lens
    col = ctx.upload_bytes(ptr, buf, 1024)
    probes = stems + [b"abcdefgh\x01", b"b", bytes([1, 2, 0, 1])]
    for a in probes:
        for mode in (FilterMode.Lt, FilterMode.Le, FilterMode.Gt, FilterMode.Ge, FilterMode.Range):
            b = probes[(probes.index(a) + 5) % len(probes)]
            leaf = Leaf(0, mode, a, b) if mode == FilterMode.Range else Leaf(0, mode, a)
            want = oracle_match(And(leaf), [(ptr, buf)], n)
            for form in ("", "general"):
                if form:
                    monkeypatch.setenv("KNOX_BYTES_ORD_FORM", form)
                else:
                    monkeypatch.delenv("KNOX_BYTES_ORD_FORM", raising=False)
                bits = ctx.bits(n)
                ctx.scan(And(leaf), [col], bits_out=bits)
                assert (bits.bytes(poison=32) == want).all(), (a, b, mode, form)
                bits.free()
    monkeypatch.delenv("KNOX_BYTES_ORD_FORM", raising=False)
    col.free()
