"""The reference's own StringPool matcher vectors (pkg/stringx/match_test.go:25-33 TestCompare; tables :232-368, data
:370-470; fixture tests/golden/bytes_cases.json written by tests/golden/make_bytes_golden.py) through the DEVICE's
bytes kernels, the way the reference's driver checks them (match_test.go:48-78): count, exact result bytes, 32 poison
bytes behind the result untouched.

Every case runs (a) as ONE StringPool (`knox_col_upload_bytes`), (b) as one pool PER PACK (`knox_col_bytes_create` /
`_append_pack`, 64-row packs, the last one ragged), (c) with equal rows sharing their bytes, (d) inside a tree with a numeric
leaf that is always true (the multi-pass plan's AND), and (e) under a mask (StringPool.Match*'s mask branch, match.go:21-29:
only mask-selected rows are tested, hits OR-ed into the caller's bits)."""
import numpy as np
import pytest

import helpers
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import FilterMode, ReducerFunc
from oracle import oracle as O

pytestmark = pytest.mark.gpu
R = ReducerFunc
MODE = {"eq": FilterMode.Equal, "ne": FilterMode.NotEqual, "lt": FilterMode.Lt, "le": FilterMode.Le, "gt": FilterMode.Gt,
        "ge": FilterMode.Ge, "bw": FilterMode.Range}
CASES = helpers.bytes_golden_cases()


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def leaf(c):
    a, b = bytes.fromhex(c["match"]), bytes.fromhex(c["match2"])
    return Leaf(0, MODE[c["op"]], a, b) if c["op"] == "bw" else Leaf(0, MODE[c["op"]], a)


def check(ctx, c, col, ids, tag, tree=None):
    n = c["n"]
    bits = ctx.bits(n)
    r = ctx.scan(tree or And(leaf(c)), [col, ids], aggs=[(R.Count, 1)], bits_out=bits, pack_counts=True)
    got = bits.bytes(poison=32)                     # asserts the 32 bytes behind the result are untouched
    assert got.tobytes().hex() == c["result"], tag
    assert r.count == c["count"], tag               # set.Count()
    assert int(r.pack_counts.sum()) == c["count"], tag
    bits.free()


@pytest.mark.parametrize("op", ["eq", "ne", "lt", "le", "gt", "ge", "bw"])
def test_bytes_golden_on_device(ctx, op):
    cases = [c for c in CASES if c["op"] == op]
    assert len(cases) == 15
    for c in cases:
        n = c["n"]
        ids = ctx.upload(np.arange(n, dtype=np.uint64), 64)
        for dedup in (False, True):
            ptr, buf = helpers.bytes_golden_pool(c, dedup=dedup)
            one = ctx.upload_bytes(ptr, buf, 64)
            check(ctx, c, one, ids, "%s/%s single pool dedup=%s" % (op, c["name"], dedup))
            if not dedup:
                # an always-true numeric leaf beside it: the bytes leaf inside MatchAnd's multi-pass plan
                check(ctx, c, one, ids, "%s/%s in a tree" % (op, c["name"]), tree=And(leaf(c), Leaf(1, FilterMode.Ge, 0)))
            one.free()
            if n:
                packs = [helpers.bytes_golden_pool(c, p0, min(n, p0 + 64), dedup=dedup) for p0 in range(0, n, 64)]
                col = ctx.upload_bytes_packs(packs, 64)
                assert col.rows == n
                check(ctx, c, col, ids, "%s/%s pool per pack dedup=%s" % (op, c["name"], dedup))
                col.free()
        ids.free()


def test_bytes_golden_masked(ctx):
    """the mask branch (match.go:21-29 and siblings): bits |= hit for mask-selected rows only.  Expected =
    golden result & mask, on top of whatever the caller's bits held (here: zero, as reader.go:334 hands them in)."""
    rng = np.random.default_rng(4)
    for c in CASES:
        n = c["n"]
        if n == 0:
            continue
        ptr, buf = helpers.bytes_golden_pool(c)
        mask = rng.integers(0, 256, size=O.bitfield_len(n)).astype(np.uint8)
        if n % 8:
            mask[-1] &= 0xFF >> (8 - n % 8)
        want = np.frombuffer(bytes.fromhex(c["result"]), dtype=np.uint8) & mask
        ob, _ = O.bytes_match(O.OPNAME[c["op"]], ptr, buf, bytes.fromhex(c["match"]), mask=mask, val2=bytes.fromhex(c["match2"]))
        assert (ob == want).all()
        # device: the mask is the selection an earlier leaf left -- AND of a bitset leaf and the bytes leaf
        col = ctx.upload_bytes(ptr, buf, 64)
        ids = ctx.upload(np.arange(n, dtype=np.uint64), 64)
        bits, m = ctx.bits(n), ctx.bits(n).upload(mask)
        ctx.scan(And(leaf(c)), [col, ids], bits_out=bits)
        bits.and_(m)
        assert (bits.bytes(poison=32) == want).all(), (c["op"], c["name"])
        for o in (col, ids, bits, m):
            o.free()
