"""GPU parity for the device bitset combinators / popcount / indexes
(internal/bitset/generic/bitset.go:13-435), mirroring generic/bitset_test.go."""
import json
import os

import numpy as np
import pytest

import knoxdb_amd as K
from oracle import oracle as O

pytestmark = pytest.mark.gpu
BITS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bitset_cases.json")))
SIZES = BITS["sizes"] + [128, 129, 1000, 8191, 8192, 8193, 65536 + 17, 1 << 20]


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def _fill(size, val):
    b = np.full(O.bitfield_len(size), val, dtype=np.uint8)
    b[-1] &= 0xFF >> (7 - ((size - 1) & 7))
    return b


def _rand(size, rng, density=0.5):
    return np.packbits(rng.random(size) < density, bitorder="little")


def test_patterns_vs_oracle(ctx):
    rng = np.random.default_rng(1)
    for sz in SIZES:
        pats = [_fill(sz, p) for p in BITS["patterns"]] + [_fill(sz, 0), _rand(sz, rng)]
        for a in pats:
            for b in (pats[0], pats[-1], pats[-2], pats[5]):
                for name, meth in (("and", "and_"), ("andnot", "andnot"), ("or", "or_"), ("xor", "xor")):
                    da, db = ctx.bits(sz).upload(a), ctx.bits(sz).upload(b)
                    getattr(da, meth)(db)
                    assert (da.bytes(poison=8) == O.bits_op(name, a, b, sz)).all(), (sz, name)
                    da.free(); db.free()
                for name in ("and", "or"):
                    da, db = ctx.bits(sz).upload(a), ctx.bits(sz).upload(b)
                    any_, all_ = getattr(da, name + "_flag")(db)
                    w, wany, wall = O.bits_flag_op(name, a, b, sz)
                    assert (da.bytes() == w).all() and any_ == wany and all_ == wall, (sz, name)
                    da.free(); db.free()
            d = ctx.bits(sz).upload(a)
            assert d.count() == O.popcount(a, sz)
            d.neg()
            assert (d.bytes(poison=8) == O.bits_neg(a, sz)).all()
            d.one()
            assert (d.bytes() == O.bits_one(sz)).all() and d.count() == sz
            d.zero()
            assert d.count() == 0
            d.free()


def test_popcount_cases(ctx):
    for c in BITS["pop"]:
        src = np.frombuffer(bytes.fromhex(c["source"]), dtype=np.uint8)
        d = ctx.bits(c["size"]).upload(src)  # garbage above `size` in the last byte must be masked
        assert d.count() == c["count"], c["name"]
        d.free()


def test_indexes_reference_cases(ctx):
    for c in BITS["run"]:
        if c["size"] == 0:
            continue
        buf = np.frombuffer(bytes.fromhex(c["buf"]), dtype=np.uint8)
        d = ctx.bits(c["size"]).upload(buf)
        assert d.indexes().tolist() == c["idx"], c["name"]
        d.free()


def test_indexes_random_and_segments(ctx):
    rng = np.random.default_rng(2)
    for sz, dens in ((70000, 0.5), (1 << 20, 0.01), (1 << 20, 1.0), (300001, 0.2)):
        a = _rand(sz, rng, dens)
        d = ctx.bits(sz).upload(a)
        want = O.indexes(a, sz)
        got = d.indexes()
        assert got.size == want.size and (got == want).all()
        seg = 4096
        counts = d.segment_counts(seg)
        bitsu = np.unpackbits(a, bitorder="little")[:sz]
        want_c = [int(bitsu[i:i + seg].sum()) for i in range(0, sz, seg)]
        assert counts.tolist() == want_c
        d.free()


def test_bool_column_matchers(ctx):
    # internal/bitset/match.go:14-62: bool columns are bitsets, every mode is a copy / negation / fill
    rng = np.random.default_rng(3)
    from knoxdb_amd.types import FilterMode as F
    for sz in (7, 64, 1000, 65536 + 5):
        col_h = _rand(sz, rng, 0.3)
        col = ctx.bits(sz).upload(col_h)
        for mode in (F.Equal, F.NotEqual, F.Lt, F.Le, F.Gt, F.Ge):
            for a in (False, True):
                got = col.match_bool(mode, a)
                assert (got.bytes(poison=8) == O.bool_match(int(mode), a, False, col_h, sz)).all(), (sz, mode, a)
                got.free()
        for a in (False, True):
            for b in (False, True):
                got = col.match_bool(F.Range, a, b)
                assert (got.bytes() == O.bool_match(O.RG, a, b, col_h, sz)).all(), (sz, a, b)
                got.free()
        col.free()
