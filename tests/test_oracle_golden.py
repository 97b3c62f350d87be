"""The CPU oracle against every golden vector the reference's own tests hold for the path
(SURVEY.md 8c).  Mirrors internal/cmp/tests/tests.go:24-48 (count, exact bytes, poison tail),
internal/bitset/generic/bitset_test.go, internal/hash/xxh3_test.go."""
import json
import os

import numpy as np
import pytest

import helpers
from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
CMP = json.load(open(os.path.join(G, "cmp_cases.json")))["cases"]
BITS = json.load(open(os.path.join(G, "bitset_cases.json")))


def _src(c):
    dt = O.NAME[c["dtype"]]
    if c["dtype"] == "f64":
        return np.array(c["src"], dtype=np.uint64).view(np.float64), dt
    if c["dtype"] == "f32":
        return np.array(c["src"], dtype=np.uint32).view(np.float32), dt
    return np.array(c["src"], dtype=O.NP[dt]), dt


def _scalar(c, key):
    if c["dtype"] == "f64":
        return np.array([c[key]], dtype=np.uint64).view(np.float64)[0]
    if c["dtype"] == "f32":
        return np.array([c[key]], dtype=np.uint32).view(np.float32)[0]
    return c[key]


@pytest.mark.parametrize("dtype", sorted(O.NAME))
def test_cmp_golden(dtype):
    cases = [c for c in CMP if c["dtype"] == dtype]
    assert len(cases) > 100
    for c in cases:
        src, dt = _src(c)
        bits, cnt = O.cmp(dt, O.OPNAME[c["op"]], src, _scalar(c, "a"), _scalar(c, "b"), poison=32)
        tag = "%s/%s/%s" % (dtype, c["op"], c["name"])
        assert cnt == c["count"], tag
        assert bits.tobytes().hex() == c["result"], tag


def test_popcount_cases():
    for c in BITS["pop"]:
        src = np.frombuffer(bytes.fromhex(c["source"]), dtype=np.uint8)
        assert O.popcount(src, c["size"]) == c["count"], c["name"]


def _fill(size, val):
    b = np.full(O.bitfield_len(size), val, dtype=np.uint8)
    if b.size:
        b[-1] &= 0xFF >> (7 - ((size - 1) & 7))
    return b


def test_popcount_patterns():
    # generic/bitset_test.go:31-47
    for sz in BITS["sizes"]:
        for pt in BITS["patterns"]:
            buf = _fill(sz, pt)
            assert O.popcount(buf, sz) == int(np.unpackbits(buf).sum())


def test_indexes_cases():
    for c in BITS["run"]:
        buf = np.frombuffer(bytes.fromhex(c["buf"]), dtype=np.uint8)
        idx = O.indexes(buf, c["size"])
        assert idx.tolist() == c["idx"], c["name"]


@pytest.mark.parametrize("op", ["and", "andnot", "or", "xor"])
def test_combinators_vs_self_zeros_ones(op):
    # generic/bitset_test.go:49-330: dst op src for src in {same, zeros, ones}
    for sz in BITS["sizes"]:
        zeros, ones = _fill(sz, 0), _fill(sz, 0xFF)
        for pt in BITS["patterns"]:
            src = _fill(sz, pt)
            exp = {
                "and": (src, zeros, src), "andnot": (zeros, src, zeros),
                "or": (src, src, ones), "xor": (zeros, src, O.bits_neg(src, sz)),
            }[op]
            for other, want in zip((src, zeros, ones), exp):
                got = O.bits_op(op, src, other, sz)
                assert (got == want).all(), (op, sz, pt)


def test_and_flag_or_flag():
    # generic/bitset_test.go:101-211, :331-440: (any, all) over the valid bits
    for sz in BITS["sizes"]:
        zeros, ones = _fill(sz, 0), _fill(sz, 0xFF)
        for pt in BITS["patterns"]:
            src = _fill(sz, pt)
            pc = int(np.unpackbits(src).sum())
            d, any_, all_ = O.bits_flag_op("and", src, src, sz)
            assert (d == src).all() and any_ == (pc > 0) and all_ == (pc == sz)
            d, any_, all_ = O.bits_flag_op("and", src, zeros, sz)
            assert (d == zeros).all() and not any_ and not all_
            d, any_, all_ = O.bits_flag_op("and", src, ones, sz)
            assert (d == src).all() and any_ == (pc > 0) and all_ == (pc == sz)
            d, any_, all_ = O.bits_flag_op("or", src, zeros, sz)
            assert (d == src).all() and any_ == (pc > 0) and all_ == (pc == sz)
            d, any_, all_ = O.bits_flag_op("or", src, ones, sz)
            assert (d == ones).all() and any_ and all_


def test_neg():
    for sz in BITS["sizes"]:
        for pt in BITS["patterns"]:
            src = _fill(sz, pt)
            neg = O.bits_neg(src, sz)
            assert int(np.unpackbits(neg).sum()) == sz - int(np.unpackbits(src).sum())
            assert (O.bits_neg(neg, sz) == src).all()


def test_xxh3_fixed_width_kat():
    kat = json.load(open(os.path.join(G, "xxh3_kat.json")))
    for i in range(8):
        data = bytes(range(i, i + 8))
        assert O.lib().ko_xxh3_u64(int.from_bytes(data, "little")) == kat["u64"][i]
        assert O.lib().ko_xxh3_u32(int.from_bytes(data[:4], "little")) == kat["u32"][i]
        # the []byte hash of the same 8 / 4 bytes is the same function (hash.go:45-53 vs :67-92)
        assert O.xxh3_bytes(data) == kat["u64"][i]
        assert O.xxh3_bytes(data[:4]) == kat["u32"][i]


def test_xxh3_small_widths_match_bytes_hash():
    # xxh3_u16 / xxh3_u8 are the 1-3 byte XXH3 path specialised (xxh3.go:48-58)
    for v in (0, 1, 0x7F, 0x80, 0xFF):
        assert O.lib().ko_xxh3_u8(v) == O.xxh3_bytes(bytes([v]))
    for v in (0, 1, 0x1234, 0xFFFF, 0x8000):
        assert O.lib().ko_xxh3_u16(v) == O.xxh3_bytes(v.to_bytes(2, "little"))


def test_xxh3_bytes_golden():
    g = json.load(open(os.path.join(G, "xxh3_bytes.json")))
    for n, want in g["digests"].items():
        n = int(n)
        data = bytes((i * 131 + 7) & 0xFF for i in range(n))
        assert O.xxh3_bytes(data) == want, n


def test_bloom_no_false_negatives():
    # internal/filter/bloom/bloom_test.go:18-150 (property: added => contained; fp rate sane)
    rng = np.random.default_rng(7)
    keys = rng.integers(0, 2**63, size=4096, dtype=np.uint64)
    f = O.bloom_new(4096, 2)
    assert f.size == 1 + 8192 and f[0] == 4
    hs = [O.lib().ko_xxh3_u64(int(k)) for k in keys]
    for h in hs:
        O.bloom_add(f, h)
    assert all(O.bloom_contains(f, h) for h in hs)
    other = rng.integers(2**63, 2**64 - 1, size=20000, dtype=np.uint64)
    fp = sum(O.bloom_contains(f, O.lib().ko_xxh3_u64(int(k))) for k in other) / other.size
    assert fp < 10 * 0.002394  # design p for factor 2, pack/stats/filter.go:303-311


WIDE = json.load(open(os.path.join(G, "cmp_wide_cases.json")))["cases"]


def wide_limbs(c):
    """stride layout (num.Int128Stride / Int256Stride): one uint64 array per limb, X0 first"""
    nl = 2 if c["dtype"] == "i128" else 4
    src = np.array(c["src"], dtype=np.uint64).reshape(-1, nl)
    return [np.ascontiguousarray(src[:, l]) for l in range(nl)]


@pytest.mark.parametrize("dtype", ["i128", "i256"])
def test_cmp_wide_golden(dtype):
    """internal/cmp/tests/int128.go, int256.go: count, exact bytes, poison tail"""
    cases = [c for c in WIDE if c["dtype"] == dtype]
    assert len(cases) == 119
    for c in cases:
        bits, cnt = O.cmp_wide(O.OPNAME[c["op"]], wide_limbs(c), c["a"], c["b"])
        assert bits.tobytes().hex() == c["result"], (dtype, c["op"], c["name"])
        assert cnt == c["count"], (dtype, c["op"], c["name"])


def test_cmp_wide_mask_semantics():
    """mask restatement: bits = hits & mask everywhere; the unrolled i128 loops count the
    UNMASKED byte (int128.go:40-41), the per-row loops count masked hits only"""
    rng = np.random.default_rng(5)
    n = 77
    for nl in (2, 4):
        limbs = [rng.integers(0, 4, size=n).astype(np.uint64) for _ in range(nl)]
        a = [1] * nl
        mask = rng.integers(0, 256, size=O.bitfield_len(n)).astype(np.uint8)
        mask[3] = 0
        full, full_cnt = O.cmp_wide(O.OPNAME["ge"], limbs, a)
        got, cnt = O.cmp_wide(O.OPNAME["ge"], limbs, a, mask=mask)
        want = full & mask
        want[-1] &= 0xFF >> (8 - n % 8)
        assert (got == want).all()
        if nl == 4:
            assert cnt == O.popcount(want, n)
        else:
            nb = n // 8
            unmasked_full_bytes = sum(bin(int(full[i])).count("1") for i in range(nb) if mask[i])
            assert cnt == unmasked_full_bytes + bin(int(want[nb])).count("1")


def test_bytes_ordered_modes_vs_python_bytes_order():
    """StringPool.MatchLess.. / MatchBetween (pkg/stringx/match.go:86-251) use bytes.Compare, whose
    order is Python's bytes order: an independent check of the restatement, mask path included."""
    rng = np.random.default_rng(3)
    keys = [b"", b"a", b"ab", b"abc", b"b", b"\x00", b"\xff", b"abcdefgh", b"abcdefghi", b"abcdefgh\x80"]
    keys += [bytes(rng.integers(0, 256, size=int(rng.integers(0, 20)), dtype=np.uint8)) for _ in range(40)]
    n = 1003
    vals = [keys[i] for i in rng.integers(0, len(keys), size=n)]
    buf = np.frombuffer(b"".join(vals) + b"\x00", dtype=np.uint8)
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ptr = (offs << np.uint64(32)) | lens
    mask = rng.integers(0, 256, size=O.bitfield_len(n)).astype(np.uint8)
    pred = {O.LT: lambda v, a, b: v < a, O.LE: lambda v, a, b: v <= a, O.GT: lambda v, a, b: v > a, O.GE: lambda v, a, b: v >= a,
            O.RG: lambda v, a, b: a <= v <= b, O.EQ: lambda v, a, b: v == a, O.NE: lambda v, a, b: v != a}
    for a, b in ((b"abc", b"abcdefghi"), (b"", b"\xff"), (keys[15], keys[30]), (b"b", b"a")):
        for op, fn in pred.items():
            want = np.packbits(np.array([fn(v, a, b) for v in vals]), bitorder="little")
            got, cnt = O.bytes_match(op, ptr, buf, a, val2=b)
            assert (got == want).all() and cnt == int(np.unpackbits(want, bitorder="little")[:n].sum()), (op, a, b)
            gm, _ = O.bytes_match(op, ptr, buf, a, mask=mask, val2=b)
            wm = want & mask
            wm[-1] &= 0xFF >> (8 - n % 8)
            assert (gm == wm).all()


BYTES = helpers.bytes_golden_cases()


@pytest.mark.parametrize("op", ["eq", "ne", "lt", "le", "gt", "ge", "bw"])
def test_bytes_golden(op):
    """pkg/stringx/match_test.go:25-33 (TestCompare) through the oracle's StringPool matchers: per case the count
    (`set.Count()`), the exact result bytes and 32 untouched poison bytes behind them (match_test.go:73-76), for the
    pool as Append lays it out and for a pool whose equal rows share their bytes"""
    cases = [c for c in BYTES if c["op"] == op]
    assert len(cases) == 15 and sorted(c["n"] for c in cases) == [0, 7, 15, 23, 31, 31, 32, 32, 32, 63, 64, 64, 64, 127, 128]
    for c in cases:
        for dedup in (False, True):
            ptr, buf = helpers.bytes_golden_pool(c, dedup=dedup)
            l = O.bitfield_len(c["n"])
            bits = np.concatenate([np.zeros(l, np.uint8), np.full(32, 0xFA, np.uint8)])     # MakeBitsAndMaskPoison
            got, cnt = O.bytes_match(O.OPNAME[op], ptr, buf, bytes.fromhex(c["match"]), bits=bits, val2=bytes.fromhex(c["match2"]))
            tag = "%s/%s dedup=%s" % (op, c["name"], dedup)
            assert cnt == c["count"], tag
            assert got[:l].tobytes().hex() == c["result"], tag
            assert (got[l:] == 0xFA).all(), tag
