"""Bytes columns held as ONE StringPool per pack (the reference's layout: pkg/stringx/pool.go:45-58, one pool per
block; matchers run per pool, match.go:20-84) -- BASELINE configs[3]'s []byte leg: bloom pre-filter over the bytes
column, then EQ, count.  Parity against the oracle's StringPool matchers pack by pack; the pruned scan equals the full
scan."""
import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf, Or
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
from oracle import oracle as O

pytestmark = pytest.mark.gpu
R = ReducerFunc


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def make_packs(rng, n, pack, keys):
    """n rows drawn from `keys`, cut into packs of `pack` rows: per pack (ptr relative to its own pool, buf)"""
    pick = rng.integers(0, len(keys), size=n)
    vals = [keys[i] for i in pick]
    packs = []
    for p0 in range(0, n, pack):
        vs = vals[p0:p0 + pack]
        lens = np.array([len(v) for v in vs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        packs.append(((offs << np.uint64(32)) | lens, np.frombuffer(b"".join(vs) + b"\x00" * 3, dtype=np.uint8)))
    return vals, packs


def oracle_packs(fn, packs, n):
    """run a per-pool oracle matcher over every pack and concatenate the bitsets (pack sizes are multiples of 8)"""
    return np.concatenate([fn(ptr, buf) for ptr, buf in packs])[: (n + 7) // 8]


KEYS = [b"", b"a", b"short", b"0123456789abcdef", b"0123456789abcdeF", b"0123456789abcdefX", b"0123456789abcdefghijklmnopqrstu",
        b"0123456789abcdefghijklmnopqrstuv", b"0123456789abcdefghijklmnopqrstuV", b"0123456789abcdefghijklmnopqrstuvw",
        b"0123456789abcdefghijklmnopqrstuvwxyz0123456789abcdefghijklmnopqrstuvwxyz", b"0123456789abcdefghijklmnopqrstuvwxyz0123456789abcdefghijklmnopqrstuvwxyZ",
        b"\xff" * 16, b"\xff" * 15 + b"\xfe", b"\x00" * 32]


@pytest.mark.parametrize("n,pack", [(2048 + 77, 256), (64 * 5, 64), (1000, 1024)])
def test_per_pack_pools_equal_single_pool_and_oracle(ctx, n, pack):
    rng = np.random.default_rng(n)
    keys = KEYS + [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in (16, 17, 31, 32, 33, 40, 64, 100)]
    vals, packs = make_packs(rng, n, pack, keys)
    col = ctx.upload_bytes_packs(packs, pack)
    assert col.rows == n
    # the same rows as ONE pool (the round-1 layout): results must be identical
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    one = ctx.upload_bytes((offs << np.uint64(32)) | lens, np.frombuffer(b"".join(vals) + b"\x00" * 3, dtype=np.uint8), pack)
    ids = ctx.upload(np.arange(n, dtype=np.uint64), pack)
    bits, bits1 = ctx.bits(n), ctx.bits(n)
    probes = keys[:12] + [keys[-1], keys[-3], b"absent-key-of-some-length-xx", b"0123456789abcdeg"]
    for key in probes:
        for op, oo in ((FilterMode.Equal, O.EQ), (FilterMode.NotEqual, O.NE), (FilterMode.Lt, O.LT), (FilterMode.Ge, O.GE)):
            want = oracle_packs(lambda p, b: O.bytes_match(oo, p, b, key)[0], packs, n)
            r = ctx.scan(And(Leaf(0, op, key)), [col, ids], aggs=[(R.Count, 1)], bits_out=bits, pack_counts=True)
            r1 = ctx.scan(And(Leaf(0, op, key)), [one, ids], aggs=[(R.Count, 1)], bits_out=bits1, pack_counts=True)
            got = bits.bytes(poison=32)
            assert (got == want).all(), (key, op)
            assert (bits1.bytes() == want).all()
            assert r.count == r1.count == O.popcount(want, n)
            assert (r.pack_counts == r1.pack_counts).all()
    # RANGE, IN / NOT IN, a mixed tree with a numeric leaf, REGEXP
    want = oracle_packs(lambda p, b: O.bytes_match(O.RG, p, b, b"0123456789abcdef", val2=b"0123456789abcdefghijklmnopqrstuvw")[0], packs, n)
    ctx.scan(And(Leaf(0, FilterMode.Range, b"0123456789abcdef", b"0123456789abcdefghijklmnopqrstuvw")), [col, ids], bits_out=bits)
    assert (bits.bytes(poison=32) == want).all()
    for neg, op in ((False, FilterMode.In), (True, FilterMode.NotIn)):
        want = oracle_packs(lambda p, b: O.bytes_inset(neg, p, b, [keys[3], keys[7], b"", b"nope"]), packs, n)
        ctx.scan(And(Leaf(0, op, [keys[3], keys[7], b"", b"nope"])), [col, ids], bits_out=bits)
        assert (bits.bytes(poison=32) == want).all()
    weq = oracle_packs(lambda p, b: O.bytes_match(O.EQ, p, b, keys[7])[0], packs, n)
    wlt, _ = O.cmp(O.U64, O.LT, np.arange(n, dtype=np.uint64), n // 2)
    r = ctx.scan(And(Leaf(0, FilterMode.Equal, keys[7]), Leaf(1, FilterMode.Lt, n // 2)), [col, ids], aggs=[(R.Sum, 1)], bits_out=bits)
    want = O.bits_op("and", weq, wlt, n)
    assert (bits.bytes(poison=32) == want).all()
    assert int(r.value(0)[0]) == int(np.arange(n, dtype=np.uint64)[np.unpackbits(want, bitorder="little")[:n].astype(bool)].sum())
    b1, c1 = ctx.match_regexp(col, r"^0123456789abcde[fF]")
    b2, c2 = ctx.match_regexp(one, r"^0123456789abcde[fF]")
    assert c1 == c2 == sum(v.startswith(b"0123456789abcdef") or v.startswith(b"0123456789abcdeF") for v in vals)
    assert (b1.bytes() == b2.bytes()).all()
    # the pack comes back as the reference would hold it
    for p in (0, len(packs) - 1):
        ptr, buf = ctx.download_bytes_pack(col, p)
        assert (ptr == packs[p][0]).all() and (buf[: packs[p][1].size] == packs[p][1]).all()
    for o in (col, one, ids, bits, bits1, b1, b2):
        o.free()


def test_append_rules(ctx):
    """every pack but the last holds exactly pack_rows rows; offsets must stay inside the pack's own pool"""
    ptr = (np.arange(64, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
    buf = np.arange(64, dtype=np.uint8)
    with pytest.raises(K.KnoxError):
        ctx.upload_bytes_packs([(ptr[:10], buf), (ptr, buf)], 64)       # short pack followed by another one
    with pytest.raises(K.KnoxError):
        ctx.upload_bytes_packs([(ptr, buf[:63])], 64)                   # last row leaves the pool
    with pytest.raises(K.KnoxError):
        ctx.upload_bytes_packs([(ptr, buf)], 100)                       # pack_rows not a multiple of 64
    col = ctx.upload_bytes_packs([(ptr, buf), (ptr[:10], buf)], 64)
    assert col.rows == 74
    col.free()


def test_generate_bytes_matches_spec(ctx):
    """device generator == the spec text (include/knoxgpu.h), restated in numpy by the oracle"""
    pack, n = 1024, 5 * 1024 + 100
    for key_len, wb, ppw in ((32, 6, 2), (20, 4, 1), (8, 10, 4)):
        col = ctx.generate_bytes(11, n, pack, key_len, wb, ppw)
        assert col.rows == n
        for p in range((n + pack - 1) // pack):
            rows = min(pack, n - p * pack)
            want_ptr, want_buf, _ = O.generate_bytes_pack(11, p, rows, pack, key_len, wb, ppw)
            ptr, buf = ctx.download_bytes_pack(col, p)
            assert (ptr == want_ptr).all(), (key_len, p)
            assert (buf[: want_buf.size] == want_buf).all(), (key_len, p)
        col.free()


def test_bloom_prefilter_then_eq_count(ctx):
    """configs[3] flow at 2^22 rows: per-pack bloom filters over the bytes column built on the device (byte-identical
    to the oracle's on sampled packs), probe a present and an absent key, scan only the surviving packs: pruned scan ==
    full scan == the oracle's StringPool.MatchEqual on every pack the key lives in"""
    pack, n_packs, key_len, wb, ppw = 1 << 16, 64, 32, 12, 4
    n = pack * n_packs - 1000
    col = ctx.generate_bytes(3, n, pack, key_len, wb, ppw)
    ids = ctx.generate(BlockType.Uint64, 3, 0, n, lo=0, pack_rows=pack)   # id = row number
    bloom = ctx.bloom_build(col, 2)
    filters, offsets = bloom.download()
    for p in (0, n_packs - 1):
        rows = min(pack, n - p * pack)
        _, buf, _ = O.generate_bytes_pack(3, p, rows, pack, key_len, wb, ppw)
        f = O.bloom_new(rows, 2)
        for i in range(rows):
            O.bloom_add(f, O.xxh3_bytes(bytes(buf[i * key_len:(i + 1) * key_len])))
        assert (filters[int(offsets[p]):int(offsets[p + 1])] == f).all(), p
    target_row = 37 * pack + 4321
    k = int(O.bytes_key_index(3, target_row, 1, pack, wb, ppw)[0])
    present = bytes(O.bytes_key(3, k, key_len))
    absent = bytes(O.bytes_key(3, (1 << 30) + 5, key_len))
    bits = ctx.bits(n)
    for key, is_present in ((present, True), (absent, False)):
        h = K.hash_bytes(key)
        assert h == O.xxh3_bytes(key)
        maybe = bloom.probe(h)
        tree = And(Leaf(0, FilterMode.Equal, key))
        # ground truth from the spec: rows whose key index is k (only packs of k's window can hold it)
        want_rows = []
        if is_present:
            w0 = (k >> wb) * ppw
            for p in range(w0, min(w0 + ppw, n_packs)):
                rows = min(pack, n - p * pack)
                ki = O.bytes_key_index(3, p * pack, rows, pack, wb, ppw)
                want_rows += (np.flatnonzero(ki == np.uint64(k)) + p * pack).tolist()
                if np.any(ki == np.uint64(k)):
                    assert maybe[p] == 1                      # no false negatives
            assert target_row in want_rows
        assert int(maybe.sum()) <= (ppw if is_present else 0) + 3
        if is_present:
            present_rows = want_rows
        r_pruned = ctx.scan(tree, [col, ids], aggs=[(R.Count, 1), (R.Sum, 1)], bits_out=bits, pack_counts=True,
                            pack_skip=(1 - maybe).astype(np.uint8))
        pruned_bits = bits.bytes(poison=32).copy()
        r_full = ctx.scan(tree, [col, ids], aggs=[(R.Count, 1), (R.Sum, 1)], bits_out=bits, pack_counts=True)
        assert (pruned_bits == bits.bytes()).all()
        assert bits.indexes().tolist() == want_rows
        assert r_pruned.count == r_full.count == len(want_rows)
        assert int(r_pruned.value(1)[0]) == int(r_full.value(1)[0]) == sum(want_rows)
        assert (r_pruned.pack_counts == r_full.pack_counts).all()
        assert r_pruned.rows_scanned == sum(min(pack, n - p * pack) for p in np.flatnonzero(maybe))
        assert r_full.rows_scanned == n
        # one pack against the oracle's StringPool matcher itself
        p = target_row // pack
        ptr, buf, _ = O.generate_bytes_pack(3, p, pack, pack, key_len, wb, ppw)
        wbits, wcnt = O.bytes_match(O.EQ, ptr, buf, key)
        assert (pruned_bits[p * pack // 8:(p + 1) * pack // 8] == wbits).all() and r_full.pack_counts[p] == wcnt
    # bytes EQ AND a numeric range on the id column, pruned: the multi-pass plan with a live-pack list
    maybe = bloom.probe(K.hash_bytes(present))
    tree = And(Leaf(0, FilterMode.Equal, present), Leaf(1, FilterMode.Le, target_row))
    r = ctx.scan(tree, [col, ids], aggs=[(R.Max, 1)], bits_out=bits, pack_skip=(1 - maybe).astype(np.uint8))
    assert int(r.value(0)[0]) == target_row
    # an arbitrary caller-side skip list is honoured too: skipped packs report no matches
    skip = np.zeros(n_packs, dtype=np.uint8)
    skip[target_row // pack] = 1
    r = ctx.scan(And(Leaf(0, FilterMode.Equal, present)), [col, ids], aggs=[(R.Count, 1)], bits_out=bits, pack_counts=True, pack_skip=skip)
    assert r.pack_counts[target_row // pack] == 0 and target_row not in bits.indexes().tolist()
    r = ctx.scan(Or(Leaf(0, FilterMode.NotEqual, present), Leaf(1, FilterMode.Lt, 5)), [col, ids], bits_out=bits, pack_counts=True, pack_skip=skip)
    assert r.pack_counts[target_row // pack] == 0 and r.count == n - pack - sum(1 for x in present_rows if x // pack != target_row // pack and x >= 5)
    for o in (col, ids, bloom, bits):
        o.free()


def test_every_bytes_mode_on_a_column_large_enough_for_full_grids(ctx):
    """2^22 generated rows (64 packs of 2^16, one pool per pack): every wave takes ONE group of 64 rows in EQ / IN (no
    grid-stride walk) and the ordered modes run their full workgroup count -- compared with numpy on the generator's
    key indices (keys are 32 random bytes: their order is the order of their first 8 bytes, big endian)."""
    n, pack, seed = 1 << 22, 1 << 16, 11
    col = ctx.generate_bytes(seed, n, pack)
    kidx = np.concatenate([O.bytes_key_index(seed, p * pack, pack, pack) for p in range(n // pack)])
    uniq = np.unique(kidx)
    keyb = O.bytes_key(seed, uniq)                                            # (n_keys, 32)
    head = keyb[:, :8].copy().view(">u8").reshape(-1).astype(np.uint64)       # first 8 bytes as big-endian numbers
    assert np.unique(head).size == head.size                                  # so bytes.Compare == compare of the heads
    row_head = head[np.searchsorted(uniq, kidx)]
    pick = uniq[[3, len(uniq) // 2, len(uniq) - 5]]
    vals = [bytes(O.bytes_key(seed, np.array([k], dtype=np.uint64))[0]) for k in pick]
    hv = [int(head[np.searchsorted(uniq, k)]) for k in pick]
    bits = ctx.bits(n)

    def got(tree):
        res = ctx.scan(tree, [col], bits_out=bits)
        b = np.unpackbits(bits.bytes(), bitorder="little")[:n].astype(bool)
        assert res.count == int(b.sum())
        return b

    assert (got(And(Leaf(0, FilterMode.Equal, vals[1]))) == (kidx == pick[1])).all()
    assert (got(And(Leaf(0, FilterMode.NotEqual, vals[1]))) == (kidx != pick[1])).all()
    assert (got(And(Leaf(0, FilterMode.Lt, vals[1]))) == (row_head < np.uint64(hv[1]))).all()
    assert (got(And(Leaf(0, FilterMode.Ge, vals[1]))) == (row_head >= np.uint64(hv[1]))).all()
    lo, hi = sorted(zip(hv, vals))[0], sorted(zip(hv, vals))[-1]
    assert (got(And(Leaf(0, FilterMode.Range, lo[1], hi[1]))) == ((row_head >= np.uint64(lo[0])) & (row_head <= np.uint64(hi[0])))).all()
    member = np.isin(kidx, pick)
    assert (got(And(Leaf(0, FilterMode.In, vals))) == member).all()
    assert (got(And(Leaf(0, FilterMode.NotIn, vals))) == ~member).all()
    # a tree mixing them (multi-pass plan over scratch bitsets)
    want = (kidx == pick[0]) | ((row_head < np.uint64(hv[1])) & ~member)
    assert (got(Or(Leaf(0, FilterMode.Equal, vals[0]), And(Leaf(0, FilterMode.Lt, vals[1]), Leaf(0, FilterMode.NotIn, vals)))) == want).all()
    col.free(); bits.free()
