"""GPU parity: synthetic generator, bloom build/probe, full-size properties."""
import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def test_device_generator_matches_spec(ctx):
    for bt, o, kind, lo, span in ((BlockType.Int64, O.I64, 0, -(1 << 40), 1 << 41), (BlockType.Float64, O.F64, 1, 0, 0),
                                  (BlockType.Float64, O.F64, 2, 0, 0), (BlockType.Uint64, O.U64, 3, 100, 0),
                                  (BlockType.Float64, O.F64, 4, 0, 0), (BlockType.Int32, O.I32, 0, -1000, 5000),
                                  (BlockType.Uint8, O.U8, 0, 0, 256), (BlockType.Float32, O.F32, 2, 0, 0),
                                  (BlockType.Uint64, O.U64, 0, 0, 0)):
        col = ctx.generate(bt, kind, 42, 100000, first_row=12345, lo=lo, span=span)
        got = col.download()
        want = O.generate(o, kind, 42, 12345, 100000, lo=lo, span=span)
        assert got.tobytes() == want.tobytes(), (bt, kind)
        col.free()


def test_bloom_build_and_probe(ctx):
    pack, n = 4096, 16 * 4096
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) ^ np.uint64(0xABCDEF)
    col = ctx.upload(ids, pack)
    bloom = ctx.bloom_build_u64(col, 2)
    filters, offsets = bloom.download()
    # device-built filters are byte-identical to the reference construction (bloom.go:48-60,162-169)
    for p in range(16):
        f = O.bloom_new(pack, 2)
        for v in ids[p * pack:(p + 1) * pack]:
            O.bloom_add(f, O.lib().ko_xxh3_u64(int(v)))
        assert (filters[int(offsets[p]):int(offsets[p + 1])] == f).all(), p
    present = ids[5 * pack + 17]
    h = K.hash_value(BlockType.Uint64, present)
    maybe = bloom.probe(h)
    want = [int(O.bloom_contains(filters[int(offsets[p]):int(offsets[p + 1])], h)) for p in range(16)]
    assert maybe.tolist() == want and maybe[5] == 1
    absent = np.uint64(12345678901234567)
    h2 = K.hash_value(BlockType.Uint64, absent)
    maybe2 = bloom.probe(h2)
    want2 = [int(O.bloom_contains(filters[int(offsets[p]):int(offsets[p + 1])], h2)) for p in range(16)]
    assert maybe2.tolist() == want2
    # filters uploaded from the host (the reference's serialized form) probe identically
    up = ctx.bloom_upload(filters, offsets)
    assert up.probe(h).tolist() == want and up.probe(h2).tolist() == want2
    up.free()
    # pruned scan == full scan (config 4 shape: bloom pre-filter then EQ, count)
    col.pack_rows = pack
    bits = ctx.bits(n)
    r1 = ctx.scan(And(Leaf(0, FilterMode.Equal, present)), [col], aggs=[(ReducerFunc.Count, 0)], bits_out=bits,
                  pack_skip=1 - maybe)
    b1 = bits.bytes().copy()
    r2 = ctx.scan(And(Leaf(0, FilterMode.Equal, present)), [col], aggs=[(ReducerFunc.Count, 0)], bits_out=bits)
    assert (b1 == bits.bytes()).all() and r1.count == r2.count == 1
    assert r1.rows_scanned == int(maybe.sum()) * pack
    col.free(); bits.free(); bloom.free()


@pytest.mark.parametrize("bt", [BlockType.Int32, BlockType.Uint32, BlockType.Float32, BlockType.Float64, BlockType.Int16, BlockType.Uint16,
                                BlockType.Int8, BlockType.Uint8, BlockType.Int64])
def test_bloom_build_every_width_matches_reference_hashes(ctx, bt):
    """device bloom build for every numeric width: byte-identical to filters built with the oracle's
    hash.HashT restatement (xxh3 u64 / u32 / u16 / u8; floats hash their IEEE bits) and bloom.Add"""
    from helpers import rand_col
    rng = np.random.default_rng(70 + int(bt))
    pack, n_packs = 512, 5
    n = pack * n_packs - 37  # short last pack
    x = rand_col(bt, n, rng)
    col = ctx.upload(x, pack)
    bloom = ctx.bloom_build(col, 2)
    filters, offsets = bloom.download()
    dt = int(bt)
    for p in range(n_packs):
        vals = x[p * pack:(p + 1) * pack]
        f = O.bloom_new(len(vals), 2)
        for v in vals:
            O.bloom_add(f, O.hash_value(dt, v))
        assert (filters[int(offsets[p]):int(offsets[p + 1])] == f).all(), (bt, p)
    # every stored value probes positive in its own pack through the public hash
    for p in (0, n_packs - 1):
        v = x[p * pack]
        assert bloom.probe(K.hash_value(bt, v))[p] == 1
    col.free()
    bloom.free()


def _oracle_filters(x, pack, factor=2):
    """per-pack filters as the reference builds them (bloom.go:48-60,162-169), one bloom.Add per DISTINCT value (OR is idempotent)"""
    out = []
    for p0 in range(0, len(x), pack):
        vals = x[p0:p0 + pack]
        f = O.bloom_new(len(vals), factor)
        for v in np.unique(vals):
            O.bloom_add(f, O.lib().ko_xxh3_u64(int(v)))
        out.append(f)
    return out


def test_bloom_build_one_read_and_its_exact_fallback(ctx):
    """Round 4: the build reads the column once, binning probe locations into FIXED regions per (pack, filter segment); a
    heavy hitter overflows its four segments' regions and the batch is rebuilt with the exact count / scan / bin / fill
    passes.  Both ways the filters are the reference's, byte for byte: a uniform column (one-read path, no exact batch), a
    column where one value fills half of every pack (exact fallback), and one with a short last pack."""
    rng = np.random.default_rng(17)
    pack = 1 << 16
    for name, n, hitter in (("uniform", 3 * pack, False), ("heavy hitter", 2 * pack, True), ("short last pack", 2 * pack + 5000, False)):
        x = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
        if hitter:
            x[rng.random(n) < 0.5] = np.uint64(0x1234567890ABCDEF)
        col = ctx.upload(x, pack)
        bloom = ctx.bloom_build(col, 2)
        batches, exact = bloom.build_info()
        assert batches >= 1 and (exact >= 1) == hitter, (name, batches, exact)
        filters, offsets = bloom.download()
        for p, f in enumerate(_oracle_filters(x, pack)):
            got = filters[int(offsets[p]):int(offsets[p + 1])]
            assert got.size == f.size and (got == f).all(), (name, p, int((got != f).sum()))
        col.free(); bloom.free()


def test_bloom_build_bytes_column(ctx):
    """device bloom build over a bytes (StringPool) column: XXH3-64 of every row on the device, filters
    byte-identical to the oracle's construction; lengths 0..300 cover every branch of the hash"""
    rng = np.random.default_rng(91)
    pack, n_packs = 256, 4
    n = pack * n_packs - 9
    lens = np.concatenate([np.arange(0, 130), [136, 144, 159, 160, 161, 200, 239, 240, 241, 255, 256, 257, 300],
                           rng.integers(0, 48, size=n - 143)])[:n]
    rng.shuffle(lens)
    vals = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in lens]
    buf = np.frombuffer(b"".join(vals) + b"\x00" * 8, dtype=np.uint8)
    ln = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
    col = ctx.upload_bytes((offs << np.uint64(32)) | ln, buf, pack)
    bloom = ctx.bloom_build(col, 2)
    filters, offsets = bloom.download()
    for p in range(n_packs):
        vs = vals[p * pack:(p + 1) * pack]
        f = O.bloom_new(len(vs), 2)
        for v in vs:
            O.bloom_add(f, O.xxh3_bytes(v))
        assert (filters[int(offsets[p]):int(offsets[p + 1])] == f).all(), p
    # probe through the library's own host hash (knox_hash_bytes)
    for i in (0, 17, n - 1):
        assert K.hash_bytes(vals[i]) == O.xxh3_bytes(vals[i])
        assert bloom.probe(K.hash_bytes(vals[i]))[i // pack] == 1
    col.free()
    bloom.free()


def test_bloom_build_bytes_rows_of_at_most_32_bytes(ctx):
    """the bytes build's batched-load path (knox_bloom.hip bloom_tile_hashes: no row of a wave's tile longer than 32 bytes -- the
    branches of XXH3 that need at most four 8-byte words, loaded for eight rows before any arithmetic): every length 0 ... 32,
    byte-unaligned rows, a ragged last pack, a pack of 17 ... 32-byte rows only, and a pack with ONE longer row (its waves take
    the general path) -- filters byte-identical to the oracle's construction either way; the last row ends at the pool's end"""
    rng = np.random.default_rng(92)
    pack, n_packs = 8192, 3
    n = pack * n_packs - 1234
    lens = rng.integers(0, 33, size=n)
    lens[:33] = np.arange(0, 33)
    lens[pack:2 * pack] = rng.integers(17, 33, size=pack)
    lens[2 * pack + 77] = 40                  # third pack: one row beyond the fast path
    lens[n - 1] = 3                           # the last row's first word reaches past the pool's end (device slack)
    vals = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in lens]
    buf = np.frombuffer(b"\x01" + b"".join(vals), dtype=np.uint8)   # + 1: rows start at odd addresses; no slack behind the last row
    ln = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = (1 + np.concatenate([[0], np.cumsum(ln)[:-1]])).astype(np.uint64)
    col = ctx.upload_bytes((offs << np.uint64(32)) | ln, buf, pack)
    bloom = ctx.bloom_build(col, 2)
    filters, offsets = bloom.download()
    for p in range(n_packs):
        vs = vals[p * pack:(p + 1) * pack]
        f = O.bloom_new(len(vs), 2)
        for v in vs:
            O.bloom_add(f, O.xxh3_bytes(v))
        assert (filters[int(offsets[p]):int(offsets[p + 1])] == f).all(), p
    col.free()
    bloom.free()
    # the same rows as one pool per pack (knox_col_bytes_create / _append_pack: every pool on its own 64-byte boundary)
    packs = []
    for p in range(n_packs):
        vs = vals[p * pack:(p + 1) * pack]
        l = np.array([len(v) for v in vs], dtype=np.uint64)
        o = np.concatenate([[0], np.cumsum(l)[:-1]]).astype(np.uint64)
        packs.append(((o << np.uint64(32)) | l, np.frombuffer(b"".join(vs) or b"\x00", dtype=np.uint8)))
    col = ctx.upload_bytes_packs(packs, pack)
    bloom2 = ctx.bloom_build(col, 2)
    f2, o2 = bloom2.download()
    assert (o2 == offsets).all() and (f2 == filters).all()
    col.free()
    bloom2.free()


def test_bloom_build_bytes_rows_of_33_to_64_bytes(ctx):
    """the second batched-load path of the bytes build (every row of a wave's tile 33 ... 64 bytes long -- URLs, paths: the eight
    8-byte words of four rows loaded before any arithmetic, knox_xxh3::hash_33_64_words): every length of the range, unaligned rows,
    a pack with one row outside it (general path for its waves), filters byte-identical to the oracle's construction"""
    rng = np.random.default_rng(93)
    pack, n_packs = 8192, 2
    n = pack * n_packs - 321
    lens = rng.integers(33, 65, size=n)
    lens[:32] = np.arange(33, 65)
    lens[pack + 5] = 20
    vals = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in lens]
    buf = np.frombuffer(b"\x07\x07\x07" + b"".join(vals), dtype=np.uint8)
    ln = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = (3 + np.concatenate([[0], np.cumsum(ln)[:-1]])).astype(np.uint64)
    col = ctx.upload_bytes((offs << np.uint64(32)) | ln, buf, pack)
    bloom = ctx.bloom_build(col, 2)
    filters, offsets = bloom.download()
    for p in range(n_packs):
        vs = vals[p * pack:(p + 1) * pack]
        f = O.bloom_new(len(vs), 2)
        for v in vs:
            O.bloom_add(f, O.xxh3_bytes(v))
        assert (filters[int(offsets[p]):int(offsets[p + 1])] == f).all(), p
    col.free()
    bloom.free()


def test_large_scan_properties(ctx):
    """2^26 rows (512 MiB): size-independent properties + sampled packs against the oracle."""
    n, pack = 1 << 26, 1 << 20
    a, b = -(1 << 38), 1 << 38
    col = ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
    col.pack_rows = pack
    bits = ctx.bits(n)
    res = ctx.scan(And(Leaf(0, FilterMode.Range, a, b)), [col], aggs=[(ReducerFunc.Sum, 0), (ReducerFunc.Min, 0), (ReducerFunc.Max, 0)],
                   bits_out=bits, pack_counts=True)
    # checksum of per-pack counts == total; popcount kernel agrees; ~25 % selectivity
    assert int(res.pack_counts.sum()) == res.count == bits.count()
    assert abs(res.count / n - 0.25) < 0.001
    # complement partition: NOT-between count + between count == n, sums add up to the column sum
    nb = ctx.bits(n)
    r_all = ctx.scan(And(Leaf(0, FilterMode.True_, 0)), [col], aggs=[(ReducerFunc.Sum, 0)])
    nb.copy_from(bits).neg()
    st = ctx.reduce(col, nb)
    assert K.agg_value(st, ReducerFunc.Count)[0] + res.count == n
    assert (int(K.agg_value(st, ReducerFunc.Sum)[0]) + int(res.value(0)[0])) % (1 << 64) == int(r_all.value(0)[0]) % (1 << 64)
    assert a <= res.value(1)[0] <= res.value(2)[0] <= b
    # sampled packs bit-exact against the oracle (same counter-based generator spec)
    got = bits.bytes()
    for p in (0, 17, 63):
        x = O.generate(O.I64, 0, 2, p * pack, pack, lo=-(1 << 40), span=1 << 41)
        wb, wc = O.cmp(O.I64, O.RG, x, a, b)
        assert (got[p * pack // 8:(p + 1) * pack // 8] == wb).all() and res.pack_counts[p] == wc
    col.free(); bits.free(); nb.free()


def test_rccl_single_rank_roundtrip(ctx):
    import ctypes as C
    from knoxdb_amd import _lib
    L = _lib.lib()
    uid = (C.c_uint8 * 128)()
    assert L.knox_comm_unique_id(uid) == 0
    assert L.knox_comm_init(ctx.h, 1, 0, uid) == 0
    x = np.arange(10000, dtype=np.int64)
    col = ctx.upload(x)
    st = ctx.reduce(col)
    arr = (_lib.AggState * 1)(st)
    out = (_lib.AggState * 1)()
    cnt = C.c_int64(st.count)
    assert L.knox_comm_allgather_states(ctx.h, arr, 1, out, C.byref(cnt)) == 0
    assert cnt.value == 10000 and out[0].isum == st.isum and out[0].max_row == 9999
    assert L.knox_comm_destroy(ctx.h) == 0
    col.free()


def test_zone_map_pruning_match_range_vectors(ctx):
    """'Next' row f3: the statistics walk on the device.  Per-pack min/max columns -> packs that may
    match (numeric MatchRangeVectors, match_num.go:357-588), then the pruned scan equals the full scan."""
    rng = np.random.default_rng(51)
    pack, n_packs = 2048, 64
    # clustered data: pack p holds values around 1000*p
    x = (np.repeat(np.arange(n_packs) * 1000, pack) + rng.integers(0, 900, pack * n_packs)).astype(np.int64)
    mins_h = x.reshape(n_packs, pack).min(axis=1)
    maxs_h = x.reshape(n_packs, pack).max(axis=1)
    col = ctx.upload(x, pack)
    mins, maxs = ctx.upload(mins_h), ctx.upload(maxs_h)
    mask_h = np.packbits(rng.random(n_packs) < 0.7, bitorder="little")
    mask = ctx.bits(n_packs).upload(mask_h)
    maskb = np.unpackbits(mask_h, bitorder="little")[:n_packs].astype(bool)
    v, frm, to = 17450, 9100, 13020
    cases = [(FilterMode.Equal, v, None, (mins_h <= v) & (maxs_h >= v), True),
             (FilterMode.Range, frm, to, (mins_h <= to) & (maxs_h >= frm), True),
             (FilterMode.Gt, v, None, maxs_h > v, False), (FilterMode.Ge, v, None, maxs_h >= v, False),
             (FilterMode.Lt, v, None, mins_h < v, False), (FilterMode.Le, v, None, mins_h <= v, False),
             (FilterMode.NotEqual, v, None, np.ones(n_packs, bool), True)]
    for mode, a, b, want, masked in cases:
        for m in (None, mask):
            got = ctx.match_range_vectors(mins, maxs, mode, a, b, mask=m)
            w = want & maskb if (m is not None and masked) else want
            assert (np.unpackbits(got.bytes(), bitorder="little")[:n_packs].astype(bool) == w).all(), (mode, m is not None)
            got.free()
    # prune + scan == full scan
    maybe = ctx.match_range_vectors(mins, maxs, FilterMode.Range, frm, to)
    skip = 1 - np.unpackbits(maybe.bytes(), bitorder="little")[:n_packs]
    bits = ctx.bits(x.size)
    tree = And(Leaf(0, FilterMode.Range, frm, to))
    r1 = ctx.scan(tree, [col], aggs=[(ReducerFunc.Sum, 0)], bits_out=bits, pack_skip=skip.astype(np.uint8))
    b1 = bits.bytes().copy()
    r2 = ctx.scan(tree, [col], aggs=[(ReducerFunc.Sum, 0)], bits_out=bits)
    assert (b1 == bits.bytes()).all() and r1.count == r2.count and r1.value(0)[0] == r2.value(0)[0]
    assert r1.rows_scanned == int((1 - skip).sum()) * pack < x.size
    for h in (col, mins, maxs, mask, maybe, bits):
        h.free()


def test_zone_map_match_range_vectors_bytes(ctx):
    """the bytes forms of MatchRangeVectors (match_bytes.go:97-299) over StringPool min / max statistics
    columns: expected values from Python's bytes order (= bytes.Compare)"""
    rng = np.random.default_rng(53)
    n_packs = 200
    def pool(vals):
        buf = np.frombuffer(b"".join(vals) + b"\x00" * 8, dtype=np.uint8)
        lens = np.array([len(v) for v in vals], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        return (offs << np.uint64(32)) | lens, buf
    keys = sorted({bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(3 * n_packs)})
    los = [keys[i] for i in sorted(rng.integers(0, len(keys) - 40, size=n_packs))]
    his = [max(lo, keys[min(len(keys) - 1, keys.index(lo) + int(rng.integers(0, 40)))]) for lo in los]
    mins, maxs = ctx.upload_bytes(*pool(los)), ctx.upload_bytes(*pool(his))
    mask_h = np.packbits(rng.random(n_packs) < 0.6, bitorder="little")
    mask = ctx.bits(n_packs).upload(mask_h)
    maskb = np.unpackbits(mask_h, bitorder="little")[:n_packs].astype(bool)
    v, frm, to = keys[len(keys) // 2], keys[len(keys) // 3], keys[len(keys) // 3 + 25]
    A = lambda f: np.array([f(lo, hi) for lo, hi in zip(los, his)])
    cases = [(FilterMode.Equal, v, None, A(lambda lo, hi: lo <= v <= hi), True),
             (FilterMode.Range, frm, to, A(lambda lo, hi: lo <= to and hi >= frm), True),
             (FilterMode.Gt, v, None, A(lambda lo, hi: hi > v), False), (FilterMode.Ge, v, None, A(lambda lo, hi: hi >= v), False),
             (FilterMode.Lt, v, None, A(lambda lo, hi: lo < v), False), (FilterMode.Le, v, None, A(lambda lo, hi: lo <= v), False),
             (FilterMode.NotEqual, v, None, np.ones(n_packs, bool), True)]
    for mode, a, b, want, masked in cases:
        for m in (None, mask):
            got = ctx.match_range_vectors_bytes(mins, maxs, mode, a, b, mask=m)
            w = want & maskb if (m is not None and masked) else want
            assert (np.unpackbits(got.bytes(), bitorder="little")[:n_packs].astype(bool) == w).all(), (mode, m is not None)
            got.free()
    for h in (mins, maxs, mask):
        h.free()


def test_zone_map_range_vectors_for_sets_and_wide_integers(ctx):
    """Round 4: the remaining MatchRangeVectors forms on the device.  numInSetMatcher (match_num.go:693-727: a set member inside
    [uint64(min), uint64(max)], Go's sign-extending conversion, reversed pairs swapped) and numNotInSetMatcher (:810-817) for
    integer statistics columns; the Int128 / Int256 matchers (match_i128.go:78-375, match_i256.go:64-229) over limb columns.
    Against the oracle's restatements (oracle.range_vectors_set / range_vectors_wide), with and without a mask."""
    rng = np.random.default_rng(23)
    n = 1000 + 37                      # packs (rows of the statistics columns)
    mask_h = rng.integers(0, 256, size=O.bitfield_len(n)).astype(np.uint8)
    mask_h[-1] &= 0xFF >> (8 - n % 8)
    mask = ctx.bits(n).upload(mask_h)
    for bt, dt, npdt in ((BlockType.Int64, O.I64, np.int64), (BlockType.Uint64, O.U64, np.uint64), (BlockType.Int32, O.I32, np.int32),
                         (BlockType.Int16, O.I16, np.int16), (BlockType.Uint8, O.U8, np.uint8)):
        info = np.iinfo(npdt)
        lo_v = max(int(info.min), -(1 << 40))
        hi_v = min(int(info.max), 1 << 40)
        a = rng.integers(lo_v, hi_v, size=n, endpoint=True)
        w = rng.integers(0, max(1, (hi_v - lo_v) // 50), size=n)
        mins_h = a.astype(npdt)
        maxs_h = np.minimum(a + w, hi_v).astype(npdt)
        mins, maxs = ctx.upload(mins_h), ctx.upload(maxs_h)
        sets = [rng.integers(lo_v, hi_v, size=k, endpoint=True).astype(npdt) for k in (1, 7, 300)]
        sets.append(np.array([mins_h[3], maxs_h[5], mins_h[3]], dtype=npdt))                # members that sit exactly on pack edges, a duplicate
        if info.min < 0:
            sets.append(np.array([-1, 0, info.min, info.max], dtype=npdt))                 # both sides of the sign change: uint64(-1) is the largest key
        for vals in sets:
            for neg in (False, True):
                for m, mh in ((None, None), (mask, mask_h)):
                    got = ctx.match_range_vectors_set(mins, maxs, vals, negate=neg, mask=m)
                    want = O.range_vectors_set(dt, mins_h, maxs_h, vals, negate=neg, mask=mh)
                    assert (got.bytes(poison=32) == want).all(), (bt, len(vals), neg, m is not None)
                    got.free()
        # an empty set matches nothing
        got = ctx.match_range_vectors_set(mins, maxs, np.zeros(0, dtype=npdt))
        assert got.count() == 0
        got.free()
        mins.free(); maxs.free()
    # Int128 / Int256: limb columns of min / max (most significant first, top limb signed)
    for nl in (2, 4):
        top = rng.integers(-3, 4, size=n).astype(np.int64)
        rest = [rng.integers(0, 1 << 62, size=n).astype(np.uint64) for _ in range(nl - 1)]
        mins_h = [top] + rest
        bump = rng.integers(0, 1 << 61, size=n).astype(np.uint64)
        maxs_h = [top] + rest[:-1] + [rest[-1] + bump]                                      # max >= min: same upper limbs, a larger lowest limb
        mins_c = [ctx.upload(x) for x in mins_h]
        maxs_c = [ctx.upload(x) for x in maxs_h]
        probes = [[int(mins_h[l][11]) for l in range(nl)], [int(maxs_h[l][500]) for l in range(nl)], [0] * nl, [-1 & (2**64 - 1)] + [5] * (nl - 1)]
        for mode, om in ((FilterMode.Equal, O.EQ), (FilterMode.NotEqual, O.NE), (FilterMode.Lt, O.LT), (FilterMode.Le, O.LE), (FilterMode.Gt, O.GT),
                         (FilterMode.Ge, O.GE), (FilterMode.Range, O.RG), (FilterMode.NotIn, O.NIN)):
            for pa in probes:
                pb = probes[1] if mode == FilterMode.Range else None
                for m, mh in ((None, None), (mask, mask_h)):
                    got = ctx.match_range_vectors_wide(mins_c, maxs_c, mode, pa, pb, mask=m)
                    want = O.range_vectors_wide(om, mins_h, maxs_h, pa, pb, mask=mh)
                    want[-1] &= 0xFF >> (8 - n % 8)
                    assert (got.bytes(poison=32) == want).all(), (nl, mode, m is not None)
                    got.free()
        for c in mins_c + maxs_c:
            c.free()
    mask.free()
