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
