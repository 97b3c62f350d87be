"""Scan on bit-packed containers without materialization (knox_packed_*): identical to decoding the same
containers and scanning the vectors, and to the oracle's compare on the decoded values; every bit width
0..64, every mode, short last pack, signed and unsigned columns."""
import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
from oracle import oracle as O

pytestmark = pytest.mark.gpu
MODES = [FilterMode.Equal, FilterMode.NotEqual, FilterMode.Lt, FilterMode.Le, FilterMode.Gt, FilterMode.Ge, FilterMode.Range]


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def make_packs(rng, npdt, bits, pack, n_packs, last):
    info = np.iinfo(npdt)
    base = int(rng.integers(max(info.min, -(1 << 62)), min(info.max, 1 << 62) - (1 << bits) + 1)) if bits < 62 else int(info.min)
    packs = []
    for p in range(n_packs):
        n = pack if p < n_packs - 1 else last
        if bits == 0:
            v = np.full(n, base, dtype=np.int64)
        elif bits < 63:
            v = base + rng.integers(0, 1 << bits, size=n, dtype=np.int64)
            v[0], v[-1] = base, base + (1 << bits) - 1   # the container's range really needs `bits` bits
        else:
            v = rng.integers(info.min, info.max, size=n, dtype=npdt, endpoint=True).astype(np.int64) if npdt == np.int64 else None
            if v is None:
                v = rng.integers(0, info.max, size=n, dtype=np.uint64, endpoint=True).view(np.int64)
            v[0] = info.min if npdt == np.int64 else 0
            v[-1] = info.max if npdt == np.int64 else -1
        packs.append(v.astype(np.int64).view(npdt) if npdt == np.uint64 else v.astype(npdt))
    return packs


@pytest.mark.parametrize("bt,dt,npdt", [(BlockType.Int64, O.I64, np.int64), (BlockType.Uint64, O.U64, np.uint64)])
@pytest.mark.parametrize("bits", [0, 1, 2, 3, 7, 8, 13, 20, 31, 32, 33, 47, 63, 64])
def test_packed_scan_equals_decoded_scan(ctx, bt, dt, npdt, bits):
    rng = np.random.default_rng(1000 + bits + (7 if npdt == np.uint64 else 0))
    pack, n_packs, last = 8192, 3, 5000 + 37
    packs = make_packs(rng, npdt, bits, pack, n_packs, last)
    conts = [O.bitpack_container_store(dt, p) for p in packs]
    whole = np.concatenate(packs)
    n = whole.size
    pk = ctx.packed_upload(bt, conts, pack)
    assert pk.rows == n
    col = ctx.upload_raw_containers(bt, conts, pack)   # the materializing path on the same containers
    srt = np.sort(whole)
    a, b = srt[n // 3].item(), srt[2 * n // 3].item()
    for mode in MODES:
        bits_p, bits_d = ctx.bits(n), ctx.bits(n)
        bits_p.one()  # stale ones must be overwritten
        cnt, st, pc = pk.scan(mode, a, b if mode == FilterMode.Range else None, want_sum=True, bits_out=bits_p, pack_counts=True)
        res = ctx.scan(And(Leaf(0, mode, a, b if mode == FilterMode.Range else None)), [col], aggs=[(R.Sum, 0)], bits_out=bits_d, pack_counts=True)
        want, wcnt = O.cmp(dt, int(mode), whole, a, b if mode == FilterMode.Range else 0)
        gp = bits_p.bytes(poison=32)
        assert (gp == want).all() and (bits_d.bytes() == want).all(), (bits, mode)
        assert cnt == wcnt == res.count
        assert (pc == res.pack_counts).all()
        assert K.agg_value(st, R.Sum)[0] == res.value(0)[0], (bits, mode)
        bits_p.free()
        bits_d.free()
    # count only: no bitset, no sum
    cnt, _, _ = pk.scan(FilterMode.Ge, a)
    assert cnt == int((whole >= a).sum())
    pk.free()
    col.free()


def test_packed_upload_rejects_what_it_cannot_scan(ctx):
    x = np.arange(8192, dtype=np.int64)
    bp = O.bitpack_container_store(O.I64, x)
    with pytest.raises(K.KnoxError):   # pack size must be a multiple of 4096
        ctx.packed_upload(BlockType.Int64, [bp], 8000)
    with pytest.raises(K.KnoxError):   # a short pack in the middle
        ctx.packed_upload(BlockType.Int64, [O.bitpack_container_store(O.I64, x[:100]), bp], 8192)
    with pytest.raises(K.KnoxError) as e:   # raw containers do not stay packed
        ctx.packed_upload(BlockType.Int64, [O.raw_container_store(O.I64, x)], 8192)
    assert e.value.code == -5
    with pytest.raises(K.KnoxError):   # float columns are never bit-packed
        ctx.packed_upload(BlockType.Float64, [bp], 8192)


@pytest.mark.parametrize("bt,dt,npdt", [(BlockType.Int32, O.I32, np.int32), (BlockType.Uint32, O.U32, np.uint32), (BlockType.Int16, O.I16, np.int16),
                                        (BlockType.Uint16, O.U16, np.uint16), (BlockType.Int8, O.I8, np.int8), (BlockType.Uint8, O.U8, np.uint8)])
def test_packed_scan_narrow_types(ctx, bt, dt, npdt):
    """narrower integer columns: value = T(field + For) wraps in T and so does the compare (signed minimum
    as For, full-range containers)"""
    rng = np.random.default_rng(500 + int(bt))
    info = np.iinfo(npdt)
    pack, last = 4096, 1000
    for spread in (3, info.max // 2 if info.min < 0 else info.max // 3, None):
        packs = []
        for n in (pack, pack, last):
            if spread is None:
                v = rng.integers(info.min, info.max, size=n, endpoint=True).astype(npdt)
                v[0], v[-1] = info.min, info.max
            else:
                v = (info.min + rng.integers(0, spread + 1, size=n)).astype(npdt)
            packs.append(v)
        conts = [O.bitpack_container_store(dt, p) for p in packs]
        whole = np.concatenate(packs)
        n = whole.size
        pk = ctx.packed_upload(bt, conts, pack)
        col = ctx.upload_raw_containers(bt, conts, pack)
        srt = np.sort(whole)
        a, b = srt[n // 4].item(), srt[3 * n // 4].item()
        for mode in MODES:
            bits_p = ctx.bits(n)
            cnt, st, pc = pk.scan(mode, a, b if mode == FilterMode.Range else None, want_sum=True, bits_out=bits_p, pack_counts=True)
            res = ctx.scan(And(Leaf(0, mode, a, b if mode == FilterMode.Range else None)), [col], aggs=[(R.Sum, 0)])
            want, wcnt = O.cmp(dt, int(mode), whole, a, b if mode == FilterMode.Range else 0)
            assert (bits_p.bytes(poison=32) == want).all() and cnt == wcnt == res.count, (bt, mode, spread)
            assert K.agg_value(st, R.Sum)[0] == res.value(0)[0], (bt, mode, spread)
            bits_p.free()
        pk.free()
        col.free()


@pytest.mark.parametrize("bits", [0, 5, 16, 31, 32])
def test_packed_scan_general_form_for_narrow_fields(ctx, bits, monkeypatch):
    """fields of at most 32 bits scan from LDS-staged chunks (packed_scan_lds_kernel: the default, exercised by every test
    above); KNOX_PACKED_FORM=global keeps the general kernel on the same containers -- same bitset, count, sum"""
    rng = np.random.default_rng(4000 + bits)
    pack, n_packs, last = 8192, 3, 4096 + 64 + 1
    packs = make_packs(rng, np.int64, bits, pack, n_packs, last)
    conts = [O.bitpack_container_store(O.I64, p) for p in packs]
    whole = np.concatenate(packs)
    n = whole.size
    pk = ctx.packed_upload(BlockType.Int64, conts, pack)
    srt = np.sort(whole)
    a, b = srt[n // 4].item(), srt[3 * n // 4].item()
    for mode in MODES:
        got = {}
        for form in ("", "global"):
            if form:
                monkeypatch.setenv("KNOX_PACKED_FORM", form)
            else:
                monkeypatch.delenv("KNOX_PACKED_FORM", raising=False)
            bits_p = ctx.bits(n)
            bits_p.one()
            cnt, st, pc = pk.scan(mode, a, b if mode == FilterMode.Range else None, want_sum=True, bits_out=bits_p, pack_counts=True)
            got[form] = (bits_p.bytes(poison=32).tobytes(), cnt, K.agg_value(st, R.Sum)[0], pc.tobytes())
            bits_p.free()
        want, wcnt = O.cmp(O.I64, int(mode), whole, a, b if mode == FilterMode.Range else 0)
        assert got[""] == got["global"], (bits, mode)
        assert got[""][0] == want.tobytes() and got[""][1] == wcnt, (bits, mode)
    monkeypatch.delenv("KNOX_PACKED_FORM", raising=False)
    pk.free()
