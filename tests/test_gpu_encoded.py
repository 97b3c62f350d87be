"""Dictionary [5] and run-end [3] integer containers kept ENCODED in HBM and scanned as they are (knox_enc_*; the
reference matches on them without materializing: internal/encode/int_dict.go:181-380, int_runend.go:224-330).
Parity: every mode, count, wrapping sum, bitset and per-pack counts equal (a) the oracle's compare on the decoded vector
(expected vectors built here with numpy, as tests/test_raw_container.py does) and (b) the decoding path of the SAME
containers (knox_col_upload_raw_containers + knox_scan)."""
import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
from oracle import oracle as O

pytestmark = pytest.mark.gpu
R = ReducerFunc
PACK = 16384      # a multiple of 8192
MODES = ((FilterMode.Equal, O.EQ), (FilterMode.NotEqual, O.NE), (FilterMode.Lt, O.LT), (FilterMode.Le, O.LE), (FilterMode.Gt, O.GT),
         (FilterMode.Ge, O.GE), (FilterMode.Range, O.RG))
TYPES = ((BlockType.Int64, O.I64, np.int64), (BlockType.Uint64, O.U64, np.uint64), (BlockType.Int32, O.I32, np.int32),
         (BlockType.Uint16, O.U16, np.uint16), (BlockType.Int8, O.I8, np.int8))


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def dict_pack(rng, dt, npdt, n_rows, n_dict, sort=True, child="bitpack"):
    info = np.iinfo(npdt)
    lo, hi = max(int(info.min), -(1 << 40)), min(int(info.max), 1 << 40)
    span = hi - lo + 1
    n_dict = min(n_dict, span)
    if span <= (1 << 20):
        dvals = (lo + rng.choice(span, size=n_dict, replace=False)).astype(npdt)       # distinct values, random order
    else:
        dvals = np.unique(rng.integers(lo, hi, size=4 * n_dict))[:n_dict]
        dvals = rng.permutation(dvals).astype(npdt)
    n_dict = dvals.size
    if sort:
        dvals = np.sort(dvals)      # the reference's encoder sorts the dictionary (hashprobe.BuildDict); the device does not rely on it
    codes = rng.integers(0, n_dict, size=n_rows).astype(np.uint16)
    cd = O.raw_container_store(dt, dvals) if child == "raw" else O.bitpack_container_store(dt, dvals)
    cc = O.raw_container_store(O.U16, codes) if child == "raw" else O.bitpack_container_store(O.U16, codes)
    return bytes([5]) + bytes(cd) + bytes(cc), dvals[codes]


def runend_pack(rng, dt, npdt, n_rows, mean_run, child="bitpack"):
    runs = []
    left = n_rows
    while left > 0:
        r = int(min(left, max(1, rng.geometric(1.0 / mean_run))))
        runs.append(r)
        left -= r
    runs = np.array(runs)
    info = np.iinfo(npdt)
    vals = rng.integers(max(info.min, -(1 << 40)), min(info.max, 1 << 40), size=runs.size, endpoint=False).astype(npdt)
    ends = (np.cumsum(runs) - 1).astype(np.uint32)
    cv = O.raw_container_store(dt, vals)
    ce = O.raw_container_store(O.U32, ends) if child == "raw" else O.bitpack_container_store(O.U32, ends)
    return bytes([3]) + bytes(cv) + bytes(ce), np.repeat(vals, runs)


def check_all_modes(ctx, enc, bt, dt, npdt, want, containers, tag):
    n = want.size
    assert enc.rows == n
    col = ctx.upload_raw_containers(bt, containers, PACK)     # the decoding path of the same containers
    assert (col.download() == want).all(), tag
    bits, bits2 = ctx.bits(n), ctx.bits(n)
    rng = np.random.default_rng(5)
    probes = [want[rng.integers(0, n)], want[rng.integers(0, n)], want.min(), want.max(), npdt(0)]
    for mode, om in MODES:
        for a in probes:
            b = a
            if mode == FilterMode.Range:
                a, b = sorted((int(a), int(probes[1])))
            wbits, wcnt = O.cmp(dt, om, want, npdt(a), npdt(b))
            cnt, st, pc = enc.scan(mode, int(a), int(b) if mode == FilterMode.Range else None, want_sum=True, bits_out=bits, pack_counts=True)
            got = bits.bytes(poison=32)
            assert cnt == wcnt, (tag, mode, a, cnt, wcnt)
            assert (got == wbits).all(), (tag, mode, a)
            sel = np.unpackbits(wbits, bitorder="little")[:n].astype(bool)
            assert int(pc.sum()) == wcnt and all(int(pc[p]) == int(sel[p * PACK:(p + 1) * PACK].sum()) for p in range(len(pc)))
            wsum = int(want[sel].astype(np.uint64).sum(dtype=np.uint64)) if npdt().itemsize == 8 else int(want[sel].astype(np.int64).sum())
            leaf = Leaf(0, mode, int(a), int(b)) if mode == FilterMode.Range else Leaf(0, mode, int(a))
            r = ctx.scan(And(leaf), [col], aggs=[(R.Sum, 0), (R.Count, 0)], bits_out=bits2, pack_counts=True)
            assert (bits2.bytes() == got).all() and r.count == cnt and (r.pack_counts == pc).all(), (tag, mode, a)
            assert st.isum == r.states[0].isum, (tag, mode, a)            # the wrapping sum of the selected values, bit for bit
            # and against numpy: the sum in T's own arithmetic
            mask_w = (1 << (8 * npdt().itemsize)) - 1
            assert (st.isum & mask_w) == (wsum & mask_w), (tag, mode, a)
            # without a bitset: count and sum only
            cnt2, st2, _ = enc.scan(mode, int(a), int(b) if mode == FilterMode.Range else None, want_sum=True)
            assert cnt2 == cnt and st2.isum == st.isum
    for o in (col, bits, bits2):
        o.free()


@pytest.mark.parametrize("bt,dt,npdt", TYPES)
def test_dictionary_columns_scan_like_their_decoded_vectors(ctx, bt, dt, npdt):
    rng = np.random.default_rng(int(bt))
    # three packs, the last one short; dictionaries of different sizes (the largest beyond the per-code counters of the sum (kEncHistCodes), so the L2 path runs too)
    shapes = [(PACK, 300), (PACK, 9000 if npdt().itemsize >= 4 else 200), (PACK - 3001, 17)]
    packs = [dict_pack(rng, dt, npdt, n, d, sort=(i != 1), child="raw" if i == 2 else "bitpack") for i, (n, d) in enumerate(shapes)]
    containers = [p[0] for p in packs]
    want = np.concatenate([p[1] for p in packs])
    enc = ctx.enc_upload(bt, containers, PACK)
    assert enc.kinds() == (3, 0, 0, 0) and (enc.nbytes < want.nbytes or npdt().itemsize <= 2)
    check_all_modes(ctx, enc, bt, dt, npdt, want, containers, "dict %s" % bt)
    enc.free()


@pytest.mark.parametrize("bt,dt,npdt", TYPES)
def test_runend_columns_scan_like_their_decoded_vectors(ctx, bt, dt, npdt):
    rng = np.random.default_rng(100 + int(bt))
    packs = [runend_pack(rng, dt, npdt, PACK, 40), runend_pack(rng, dt, npdt, PACK, 3, child="raw"), runend_pack(rng, dt, npdt, PACK - 777, 900)]
    containers = [p[0] for p in packs]
    want = np.concatenate([p[1] for p in packs])
    enc = ctx.enc_upload(bt, containers, PACK)
    assert enc.kinds() == (0, 3, 0, 0)
    check_all_modes(ctx, enc, bt, dt, npdt, want, containers, "run-end %s" % bt)
    enc.free()


def test_mixed_runs_of_packs_keep_every_pack_in_its_cheapest_form(ctx):
    """the reference's encoder picks the scheme per block: one column, seven packs -- dictionary, run-end, constant, delta (positive
    and wrapping negative step), raw, bit-packed, a run-end pack whose values child is a dictionary (nested: decoded) -- scanned in one
    call, every mode, against the oracle on the decoded vector and against the decoding path of the same containers"""
    rng = np.random.default_rng(77)
    for bt, dt, npdt in ((BlockType.Int64, O.I64, np.int64), (BlockType.Uint32, O.U32, np.uint32), (BlockType.Int16, O.I16, np.int16)):
        info = np.iinfo(npdt)
        mask_w = (1 << (8 * npdt().itemsize)) - 1
        d_c, d_v = dict_pack(rng, dt, npdt, PACK, 500)
        r_c, r_v = runend_pack(rng, dt, npdt, PACK, 25)
        cval = int(d_v[7])
        c_c = bytes([1]) + bytes(O.put_uvarint(cval & mask_w)) + bytes(O.put_uvarint(PACK))
        c_v = np.full(PACK, cval, dtype=npdt)
        base, step = int(info.min) + 5 if info.min < 0 else 3, 3
        de_c = bytes([2]) + bytes(O.put_uvarint(base & mask_w)) + bytes(O.put_uvarint(step)) + bytes(O.put_uvarint(PACK))
        de_v = ((base + step * np.arange(PACK, dtype=object)) & mask_w).astype(np.uint64).astype(np.dtype("u%d" % npdt().itemsize)).view(npdt)
        neg_step = mask_w            # T(-1): the sequence counts down and wraps
        dn_c = bytes([2]) + bytes(O.put_uvarint(100)) + bytes(O.put_uvarint(neg_step)) + bytes(O.put_uvarint(PACK))
        dn_v = ((100 + neg_step * np.arange(PACK, dtype=object)) & mask_w).astype(np.uint64).astype(np.dtype("u%d" % npdt().itemsize)).view(npdt)
        raw_v = rng.integers(max(int(info.min), -(1 << 40)), min(int(info.max), 1 << 40), size=PACK, endpoint=True).astype(npdt)
        raw_c = bytes(O.raw_container_store(dt, raw_v))
        bp_v = (np.abs(raw_v.astype(np.int64)) % 1000).astype(npdt)
        bp_c = bytes(O.bitpack_container_store(dt, bp_v))
        # nested: run-end whose values child is a dictionary container -> no native form, decoded on the device (short last pack)
        nd_c, nd_vals = dict_pack(rng, dt, npdt, 300, 40)           # 300 run values behind a 40-entry dictionary
        runs = rng.integers(1, 40, size=300)
        ends = (np.cumsum(runs) - 1).astype(np.uint32)
        n_last = int(ends[-1]) + 1
        assert n_last <= PACK
        ne_c = bytes([3]) + nd_c + bytes(O.raw_container_store(O.U32, ends))
        ne_v = np.repeat(nd_vals, runs)
        containers = [d_c, r_c, c_c, de_c, dn_c, raw_c, bp_c, ne_c]
        want = np.concatenate([d_v, r_v, c_v, de_v, dn_v, raw_v, bp_v, ne_v])
        enc = ctx.enc_upload(bt, containers, PACK)
        assert enc.kinds() == (1, 1, 3, 3), enc.kinds()
        assert enc.container_bytes == sum(len(c) for c in containers)
        check_all_modes(ctx, enc, bt, dt, npdt, want, containers, "mixed %s" % bt)
        enc.free()


def test_enc_upload_rules(ctx):
    rng = np.random.default_rng(1)
    d, _ = dict_pack(rng, O.I64, np.int64, PACK, 50)
    r, _ = runend_pack(rng, O.I64, np.int64, PACK, 20)
    short, _ = dict_pack(rng, O.I64, np.int64, PACK - 5, 50)
    with pytest.raises(K.KnoxError) as e:
        ctx.enc_upload(BlockType.Int64, [short, d], PACK)          # only the last pack may be short
    assert e.value.code == -1
    mixed = ctx.enc_upload(BlockType.Int64, [d, r, bytes(O.raw_container_store(O.I64, np.arange(PACK, dtype=np.int64)))], PACK)
    assert mixed.kinds() == (1, 1, 0, 1) and mixed.rows == 3 * PACK
    mixed.free()
    with pytest.raises(K.KnoxError):
        ctx.enc_upload(BlockType.Int64, [d], 1000)  # pack_rows not a multiple of 8192
    with pytest.raises(K.KnoxError) as e:
        ctx.enc_upload(BlockType.Float64, [bytes([12]) + bytes(O.raw_container_store(O.F64, np.arange(4.0))) + bytes(O.raw_container_store(O.U16, np.zeros(PACK, np.uint16)))], PACK)
    assert e.value.code == -5                       # float dictionaries take the decoding path
