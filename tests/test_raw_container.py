"""'Next' row f2: raw on-disk container ingest.  Varint + container framing restated from
pkg/num/varint.go and internal/encode/{int_raw,float_raw}.go; the reference's own varint test
values (pkg/num/varint_test.go:15-27) round-trip; on the GPU the containers of consecutive packs
land as one column that scans identically to the host-decoded vectors."""
import numpy as np
import pytest

from oracle import oracle as O

UV_CASES = [0, 1, 240, 2287, 67823, 16777215, 4294967295, 1099511627775, 281474976710655, 72057594037927935, (1 << 64) - 1]
UV_LEN = [1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9]


def test_uvarint_reference_cases():
    for v, n in zip(UV_CASES, UV_LEN):
        enc = O.put_uvarint(v)
        assert len(enc) == n, v
        assert O.uvarint(enc) == (v, n)
    for v in (241, 2288, 67824, 1 << 24, (1 << 32) - 1, 1 << 32, 1 << 40, 1 << 48, 1 << 56):
        assert O.uvarint(O.put_uvarint(v))[0] == v
    # ordering property of this varint: byte-wise order == numeric order
    encs = [O.put_uvarint(v) for v in sorted(UV_CASES)]
    assert encs == sorted(encs)


def test_raw_container_roundtrip():
    rng = np.random.default_rng(3)
    for dt, npdt in ((O.I64, np.int64), (O.U16, np.uint16), (O.I8, np.int8), (O.F64, np.float64), (O.F32, np.float32)):
        for n in (0, 1, 240, 241, 5000):
            x = (rng.standard_normal(n) * 100).astype(npdt)
            c = O.raw_container_store(dt, x)
            assert c[0] == (15 if np.dtype(npdt).kind == "f" else 7)
            assert (O.raw_container_load(dt, c) == x).all()
    # float containers count BYTES, integer containers count VALUES (float_raw.go:75 vs int_raw.go:75)
    assert O.uvarint(O.raw_container_store(O.F64, np.zeros(10))[1:])[0] == 80
    assert O.uvarint(O.raw_container_store(O.I64, np.zeros(10, np.int64))[1:])[0] == 10


@pytest.mark.gpu
def test_gpu_ingest_of_raw_containers_scans_like_decoded_vectors():
    import knoxdb_amd as K
    from knoxdb_amd import And, Leaf
    from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
    pack, n_packs = 4096, 9
    with K.Context(0) as ctx:
        for bt, dt in ((BlockType.Int64, O.I64), (BlockType.Float64, O.F64), (BlockType.Int32, O.I32)):
            packs = [O.generate(dt, 0 if bt != BlockType.Float64 else 2, 11, p * pack, pack if p < n_packs - 1 else 1000, lo=-500, span=1000)
                     for p in range(n_packs)]
            conts = [O.raw_container_store(dt, p) for p in packs]
            col = ctx.upload_raw_containers(bt, conts, pack)
            whole = np.concatenate(packs)
            assert col.rows == whole.size and (col.download() == whole).all()
            bits = ctx.bits(whole.size)
            thr = 0.25 if bt == BlockType.Float64 else 17
            res = ctx.scan(And(Leaf(0, FilterMode.Le, thr)), [col], aggs=[(ReducerFunc.Count, 0)], bits_out=bits, pack_counts=True)
            wb, wc = O.cmp(dt, O.LE, whole, thr)
            assert res.count == wc and (bits.bytes() == wb).all()
            col.free(); bits.free()
        # an encoded (non-raw) container is refused, not mis-read
        with pytest.raises(K.KnoxError) as e:
            ctx.upload_raw_containers(BlockType.Int64, [bytes([4, 1, 0, 0])], pack)
        assert e.value.code == -5
