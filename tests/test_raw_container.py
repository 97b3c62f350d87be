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
            ctx.upload_raw_containers(BlockType.Int64, [bytes([8, 1, 0, 0])], pack)  # TInt128: not a 64-bit column container
        assert e.value.code == -5


def _bp_cases(rng):
    for dt, npdt in ((O.I64, np.int64), (O.U16, np.uint16), (O.I8, np.int8), (O.U64, np.uint64), (O.I32, np.int32), (O.U8, np.uint8)):
        info = np.iinfo(npdt)
        for n in (1, 63, 64, 65, 1000):
            yield dt, npdt, np.full(n, 42 if info.max >= 42 else 1, dtype=npdt)                    # log2 = 0
            yield dt, npdt, (rng.integers(0, 4, n) + 7).astype(npdt)                                 # 2 bits
            yield dt, npdt, rng.integers(0, 200, n).astype(np.int64).astype(npdt)                    # 8 bits (wraps for int8)
            yield dt, npdt, np.array([info.min, info.max] * n, dtype=npdt)[:max(n, 2)]               # full width
            if npdt in (np.int64, np.int32):
                yield dt, npdt, (rng.integers(-(1 << 20), 1 << 20, n)).astype(npdt)                  # negative frame of reference


def test_bitpack_container_roundtrip_and_framing():
    rng = np.random.default_rng(8)
    for dt, npdt, x in _bp_cases(rng):
        c = O.bitpack_container_store(dt, x)
        assert c[0] == 4
        assert (O.bitpack_container_load(dt, c, x.size) == x).all(), (dt, x[:4])
    # framing by hand: values 7..10 -> For 7, log2 2, N 4, one 64-bit word 0b11_10_01_00
    c = O.bitpack_container_store(O.U16, np.array([7, 8, 9, 10], dtype=np.uint16))
    assert c == bytes([4, 7, 2, 4, 0b11100100, 0, 0, 0, 0, 0, 0, 0])


@pytest.mark.gpu
def test_gpu_decodes_bitpacked_containers_on_device():
    import knoxdb_amd as K
    from knoxdb_amd.types import BlockType
    BT = {O.I64: BlockType.Int64, O.U16: BlockType.Uint16, O.I8: BlockType.Int8, O.U64: BlockType.Uint64, O.I32: BlockType.Int32, O.U8: BlockType.Uint8}
    rng = np.random.default_rng(9)
    with K.Context(0) as ctx:
        for dt, npdt, x in _bp_cases(rng):
            col = ctx.upload_raw_containers(BT[dt], [O.bitpack_container_store(dt, x)], 1024)
            assert col.rows == x.size and (col.download() == x).all(), (dt, x[:4])
            col.free()
        # a run of packs mixing raw and bit-packed containers lands as one column
        packs = [rng.integers(-500, 500, 4096).astype(np.int64) for _ in range(6)]
        conts = [O.bitpack_container_store(O.I64, p) if i % 2 else O.raw_container_store(O.I64, p) for i, p in enumerate(packs)]
        assert sum(len(c) for c in conts[1::2]) < 0.2 * sum(len(c) for c in conts[0::2])   # 10 bits instead of 64
        col = ctx.upload_raw_containers(BlockType.Int64, conts, 4096)
        assert (col.download() == np.concatenate(packs)).all()
        col.free()


def _const(val, n, w_mask):
    return bytes([1]) + O.put_uvarint(val & w_mask) + O.put_uvarint(n)


def _delta(base, delta, n, w_mask):
    return bytes([2]) + O.put_uvarint(base & w_mask) + O.put_uvarint(delta & w_mask) + O.put_uvarint(n)


@pytest.mark.gpu
def test_gpu_ingest_of_const_and_delta_containers():
    """[1][Val][N] (int_const.go:67-71), [2][For][Delta][N] with value i = T(i)*Delta + For wrapping in T
    (int_delta.go:70-95), [10][LE bits][N] (float_const.go:67-71) -- filled on the device, mixed with raw
    containers inside one column"""
    import knoxdb_amd as K
    from knoxdb_amd.types import BlockType
    with K.Context(0) as ctx:
        for bt, dt, npdt in ((BlockType.Int64, O.I64, np.int64), (BlockType.Uint32, O.U32, np.uint32), (BlockType.Int16, O.I16, np.int16),
                             (BlockType.Uint8, O.U8, np.uint8)):
            bits_w = np.dtype(npdt).itemsize * 8
            m = (1 << bits_w) - 1
            n = 5000
            i = np.arange(n, dtype=np.uint64)
            def T(v):  # wrap into the column type like Go's T(...)
                return (np.asarray(v, dtype=np.uint64) & np.uint64(m)).astype(np.dtype("u%d" % (bits_w // 8))).view(npdt) if bits_w < 64 else np.asarray(v, dtype=np.uint64).view(np.int64).astype(npdt)
            parts = [(_const(-7 if np.dtype(npdt).kind == "i" else 200, n, m), T(np.full(n, (-7 if np.dtype(npdt).kind == "i" else 200) & m, dtype=np.uint64))),
                     (_delta(1000, 3, n, m), T((i * np.uint64(3) + np.uint64(1000)))),
                     (_delta(5, -2 & m, n, m), T(i * np.uint64((-2) & m) + np.uint64(5))),   # negative step, wraps in T
                     (O.raw_container_store(dt, T(i * np.uint64(7))), T(i * np.uint64(7)))]
            col = ctx.upload_raw_containers(bt, [p for p, _ in parts], n)
            want = np.concatenate([w for _, w in parts])
            got = col.download()
            assert got.dtype == want.dtype and (got == want).all(), bt
            col.free()
        for bt, npdt, code in ((BlockType.Float64, np.float64, "<d"), (BlockType.Float32, np.float32, "<f")):
            import struct
            c = bytes([10]) + struct.pack(code, -2.5) + O.put_uvarint(3000)
            col = ctx.upload_raw_containers(bt, [c, O.raw_container_store(O.F64 if npdt == np.float64 else O.F32, np.arange(10, dtype=npdt))], 3000)
            got = col.download()
            assert (got[:3000] == npdt(-2.5)).all() and (got[3000:] == np.arange(10, dtype=npdt)).all()
            col.free()


@pytest.mark.gpu
def test_gpu_ingest_of_dictionary_and_runend_containers():
    """nested containers: [5]/[12][dict of T][uint16 codes] (int_dict.go:76-103, float_dict.go:75-102) and
    [3]/[11][values of T][uint32 inclusive ends] (int_runend.go:84-112, float_runend.go:80-110), children
    raw / bit-packed / delta; expected vectors built here with numpy"""
    import knoxdb_amd as K
    from knoxdb_amd.types import BlockType
    rng = np.random.default_rng(9)
    with K.Context(0) as ctx:
        for bt, dt, npdt, isf in ((BlockType.Int64, O.I64, np.int64, False), (BlockType.Uint16, O.U16, np.uint16, False),
                                  (BlockType.Float64, O.F64, np.float64, True), (BlockType.Int32, O.I32, np.int32, False)):
            # dictionary: 300 distinct values, 20000 rows
            if isf:
                dvals = (rng.standard_normal(300) * 1e3).astype(npdt)
            else:
                info = np.iinfo(npdt)
                dvals = rng.integers(info.min // 2, info.max // 2, size=300).astype(npdt)
            codes = rng.integers(0, 300, size=20000).astype(np.uint16)
            child_dict = O.raw_container_store(dt, dvals) if isf else O.bitpack_container_store(dt, dvals)
            child_codes = O.bitpack_container_store(O.U16, codes)
            dict_c = bytes([12 if isf else 5]) + bytes(child_dict) + bytes(child_codes)
            # run-end: 500 runs of random lengths
            runs = rng.integers(1, 90, size=500)
            ends = (np.cumsum(runs) - 1).astype(np.uint32)
            rvals = dvals[rng.integers(0, 300, size=500)]
            child_vals = O.raw_container_store(dt, rvals)
            child_ends = O.bitpack_container_store(O.U32, ends)
            run_c = bytes([11 if isf else 3]) + bytes(child_vals) + bytes(child_ends)
            raw_c = O.raw_container_store(dt, dvals[:100])
            col = ctx.upload_raw_containers(bt, [dict_c, run_c, raw_c], 20000)
            want = np.concatenate([dvals[codes], np.repeat(rvals, runs), dvals[:100]])
            got = col.download()
            assert got.size == want.size and (got == want).all(), bt
            col.free()
        # run-end whose values child is a delta container, inside a column of int64
        runs = np.array([3, 1, 10, 6])
        ends = (np.cumsum(runs) - 1).astype(np.uint32)
        delta_child = bytes([2]) + O.put_uvarint(100) + O.put_uvarint(7) + O.put_uvarint(4)  # 100, 107, 114, 121
        run_c = bytes([3]) + delta_child + bytes(O.raw_container_store(O.U32, ends))
        col = ctx.upload_raw_containers(BlockType.Int64, [run_c], 64)
        assert (col.download() == np.repeat(np.array([100, 107, 114, 121], dtype=np.int64), runs)).all()
        col.free()
        # string containers ([16]) are refused, not mis-read
        with pytest.raises(K.KnoxError) as e:
            ctx.upload_raw_containers(BlockType.Float64, [bytes([16, 1, 2, 3])], 64)
        assert e.value.code == -5


S8_N = [128, 128, 60, 30, 20, 15, 12, 10, 8, 7, 6, 5, 4, 3, 2, 1]
S8_BITS = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15, 20, 30, 60]


def s8b_encode(vals):
    """a valid Simple8 stream for `vals` (each < 2^60): greedy, densest selector whose next n values
    fit; selectors 0 / 1 for runs of 128 zeros / ones; the last word may be partly filled.  (The
    reference encoder may choose other words: decoding is what the format fixes.)"""
    words, i, n = [], 0, len(vals)
    while i < n:
        if i + 128 <= n and all(v == vals[i] for v in vals[i:i + 128]) and vals[i] in (0, 1):
            words.append(int(vals[i]) << 60)
            i += 128
            continue
        for sel in range(2, 16):
            k, b = S8_N[sel], S8_BITS[sel]
            chunk = vals[i:i + k]
            if all(int(v) < (1 << b) for v in chunk):
                w = sel << 60
                for j, v in enumerate(chunk):
                    w |= int(v) << (j * b)
                words.append(w)
                i += len(chunk)
                break
        else:
            raise ValueError("value too large for simple8")
    return np.array(words, dtype=np.uint64).tobytes()


@pytest.mark.gpu
def test_gpu_ingest_of_simple8_containers():
    """[6][For][N][len][words] (int_s8b.go:87-115): selector / count table and LSB-first fields of
    s8b/generic/decode.go + unpack_for.go; value = field + For, wrapping in T"""
    import knoxdb_amd as K
    from knoxdb_amd.types import BlockType
    rng = np.random.default_rng(19)
    with K.Context(0) as ctx:
        for bt, npdt in ((BlockType.Int64, np.int64), (BlockType.Uint32, np.uint32), (BlockType.Int16, np.int16), (BlockType.Uint8, np.uint8)):
            wbits = np.dtype(npdt).itemsize * 8
            cap = min(wbits, 40)
            # mixed magnitudes so that many selectors occur, runs of zeros / ones, a ragged tail
            mags = rng.integers(0, cap + 1, size=30000)
            fields = np.array([int(rng.integers(0, 1 << int(m))) if m else 0 for m in mags], dtype=np.uint64)
            fields[1000:1000 + 300] = 0
            fields[5000:5000 + 200] = 1
            fields = fields[:29987]
            minv = -1234 if np.dtype(npdt).kind == "i" else 77
            packed = s8b_encode(fields.tolist())
            c = bytes([6]) + O.put_uvarint(minv & ((1 << wbits) - 1)) + O.put_uvarint(fields.size) + O.put_uvarint(len(packed)) + packed
            col = ctx.upload_raw_containers(bt, [c, c], fields.size)
            want64 = (fields + np.uint64(minv & ((1 << 64) - 1)))  # wraps mod 2^64, then truncates to T
            want = (want64 & np.uint64((1 << wbits) - 1)).astype(np.dtype("u%d" % (wbits // 8))).view(npdt) if wbits < 64 else want64.view(np.int64).astype(npdt)
            got = col.download()
            assert got.size == 2 * fields.size and (got[:fields.size] == want).all() and (got[fields.size:] == want).all(), bt
            col.free()
        # a header that claims more values than the words hold is refused
        packed = s8b_encode([1, 2, 3])
        bad = bytes([6]) + O.put_uvarint(0) + O.put_uvarint(100) + O.put_uvarint(len(packed)) + packed
        with pytest.raises(K.KnoxError):
            ctx.upload_raw_containers(BlockType.Int64, [bad], 64)


@pytest.mark.gpu
def test_gpu_ingest_of_alp_containers():
    """[13][Exponent][Factor][flags][Values][Patches][Positions] (float_alp.go:109-165): value =
    T(enc) * F10[factor] * IF10[exponent], two separate multiplications in T (alp/decoder.go:114-116),
    then the exceptions overwrite their positions"""
    import knoxdb_amd as K
    from knoxdb_amd.types import BlockType
    rng = np.random.default_rng(23)
    F10 = [10.0 ** k for k in range(24)]
    IF10 = [float("1e-%d" % k) if k else 1.0 for k in range(21)]
    with K.Context(0) as ctx:
        n = 20000
        for ex, fa in ((2, 0), (5, 1), (14, 3), (0, 0)):
            enc = rng.integers(-10**9, 10**9, size=n).astype(np.int64)
            want = (enc.astype(np.float64) * np.float64(F10[fa])) * np.float64(IF10[ex])
            pos = np.sort(rng.choice(n, size=37, replace=False)).astype(np.uint32)
            pat = rng.standard_normal(37) * 1e300
            pat[:3] = [np.nan, np.inf, -0.0]
            want_p = want.copy()
            want_p[pos] = pat
            vals_c = bytes(O.bitpack_container_store(O.I64, enc))
            plain = bytes([13]) + O.put_uvarint(ex) + O.put_uvarint(fa) + bytes([0]) + vals_c
            patched = bytes([13]) + O.put_uvarint(ex) + O.put_uvarint(fa) + bytes([1]) + vals_c + bytes(O.raw_container_store(O.F64, pat)) + bytes(O.raw_container_store(O.U32, pos))
            col = ctx.upload_raw_containers(BlockType.Float64, [plain, patched], n)
            got = col.download()
            assert (got[:n].view(np.uint64) == want.view(np.uint64)).all(), (ex, fa)
            assert (got[n:].view(np.uint64) == want_p.view(np.uint64)).all(), (ex, fa, "patched")
            col.free()
        # float32: E = int32, arithmetic in float32
        enc = rng.integers(-10**6, 10**6, size=n).astype(np.int32)
        want = (enc.astype(np.float32) * np.float32(10.0)) * np.float32(0.001)
        c = bytes([13]) + O.put_uvarint(3) + O.put_uvarint(1) + bytes([0]) + bytes(O.raw_container_store(O.I32, enc))
        col = ctx.upload_raw_containers(BlockType.Float32, [c], n)
        assert (col.download().view(np.uint32) == want.view(np.uint32)).all()
        col.free()
        # ALP-RD: IEEE bits = uint(left) << shift | right (float_alprd.go:80-110, alp/rd.go:200-225)
        x = rng.standard_normal(n) * 1e-3 + 97.25
        bits64 = x.view(np.uint64)
        shift = 48
        left = (bits64 >> np.uint64(shift)).astype(np.uint16)
        right = bits64 & np.uint64((1 << shift) - 1)
        c = bytes([14]) + bytes(O.bitpack_container_store(O.U16, left)) + bytes(O.bitpack_container_store(O.U64, right)) + O.put_uvarint(shift)
        col = ctx.upload_raw_containers(BlockType.Float64, [c], n)
        assert (col.download().view(np.uint64) == bits64).all()
        col.free()
        xf = (rng.standard_normal(n) * 3).astype(np.float32)
        b32 = xf.view(np.uint32)
        c = bytes([14]) + bytes(O.raw_container_store(O.U16, (b32 >> np.uint32(20)).astype(np.uint16))) + \
            bytes(O.raw_container_store(O.U32, b32 & np.uint32((1 << 20) - 1))) + O.put_uvarint(20)
        col = ctx.upload_raw_containers(BlockType.Float32, [c], n)
        assert (col.download().view(np.uint32) == b32).all()
        col.free()
        # 128-bit containers are not column containers of this path
        with pytest.raises(K.KnoxError) as e:
            ctx.upload_raw_containers(BlockType.Float64, [bytes([8, 1, 2, 3])], 64)
        assert e.value.code == -5


@pytest.mark.gpu
def test_gpu_ingest_of_string_containers():
    """[16] constant, [17] fixed, [18] compact, [19] dictionary string containers (string_*.go) become one
    StringPool column; checked through the bytes matchers against the oracle on the expected rows"""
    import knoxdb_amd as K
    from knoxdb_amd import And, Leaf
    from knoxdb_amd.types import FilterMode
    rng = np.random.default_rng(29)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(0, 14)), dtype=np.uint8)) for _ in range(40)] + [b"", b"zz"]
    u32 = lambda a: bytes(O.raw_container_store(O.U32, np.asarray(a, dtype=np.uint32)))
    bp32 = lambda a: bytes(O.bitpack_container_store(O.U32, np.asarray(a, dtype=np.uint32)))
    # constant
    c_const = bytes([16]) + O.put_uvarint(500) + O.put_uvarint(5) + b"hello"
    rows_const = [b"hello"] * 500
    # fixed
    fx = [bytes(rng.integers(97, 100, size=4, dtype=np.uint8)) for _ in range(700)]
    c_fixed = bytes([17]) + O.put_uvarint(700) + O.put_uvarint(4) + b"".join(fx)
    # compact: rows reference slices of one buffer (shared / overlapping slices are allowed)
    buf = b"".join(words)
    starts = np.cumsum([0] + [len(w) for w in words])[:-1]
    pick = rng.integers(0, len(words), size=900)
    c_compact = bytes([18]) + bp32(starts[pick]) + u32([len(words[i]) for i in pick]) + O.put_uvarint(len(buf)) + buf
    rows_compact = [words[i] for i in pick]
    # dictionary: entries (ofs, len) + codes
    codes = rng.integers(0, len(words), size=1100)
    c_dict = bytes([19]) + u32(starts) + bp32([len(w) for w in words]) + bp32(codes) + O.put_uvarint(len(buf)) + buf
    rows_dict = [words[i] for i in codes]
    rows = rows_const + fx + rows_compact + rows_dict
    n = len(rows)
    with K.Context(0) as ctx:
        col = ctx.upload_string_containers([c_const, c_fixed, c_compact, c_dict], 512)
        assert col.rows == n
        pool = np.frombuffer(b"".join(rows) + b"\\x00" * 8, dtype=np.uint8)
        lens = np.array([len(r) for r in rows], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        ptr = (offs << np.uint64(32)) | lens
        for mode, a, b in ((FilterMode.Equal, b"hello", None), (FilterMode.Equal, b"", None), (FilterMode.Lt, words[5], None),
                           (FilterMode.Ge, b"m", None), (FilterMode.Range, b"b", b"q"), (FilterMode.NotEqual, fx[0], None)):
            bits = ctx.bits(n)
            r = ctx.scan(And(Leaf(0, mode, a, b)), [col], bits_out=bits)
            want, cnt = O.bytes_match(int(mode), ptr, pool, a, val2=b)
            assert (bits.bytes() == want).all() and r.count == cnt, mode
            bits.free()
        bits = ctx.bits(n)
        st = [words[3], b"hello", fx[7], b"absent"]
        ctx.scan(And(Leaf(0, FilterMode.In, st)), [col], bits_out=bits)
        assert (bits.bytes() == O.bytes_inset(False, ptr, pool, st)).all()
        bits.free()
        col.free()
        # a compact container whose rows leave its buffer is refused
        bad = bytes([18]) + u32([0, 5]) + u32([4, 400]) + O.put_uvarint(8) + b"abcdefgh"
        with pytest.raises(K.KnoxError) as e:
            ctx.upload_string_containers([bad], 64)
        assert e.value.code == -7


@pytest.mark.gpu
def test_gpu_ingest_of_wide_and_bitmap_containers():
    """[8]/[9] Int128 / Int256 containers -> limb columns for knox_match_wide (int128.go:115-137,
    int256.go:121-150); [20]/[21]/[22] bitmap containers -> bool column bitsets (bitmap.go:103-135)"""
    import knoxdb_amd as K
    from knoxdb_amd.types import FilterMode
    rng = np.random.default_rng(41)
    n = 3000
    with K.Context(0) as ctx:
        for nl, tag in ((2, 8), (4, 9)):
            limbs = [rng.integers(0, 3, size=n).astype(np.uint64) for _ in range(nl)]
            limbs[0] = (limbs[0].astype(np.int64) - 1).view(np.uint64)
            limbs[-1] = rng.integers(0, 1 << 40, size=n).astype(np.uint64)
            kids = [bytes(O.bitpack_container_store(O.I64 if l == 0 else O.U64, limbs[l].view(np.int64) if l == 0 else limbs[l])) for l in range(nl)]
            kids[1] = bytes(O.raw_container_store(O.U64, limbs[1]))  # mixed child encodings
            c = bytes([tag]) + b"".join(kids)
            cols = ctx.upload_wide_containers(nl, [c, c], n)
            for l in range(nl):
                assert (cols[l].download().view(np.uint64) == np.concatenate([limbs[l], limbs[l]])).all()
            a = [int(x[n // 2]) for x in limbs]
            bits, cnt = ctx.match_wide(cols, FilterMode.Le, a)
            want, wcnt = O.cmp_wide(O.OPNAME["le"], [np.concatenate([x, x]) for x in limbs], a)
            assert (bits.bytes() == want).all() and cnt == wcnt
            for o in cols:
                o.free()
            bits.free()
        m = 1003
        dense = np.packbits(rng.random(m) < 0.4, bitorder="little")
        for cont, want in ((bytes([20]) + O.put_uvarint(m) + O.put_uvarint(0), np.zeros_like(dense)),
                           (bytes([21]) + O.put_uvarint(m) + O.put_uvarint(0), O.bits_one(m)),
                           (bytes([22]) + O.put_uvarint(m) + O.put_uvarint(dense.size) + dense.tobytes(), dense)):
            b = ctx.bits(m).upload_bitmap_container(cont)
            assert (b.bytes(poison=32) == want).all()
            # the bool matcher over it (match_bits.go): column == true
            out = ctx.bits(m)
            b.match_bool(FilterMode.Equal, True, out=out)
            assert (out.bytes() == want).all()
            b.free()
            out.free()
        # sparse [23] = a serialized xroar bitmap: see test_gpu_sparse_bitmap_container; a payload shorter than 8 bytes is
        # the empty bitmap (NewFromBytes, xroar/bitmap.go:98-102), a node size beyond the buffer is refused
        b = ctx.bits(m).upload_bitmap_container(bytes([23]) + O.put_uvarint(m) + O.put_uvarint(4) + b"abcd")
        assert not b.bytes().any()
        b.free()
        with pytest.raises(K.KnoxError) as e:
            ctx.bits(m).upload_bitmap_container(bytes([23]) + O.put_uvarint(m) + O.put_uvarint(16) + b"\xff" * 16)
        assert e.value.code == -7


def sroar_from_sorted(vals):
    """The bytes xroar.NewFromSorted(vals).Bytes() lays out (internal/xroar/bitmap.go:52-79,121-185, keys.go:3-26,
    container.go:33-78,356-377), restated as a builder for tests: uint16 words; a key node of u64s [node size in words,
    number of keys, (key, container offset) ...]; the empty 64-word container New() makes for key 0; then one container
    per 65536-value block: <= 2048 values -> array [size = 8 + n][type 0][cardinality][0] + the sorted low words, more ->
    bitmap [4100][type 1][cardinality lo][cardinality hi] + 4096 words, value x at word x >> 4, bit 15 - (x & 15)."""
    blocks = {}
    for v in vals:
        blocks.setdefault(int(v) & ~0xffff, []).append(int(v) & 0xffff)
    if not blocks:
        return b""                                     # Bytes() of an empty bitmap is nil
    keys = sorted(set(blocks) | {0})
    cap = len(keys) + 1                                # one spare slot, as a node that was grown for the keys has
    node_words = 4 * (2 + 2 * cap)
    words = [0] * node_words
    off0 = len(words)
    words += [64, 0, 0, 0] + [0] * 60                  # New(): the container of key 0, minContainerSize
    offs = {0: off0}
    for k in sorted(blocks):
        lows = blocks[k]
        off = len(words)
        if len(lows) <= 2048:
            words += [8 + len(lows), 0, len(lows), 0] + lows + [0, 0, 0, 0]
        else:
            card = len(lows)
            c = [4100, 1, min(card, 0xffff), max(card - 0xffff, 0)] + [0] * 4096
            for x in lows:
                c[4 + (x >> 4)] |= 1 << (15 - (x & 15))
            words += c
        offs[k] = off                                  # setKey: key 0's offset is replaced, the empty container stays behind
    node = [node_words, len(keys)]
    for k in keys:
        node += [k, offs[k]]
    node += [0] * (2 + 2 * cap - len(node))
    buf = np.array(words, dtype=np.uint16)
    buf[:node_words] = np.array(node, dtype=np.uint64).view(np.uint16)
    return buf.tobytes()


@pytest.mark.gpu
def test_gpu_sparse_bitmap_container():
    """TBitmapSparse [23]: array containers, bitmap containers (> 2048 values in a block, incl. a full block whose
    cardinality needs both header words), several blocks, the orphaned key-0 container; malformed payloads are refused"""
    import knoxdb_amd as K
    rng = np.random.default_rng(23)
    with K.Context(0) as ctx:
        cases = [
            (1 << 20, np.sort(rng.choice(1 << 20, 3000, replace=False))),                               # arrays in 16 blocks
            (300_000, np.unique(np.r_[np.arange(65536, 131072), rng.choice(300_000, 500), [0, 299_999]])),  # a FULL block: bitmap, cardinality 65536
            (200_001, np.unique(np.r_[rng.choice(np.arange(131072, 196608), 5000, replace=False), [7, 200_000]])),  # bitmap container, no key 0 values but one
            (70_000, np.array([69_999])),                                                                # a single value outside block 0
            (1000, np.array([], dtype=np.int64)),                                                        # empty
        ]
        for n, vals in cases:
            payload = sroar_from_sorted(vals)
            cont = bytes([23]) + O.put_uvarint(n) + O.put_uvarint(len(payload)) + payload
            want = np.zeros(n, dtype=bool)
            want[vals] = True
            b = ctx.bits(n).upload_bitmap_container(cont)
            assert (b.bytes(poison=32) == np.packbits(want, bitorder="little")).all(), n
            assert b.count() == len(vals)
            b.free()
        # a value beyond the bitset, a container offset beyond the buffer, an unknown container type
        n = 70_000
        for mutate in ("value", "offset", "type"):
            payload = bytearray(sroar_from_sorted(np.array([5, 66_000])))
            w = np.frombuffer(payload, dtype=np.uint16).copy()
            node_words = int(np.frombuffer(payload[:8], dtype=np.uint64)[0])
            last_off = int(np.frombuffer(payload[:2 * node_words], dtype=np.uint64)[2 + 2 * 1 + 1])   # key 65536's container
            if mutate == "value":
                w[last_off + 4] = 60_000                                                       # 65536 + 60000 >= 70000
            elif mutate == "offset":
                w[:node_words].view(np.uint64)[2 + 2 * 1 + 1] = len(w) + 10
            else:
                w[last_off + 1] = 7
            cont = bytes([23]) + O.put_uvarint(n) + O.put_uvarint(w.nbytes) + w.tobytes()
            with pytest.raises(K.KnoxError) as e:
                ctx.bits(n).upload_bitmap_container(cont)
            assert e.value.code in (-7, -1), mutate
