"""ctypes binding of the CPU oracle (oracle/knox_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- never by anything under knoxdb_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libknox_oracle.so")

# enums shared with the reference (internal/types/block.go:20-36, mode.go:13-27)
I64, I32, I16, I8, U64, U32, U16, U8, F64, F32, BYTES = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12
EQ, NE, GT, GE, LT, LE, IN, NIN, RG = 1, 2, 3, 4, 5, 6, 7, 8, 9
R_SUM, R_FIRST, R_LAST, R_MIN, R_MAX, R_MEAN, R_VAR, R_STD, R_COUNT = range(9)
R_FIRST_JOIN, R_LAST_JOIN, R_MIN_JOIN, R_MAX_JOIN, R_MEAN_JOIN, R_VAR_JOIN, R_STD_JOIN = range(9, 16)

NP = {I64: np.int64, I32: np.int32, I16: np.int16, I8: np.int8, U64: np.uint64, U32: np.uint32,
      U16: np.uint16, U8: np.uint8, F64: np.float64, F32: np.float32}
NAME = {"i64": I64, "i32": I32, "i16": I16, "i8": I8, "u64": U64, "u32": U32, "u16": U16, "u8": U8,
        "f64": F64, "f32": F32}
OPNAME = {"eq": EQ, "ne": NE, "gt": GT, "ge": GE, "lt": LT, "le": LE, "bw": RG}


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("knox_oracle.c", "knox_cpu_simd.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int
        L.ko_cmp.restype = i64
        L.ko_cmp.argtypes = [i32, i32, vp, i64, vp, vp, vp]
        for f in ("and", "andnot", "or", "xor"):
            getattr(L, "ko_bits_" + f).restype = None
            getattr(L, "ko_bits_" + f).argtypes = [vp, vp, i64]
        L.ko_bits_neg.restype = None
        L.ko_bits_neg.argtypes = [vp, i64]
        L.ko_bits_one.restype = None
        L.ko_bits_one.argtypes = [vp, i64]
        for f in ("and_flag", "or_flag"):
            getattr(L, "ko_bits_" + f).restype = i32
            getattr(L, "ko_bits_" + f).argtypes = [vp, vp, i64]
        L.ko_bits_popcount.restype = i64
        L.ko_bits_popcount.argtypes = [vp, i64]
        L.ko_bits_indexes.restype = i64
        L.ko_bits_indexes.argtypes = [vp, i64, vp]
        L.ko_reduce.restype = i32
        L.ko_reduce.argtypes = [i32, i32, vp, i64, vp, vp, C.POINTER(i64)]
        L.ko_exact_sum_f64.restype = i64
        L.ko_exact_sum_f64.argtypes = [vp, i64, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        for f, t in (("u64", u64), ("u32", C.c_uint32), ("u16", C.c_uint16), ("u8", C.c_uint8)):
            getattr(L, "ko_xxh3_" + f).restype = u64
            getattr(L, "ko_xxh3_" + f).argtypes = [t]
        L.ko_xxh3_bytes.restype = u64
        L.ko_xxh3_bytes.argtypes = [C.c_char_p, i64]
        L.ko_hash_value.restype = u64
        L.ko_hash_value.argtypes = [i32, vp]
        L.ko_bloom_pow2.restype = i64
        L.ko_bloom_pow2.argtypes = [i64]
        L.ko_bloom_add.restype = None
        L.ko_bloom_add.argtypes = [vp, i64, u64]
        L.ko_bloom_contains.restype = i32
        L.ko_bloom_contains.argtypes = [vp, i64, u64]
        L.ko_bytes_match.restype = i64
        L.ko_bytes_match.argtypes = [i32, vp, i64, vp, C.c_char_p, i64, vp, vp]
        L.ko_buckets.restype = i64
        L.ko_buckets.argtypes = [i32, i32, vp, vp, i64, vp, i64, i64, vp, vp, i64]
        L.ko_bool_match.restype = i32
        L.ko_bool_match.argtypes = [i32, i32, i32, vp, i64, vp]
        L.ko_inset.restype = i32
        L.ko_inset.argtypes = [i32, i32, vp, i64, vp, i64, vp, vp]
        L.ko_put_uvarint.restype = i32
        L.ko_put_uvarint.argtypes = [vp, u64]
        L.ko_uvarint.restype = i32
        L.ko_uvarint.argtypes = [vp, C.POINTER(u64)]
        L.ko_raw_container_store.restype = i64
        L.ko_raw_container_store.argtypes = [i32, vp, i64, vp]
        L.ko_raw_container_load.restype = i64
        L.ko_raw_container_load.argtypes = [i32, vp, i64, vp, i64]
        L.ko_bitpack_container_store.restype = i64
        L.ko_bitpack_container_store.argtypes = [i32, vp, i64, vp]
        L.ko_bitpack_container_load.restype = i64
        L.ko_bitpack_container_load.argtypes = [i32, vp, i64, vp, i64]
        L.ko_generate.restype = i32
        L.ko_generate.argtypes = [i32, i32, u64, i64, i64, i64, u64, vp]
        L.ko_simd_scan_i64_bw_sum.restype = i64
        L.ko_simd_scan_i64_bw_sum.argtypes = [vp, i64, i64, i64, i64, vp, i32, C.POINTER(i64)]
        L.ko_simd_has_avx2.restype = i32
        L.ko_simd_scan_i64_bw_sum_level.restype = i64
        L.ko_simd_scan_i64_bw_sum_level.argtypes = [vp, i64, i64, i64, i64, vp, i32, i32, C.POINTER(i64)]
        L.ko_simd_best_level.restype = i32
        L.ko_scan_i64_bw_sum.restype = i64
        L.ko_scan_i64_bw_sum.argtypes = [vp, i64, i64, i64, i64, vp, vp, C.POINTER(i64)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def bitfield_len(n):
    return (n + 7) >> 3


def cmp(dtype, op, src, a, b=None, poison=0):
    """Run a compare kernel; returns (bits ndarray[(n+7)/8], count).  With poison=k the buffer
    carries k trailing 0xfa bytes that are asserted untouched (cmp/tests/tests.go:24-48)."""
    src = np.ascontiguousarray(src, dtype=NP[dtype])
    n = src.size
    buf = np.zeros(bitfield_len(n) + poison, dtype=np.uint8)
    buf[bitfield_len(n):] = 0xFA
    av = np.array([a], dtype=NP[dtype])
    bv = np.array([b if b is not None else 0], dtype=NP[dtype])
    cnt = lib().ko_cmp(dtype, op, _p(src) if n else None, n, _p(av), _p(bv), _p(buf))
    assert (buf[bitfield_len(n):] == 0xFA).all(), "boundary violation"
    return buf[:bitfield_len(n)], cnt


def bits_op(name, dst, src, size):
    dst = np.array(dst, dtype=np.uint8, copy=True)
    src = np.ascontiguousarray(src, dtype=np.uint8)
    getattr(lib(), "ko_bits_" + name)(_p(dst), _p(src), size)
    return dst


def bits_flag_op(name, dst, src, size):
    dst = np.array(dst, dtype=np.uint8, copy=True)
    src = np.ascontiguousarray(src, dtype=np.uint8)
    f = getattr(lib(), "ko_bits_" + name + "_flag")(_p(dst), _p(src), size)
    return dst, bool(f & 1), bool(f & 2)


def bits_neg(src, size):
    d = np.array(src, dtype=np.uint8, copy=True)
    lib().ko_bits_neg(_p(d), size)
    return d


def bits_one(size):
    d = np.zeros(bitfield_len(size), dtype=np.uint8)
    lib().ko_bits_one(_p(d), size)
    return d


def popcount(buf, size):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return lib().ko_bits_popcount(_p(buf) if buf.size else None, size)


def indexes(buf, size):
    buf = np.array(buf, dtype=np.uint8, copy=True)
    out = np.zeros(size + 8, dtype=np.uint32)
    k = lib().ko_bits_indexes(_p(buf) if buf.size else None, size, _p(out))
    return out[:k].copy()


def reduce(dtype, func, src, bits=None):
    """Sequential reducer over selected rows; returns (value, seen, row)."""
    src = np.ascontiguousarray(src, dtype=NP[dtype])
    out = np.zeros(1, dtype=NP[dtype])
    row = C.c_int64(-1)
    b = np.ascontiguousarray(bits, dtype=np.uint8) if bits is not None else None
    seen = lib().ko_reduce(dtype, func, _p(src), src.size, _p(b), _p(out), C.byref(row))
    return out[0], bool(seen), row.value


def exact_sum_f64(src, bits=None):
    src = np.ascontiguousarray(src, dtype=np.float64)
    s, m = C.c_double(), C.c_double()
    b = np.ascontiguousarray(bits, dtype=np.uint8) if bits is not None else None
    cnt = lib().ko_exact_sum_f64(_p(src), src.size, _p(b), C.byref(s), C.byref(m))
    return s.value, m.value, cnt


def xxh3_bytes(data):
    return lib().ko_xxh3_bytes(bytes(data), len(data))


def hash_value(dtype, val):
    v = np.array([val], dtype=NP[dtype])
    return lib().ko_hash_value(dtype, _p(v))


def bloom_new(cardinality, factor):
    """bloom.NewFilter(cardinality*factor*8) -> serialized [k=4][m/8 zero bytes]"""
    m = lib().ko_bloom_pow2(cardinality * factor * 8)
    buf = np.zeros(1 + (m >> 3), dtype=np.uint8)
    buf[0] = 4
    return buf


def bloom_add(buf, h):
    lib().ko_bloom_add(_p(buf), buf.size, h)


def bloom_contains(buf, h):
    return bool(lib().ko_bloom_contains(_p(buf), buf.size, h))


def bytes_match(op, ptr, buf, val, mask=None, bits=None, val2=None):
    """StringPool.Match* (EQ/NE/LT/LE/GT/GE, RANGE = [val, val2])"""
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n = ptr.size
    out = np.zeros(bitfield_len(n), dtype=np.uint8) if bits is None else np.array(bits, dtype=np.uint8, copy=True)
    m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    v2 = bytes(val2) if val2 is not None else b""
    f = lib().ko_bytes_match2
    f.restype = C.c_int64
    cnt = f(int(op), _p(ptr), C.c_int64(n), _p(buf), bytes(val), C.c_int64(len(val)), v2, C.c_int64(len(v2)), _p(out), _p(m))
    assert cnt != -2, "unsupported bytes mode"
    return out, cnt


def bytes_inset(negate, ptr, buf, values, mask=None, bits=None):
    """bytes IN / NOT IN: membership of each row in `values` (list of bytes)"""
    ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n = ptr.size
    vals = [bytes(v) for v in values]
    lens = np.array([len(v) for v in vals], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if vals else np.zeros(0, np.uint64)
    sp = (offs << np.uint64(32)) | lens
    sb = np.frombuffer(b"".join(vals) + b"\x00", dtype=np.uint8)
    out = np.zeros(bitfield_len(n), dtype=np.uint8) if bits is None else np.array(bits, dtype=np.uint8, copy=True)
    m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    assert lib().ko_bytes_inset(int(bool(negate)), _p(ptr), C.c_int64(n), _p(buf), _p(sp) if len(vals) else None, C.c_int64(len(vals)), _p(sb),
                                _p(out), _p(m)) == 0
    return out


def buckets(dtype, func, times, vals, bits, t_from, step, cap=None):
    """NativeBucket.Push over the selected rows -> (window times, values) per reducer in list order."""
    times = np.ascontiguousarray(times, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=NP[dtype])
    cap = cap or (times.size + 1)
    out_t = np.zeros(cap, dtype=np.int64)
    out_v = np.zeros(cap, dtype=NP[dtype])
    b = np.ascontiguousarray(bits, dtype=np.uint8) if bits is not None else None
    k = lib().ko_buckets(dtype, func, _p(times), _p(vals), times.size, _p(b), t_from, step, _p(out_t), _p(out_v), cap)
    assert k >= 0, k
    return out_t[:k].copy(), out_v[:k].copy()


def bool_match(op, a, b, src, size):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.zeros(bitfield_len(size), dtype=np.uint8)
    assert lib().ko_bool_match(op, int(bool(a)), int(bool(b)), _p(src), size, _p(out)) == 0
    return out


def cmp_wide(op, limbs, a, b=None, mask=None):
    """Int128 / Int256 compare over limb slices (most significant first) -> (bits, count)."""
    limbs = [np.ascontiguousarray(x).view(np.uint64) for x in limbs]
    n = limbs[0].size
    arr = (C.c_void_p * len(limbs))(*[x.ctypes.data for x in limbs])
    av = np.array([int(v) & (2**64 - 1) for v in a], dtype=np.uint64)
    bv = np.array([int(v) & (2**64 - 1) for v in (b if b is not None else a)], dtype=np.uint64)
    out = np.zeros(bitfield_len(n) + 32, dtype=np.uint8)
    out[bitfield_len(n):] = 0xFA  # poison tail (internal/cmp/tests/tests.go:24-48)
    m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    f = lib().ko_cmp_wide
    f.restype = C.c_int64
    cnt = f(len(limbs), int(op), arr, C.c_int64(n), _p(av), _p(bv), _p(out), _p(m))
    assert cnt >= 0
    assert (out[bitfield_len(n):] == 0xFA).all(), "oracle wrote past the bitset"
    return out[:bitfield_len(n)].copy(), int(cnt)


def range_vectors_set(dtype, mins, maxs, values, negate=False, mask=None):
    """numInSetMatcher / numNotInSetMatcher.MatchRangeVectors on materialized statistics vectors
    (internal/operator/filter/match_num.go:693-727, :810-817): row i may match when the set holds a member in
    [uint64(min[i]), uint64(max[i])] -- Go's conversion sign-extends a signed T; the pair is swapped when reversed --,
    masked rows only; NOT IN is undecided: the mask, or all ones.  Pure numpy (xroar's ContainsRange = a sorted search)."""
    mins = np.ascontiguousarray(mins, dtype=NP[dtype])
    maxs = np.ascontiguousarray(maxs, dtype=NP[dtype])
    n = mins.size
    m = np.unpackbits(np.ascontiguousarray(mask, dtype=np.uint8), bitorder="little")[:n].astype(bool) if mask is not None else np.ones(n, dtype=bool)
    if negate:
        return np.packbits(m, bitorder="little")
    u64 = (lambda x: x.astype(np.int64).view(np.uint64)) if np.dtype(NP[dtype]).kind == "i" else (lambda x: x.astype(np.uint64))
    lo, hi = u64(mins), u64(maxs)
    lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)
    keys = np.unique(u64(np.ascontiguousarray(values, dtype=NP[dtype])))
    idx = np.searchsorted(keys, lo, side="left")
    hit = (idx < keys.size) & (keys[np.minimum(idx, max(keys.size - 1, 0))] <= hi) if keys.size else np.zeros(n, dtype=bool)
    return np.packbits(hit & m, bitorder="little")


def range_vectors_wide(op, mins, maxs, a, b=None, mask=None):
    """The Int128 / Int256 matchers' MatchRangeVectors (match_i128.go:78-375, match_i256.go:64-229) over limb slices
    (most significant first): EQ :78-91 = le(min) with the mask, and-ed with the mask, ge(max) under it, and-ed again; RANGE
    :299-313 the same with (to, from); LT / LE :213-218,:243-248 on min, GT / GE :153-158,:183-188 on max (the wide compares
    honour the mask); NE / NOT IN :111-118,:432-439 the mask or all ones."""
    n = np.ascontiguousarray(mins[0]).size
    full = np.packbits(np.ones(n, dtype=bool), bitorder="little")
    if op in (NE, NIN):
        return np.array(mask, dtype=np.uint8, copy=True) if mask is not None else full
    if op in (LT, LE):
        return cmp_wide(op, mins, a, mask=mask)[0]
    if op in (GT, GE):
        return cmp_wide(op, maxs, a, mask=mask)[0]
    assert op in (EQ, RG)
    min_bits = cmp_wide(LE, mins, b if op == RG else a, mask=mask)[0]
    if mask is not None:
        min_bits &= np.ascontiguousarray(mask, dtype=np.uint8)
    bits = cmp_wide(GE, maxs, a, mask=min_bits)[0]
    return bits & min_bits


def inset(dtype, negate, src, values, mask=None, bits=None):
    src = np.ascontiguousarray(src, dtype=NP[dtype])
    vals = np.ascontiguousarray(values, dtype=NP[dtype])
    out = np.zeros(bitfield_len(src.size), dtype=np.uint8) if bits is None else np.array(bits, dtype=np.uint8, copy=True)
    m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    rc = lib().ko_inset(dtype, int(bool(negate)), _p(src), src.size, _p(vals) if vals.size else None, vals.size, _p(out), _p(m))
    assert rc == 0
    return out


def put_uvarint(x):
    b = np.zeros(9, dtype=np.uint8)
    n = lib().ko_put_uvarint(_p(b), x)
    return b[:n].tobytes()


def uvarint(buf):
    b = np.frombuffer(bytes(buf) + b"\0" * 9, dtype=np.uint8).copy()
    v = C.c_uint64()
    n = lib().ko_uvarint(_p(b), C.byref(v))
    return v.value, n


def raw_container_store(dtype, values):
    values = np.ascontiguousarray(values, dtype=NP[dtype])
    out = np.zeros(values.nbytes + 16, dtype=np.uint8)
    n = lib().ko_raw_container_store(dtype, _p(values) if values.size else None, values.size, _p(out))
    assert n > 0
    return out[:n].tobytes()


def raw_container_load(dtype, buf):
    b = np.frombuffer(bytes(buf), dtype=np.uint8).copy()
    out = np.zeros(len(buf) // np.dtype(NP[dtype]).itemsize + 1, dtype=NP[dtype])
    n = lib().ko_raw_container_load(dtype, _p(b), b.size, _p(out), out.size)
    assert n >= 0, n
    return out[:n].copy()


def bitpack_container_store(dtype, values):
    values = np.ascontiguousarray(values, dtype=NP[dtype])
    out = np.zeros(values.size * 8 + 64, dtype=np.uint8)
    n = lib().ko_bitpack_container_store(dtype, _p(values), values.size, _p(out))
    assert n > 0, n
    return out[:n].tobytes()


def bitpack_container_load(dtype, buf, n_max):
    b = np.frombuffer(bytes(buf), dtype=np.uint8).copy()
    out = np.zeros(n_max, dtype=NP[dtype])
    n = lib().ko_bitpack_container_load(dtype, _p(b), b.size, _p(out), n_max)
    assert n >= 0, n
    return out[:n].copy()


def generate(dtype, kind, seed, first_row, n, lo=0, span=0):
    out = np.zeros(n, dtype=NP[dtype])
    rc = lib().ko_generate(dtype, kind, seed, first_row, n, lo, span, _p(out))
    assert rc == 0
    return out


def generate_mt(dtype, kind, seed, first_row, n, lo=0, span=0, threads=None, out=None):
    """generate() over several host threads (same values)"""
    import os
    threads = threads or len(os.sched_getaffinity(0))
    out = np.empty(n, dtype=NP[dtype]) if out is None else out
    L = lib()
    L.ko_generate_mt.restype = C.c_int
    L.ko_generate_mt.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_int]
    rc = L.ko_generate_mt(dtype, kind, seed, first_row, n, lo, span, _p(out), threads)
    assert rc == 0
    return out


def scan_i64_bw_sum(src, pack_rows, a, b):
    src = np.ascontiguousarray(src, dtype=np.int64)
    n = src.size
    bits = np.zeros(bitfield_len(n), dtype=np.uint8)
    sel = np.zeros(pack_rows + 8, dtype=np.uint32)
    s = C.c_int64()
    cnt = lib().ko_scan_i64_bw_sum(_p(src), n, pack_rows, a, b, _p(bits), _p(sel), C.byref(s))
    return bits, cnt, s.value


SIMD_LEVELS = {1: "scalar", 2: "AVX2", 3: "AVX-512"}


def simd_best_level():
    """1 = scalar, 2 = AVX2, 3 = AVX-512: the best form of the SIMD baseline this CPU runs"""
    return int(lib().ko_simd_best_level())


def simd_scan_i64_bw_sum(src, pack_rows, a, b, threads, bits=None, level=2):
    """SIMD pack-parallel CPU baseline (knox_cpu_simd.c; level 2 = AVX2, 3 = AVX-512, 0 = the best available): returns
    (bits, count, sum)."""
    src = np.ascontiguousarray(src, dtype=np.int64)
    if bits is None:
        bits = np.zeros(bitfield_len(src.size), dtype=np.uint8)
    s = C.c_int64()
    cnt = lib().ko_simd_scan_i64_bw_sum_level(_p(src), src.size, pack_rows, a, b, _p(bits), threads, level, C.byref(s))
    return bits, cnt, s.value


def emit_fill(dtype, mode, reducers, t_from, t_to, step, units_per_second=1, limit=None):
    """NativeBucket.Emit (internal/reducer/bucket_native.go:161-218) with Fill / linearFill (fill.go:56-85), minus
    the string formatting: reducers = [(window time, value)] sorted by time -> [(window time, value | None)] in
    emission order (None = the JSON null of FillModeNull).  Pure Python, small cases only."""
    T = NP[dtype]
    is_float = np.dtype(T).kind == "f"
    out = []
    if not reducers:
        return out
    limit = limit if limit is not None else 1 << 62
    last, idx = reducers[0], 0

    def unix(t):
        return t // units_per_second          # time.Time.Unix(): floored seconds

    step_t = t_from
    while step_t <= t_to and len(out) < limit:
        nxt = reducers[idx] if idx < len(reducers) else last
        if nxt[0] != step_t:
            if mode == "null":
                out.append((step_t, None))
            elif mode == "last":
                out.append((step_t, T(last[1])))
            elif mode == "zero":
                out.append((step_t, T(0)))
            elif mode == "linear":
                with np.errstate(all="ignore"):
                    diff = T(nxt[1]) - T(last[1])          # in T: wraps for integers
                    den = np.float64(unix(nxt[0]) - unix(last[0]))
                    m = np.float64(diff) / den             # 0/0 -> NaN, x/0 -> Inf as in Go
                    x = np.float64(unix(step_t) - unix(last[0]))
                    y = m * x + np.float64(last[1])
                    if is_float:
                        out.append((step_t, T(y)))
                    else:
                        info = np.iinfo(T)
                        ok = np.isfinite(y) and info.min <= np.trunc(y) <= info.max and abs(y) < 2.0 ** 63
                        out.append((step_t, T(np.trunc(y)) if ok else "platform-defined"))
            step_t += step
            continue
        out.append((step_t, T(nxt[1])))
        idx += 1
        last = nxt
        step_t += step
    return out


# ---- synthetic bytes column (DESIGN.md "synthetic data", kind 5; SURVEY.md 8d cfg 4) -- numpy restatement of the
# spec text in include/knoxgpu.h (knox_col_generate_bytes), independent of the device kernel
def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def bytes_key_index(seed, first_row, n, pack_rows, window_bits=16, packs_per_window=4):
    """key index of rows [first_row, first_row + n)"""
    rows = np.arange(first_row, first_row + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        r = mix64(rows + mix64(np.uint64(seed)))
    pack = rows // np.uint64(pack_rows)
    return ((pack // np.uint64(packs_per_window)) << np.uint64(window_bits)) | (r & np.uint64((1 << window_bits) - 1))


def bytes_key(seed, k, key_len=32):
    """the key_len bytes of key k"""
    nw = (key_len + 7) // 8
    with np.errstate(over="ignore"):
        km = mix64(np.uint64(seed ^ 0x6B657973))
        w = mix64(np.uint64(8) * np.asarray(k, dtype=np.uint64)[..., None] + np.arange(nw, dtype=np.uint64) + km)
    b = np.ascontiguousarray(w.astype("<u8")).view(np.uint8).reshape(w.shape[:-1] + (nw * 8,))
    return b[..., :key_len]


def generate_bytes_pack(seed, pack, n, pack_rows, key_len=32, window_bits=16, packs_per_window=4):
    """pack `pack` (n rows) of the synthetic bytes column as one StringPool: (ptr, buf, key indices)"""
    k = bytes_key_index(seed, pack * pack_rows, n, pack_rows, window_bits, packs_per_window)
    buf = np.ascontiguousarray(bytes_key(seed, k, key_len)).reshape(-1)
    ptr = ((np.arange(n, dtype=np.uint64) * np.uint64(key_len)) << np.uint64(32)) | np.uint64(key_len)
    return ptr, buf, k
