"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol
include/knoxgpu.h declares, fails loudly without a device, and the pure-host entry points
(knox_agg_value / knox_agg_combine / knox_hash_value) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import _lib
from knoxdb_amd._lib import AggState
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "knoxgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(knox_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 50
    L = C.CDLL(_lib.SO_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    want = int(re.search(r"#define KNOX_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "knoxgpu.h")).read()).group(1))
    assert _lib.lib().knox_abi_version() == want


def test_no_silent_cpu_fallback():
    if K.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(K.KnoxError) as e:
        K.Context(0)
    assert e.value.code == -2  # KNOX_E_NODEVICE


def test_strerror():
    L = _lib.lib()
    assert L.knox_strerror(0) == b"ok"
    assert b"device" in L.knox_strerror(-2)


def test_hash_value_matches_reference_hash():
    rng = np.random.default_rng(5)
    for v in rng.integers(0, 2**63, size=50, dtype=np.uint64):
        assert K.hash_value(K.BlockType.Uint64, v) == O.hash_value(O.U64, v)
        assert K.hash_value(K.BlockType.Int64, np.int64(v)) == O.hash_value(O.I64, np.int64(v))
    for v in rng.integers(0, 2**32, size=50, dtype=np.uint64):
        assert K.hash_value(K.BlockType.Uint32, v) == O.hash_value(O.U32, v)
    for v in (0, 1, 0x7FFF, 0xFFFF):
        assert K.hash_value(K.BlockType.Uint16, v) == O.hash_value(O.U16, v)
    for v in (0, 1, 0x7F, 0xFF):
        assert K.hash_value(K.BlockType.Uint8, v) == O.hash_value(O.U8, v)
    for v in (0.0, -0.0, 1.5, float("inf"), 1e-310):
        assert K.hash_value(K.BlockType.Float64, v) == O.hash_value(O.F64, v)
        assert K.hash_value(K.BlockType.Float32, v) == O.hash_value(O.F32, v)


def _state_from_oracle(dtype_o, dtype_k, x, bits, row_base):
    """what the device writes for one shard, computed with the oracle (host-side merge test)"""
    st = AggState()
    st.dtype, st.flags = int(dtype_k), 1
    sel = np.unpackbits(bits, bitorder="little")[:x.size].astype(bool) if bits is not None else np.ones(x.size, bool)
    st.count = int(sel.sum())
    raw = x.view({8: np.uint64, 4: np.uint32, 2: np.uint16, 1: np.uint8}[x.dtype.itemsize])
    st.isum = int(raw[sel].astype(np.uint64).sum(dtype=np.uint64)) if st.count else 0
    xs = x[sel].astype(np.float64)
    hi = float(np.sum(xs)) if st.count else 0.0
    s, _, _ = O.exact_sum_f64(xs)
    st.fsum_hi, st.fsum_lo = (s, 0.0)
    for name, func in (("min", O.R_MIN), ("max", O.R_MAX), ("first", O.R_FIRST), ("last", O.R_LAST)):
        if name in ("min", "max") and x.dtype.kind == "f":
            # the device never adopts a NaN: emulate by masking NaNs out
            m = sel & ~np.isnan(x)
            b2 = np.packbits(m, bitorder="little")
            v, seen, row = O.reduce(dtype_o, func, x, b2)
        else:
            v, seen, row = O.reduce(dtype_o, func, x, bits)
        setattr(st, name + "_row", row + row_base if seen else -1)
        setattr(st, name + "_bits", int(np.array([v], dtype=x.dtype).view(raw.dtype)[0]) if seen else 0)
    return st


@pytest.mark.parametrize("case", ["i64", "f64", "u8"])
def test_agg_combine_and_value_match_oracle(case):
    rng = np.random.default_rng(3)
    dt_o, dt_k, npdt = {"i64": (O.I64, K.BlockType.Int64, np.int64), "f64": (O.F64, K.BlockType.Float64, np.float64),
                        "u8": (O.U8, K.BlockType.Uint8, np.uint8)}[case]
    n, shards = 4000, 4
    if case == "f64":
        x = (rng.integers(0, 1 << 20, n) / 1024.0).astype(npdt)  # exact-sum dataset
        x[17] = np.nan
        x[300] = -0.0
    else:
        info = np.iinfo(npdt)
        x = rng.integers(info.min, info.max, n, endpoint=True).astype(npdt)
    bits = np.packbits(rng.random(n) < 0.3, bitorder="little")
    per = n // shards
    states = []
    for s in range(shards):
        sl = slice(s * per, (s + 1) * per)
        states.append(_state_from_oracle(dt_o, dt_k, x[sl], bits[s * per // 8:(s + 1) * per // 8], s * per))
    merged = K.agg_combine(states)
    for func_k, func_o in ((K.ReducerFunc.Count, O.R_COUNT), (K.ReducerFunc.Sum, O.R_SUM), (K.ReducerFunc.Min, O.R_MIN),
                           (K.ReducerFunc.Max, O.R_MAX), (K.ReducerFunc.First, O.R_FIRST), (K.ReducerFunc.Last, O.R_LAST)):
        got = K.agg_value(merged, func_k)
        want = O.reduce(dt_o, func_o, x, bits)
        if case == "f64" and func_k == K.ReducerFunc.Sum:
            # exact dataset except the NaN row (may or may not be selected)
            assert (np.isnan(got[0]) and np.isnan(want[0])) or got[0] == want[0]
        else:
            assert got[0] == want[0] or (np.isnan(got[0]) and np.isnan(want[0])), (func_k, got, want)
        assert got[1] == want[1]
        if func_k in (K.ReducerFunc.Min, K.ReducerFunc.Max, K.ReducerFunc.First, K.ReducerFunc.Last):
            assert got[2] == want[2], (func_k, got, want)


def test_min_max_first_value_nan_is_sticky():
    x = np.array([np.nan, 3.0, -1.0, 7.0])
    st = _state_from_oracle(O.F64, K.BlockType.Float64, x, None, 0)
    for f_k, f_o in ((K.ReducerFunc.Min, O.R_MIN), (K.ReducerFunc.Max, O.R_MAX)):
        got, want = K.agg_value(st, f_k), O.reduce(O.F64, f_o, x)
        assert np.isnan(got[0]) and np.isnan(want[0]) and got[2] == want[2] == 0


def test_host_hash_bytes_matches_xxh3_golden_vectors():
    """knox_hash_bytes (host side of the library: pure C++, no device needed) against the 271
    python-xxhash digests in tests/golden/xxh3_bytes.json (lengths 0..1040: every XXH3 branch) and
    the reference's own fixed-width KATs (internal/hash/xxh3_test.go:14-31 via xxh3_kat.json)"""
    import json
    import knoxdb_amd as K
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xxh3_bytes.json")))
    for n, want in g["digests"].items():
        data = bytes((i * 131 + 7) & 0xFF for i in range(int(n)))
        assert K.hash_bytes(data) == want, n
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xxh3_kat.json")))
    for i in range(8):
        data = bytes(range(i, i + 8))
        assert K.hash_bytes(data) == kat["u64"][i] and K.hash_bytes(data[:4]) == kat["u32"][i]
