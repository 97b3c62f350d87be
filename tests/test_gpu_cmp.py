"""GPU parity for the compare kernels, called through the C ABI (knox_cmp):
the reference's golden vectors at every tail length + seeded random data vs the oracle.
Mirrors internal/cmp/tests/tests.go:24-48 (count, exact bytes, 32 poison bytes intact)."""
import json
import os

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd.types import NP_DTYPE, BlockType, FilterMode
from oracle import oracle as O
from helpers import rand_col

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
BT = {"i64": BlockType.Int64, "i32": BlockType.Int32, "i16": BlockType.Int16, "i8": BlockType.Int8,
      "u64": BlockType.Uint64, "u32": BlockType.Uint32, "u16": BlockType.Uint16, "u8": BlockType.Uint8,
      "f64": BlockType.Float64, "f32": BlockType.Float32}
OPS = [FilterMode.Equal, FilterMode.NotEqual, FilterMode.Lt, FilterMode.Le, FilterMode.Gt, FilterMode.Ge, FilterMode.Range]


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def _dec(c, key):
    if c["dtype"] == "f64":
        return np.array(c[key] if isinstance(c[key], list) else [c[key]], dtype=np.uint64).view(np.float64)
    if c["dtype"] == "f32":
        return np.array(c[key] if isinstance(c[key], list) else [c[key]], dtype=np.uint32).view(np.float32)
    return np.array(c[key] if isinstance(c[key], list) else [c[key]], dtype=NP_DTYPE[BT[c["dtype"]]])


@pytest.mark.parametrize("dtype", sorted(BT))
def test_cmp_golden_vectors(ctx, dtype):
    cases = [c for c in json.load(open(os.path.join(G, "cmp_cases.json")))["cases"] if c["dtype"] == dtype]
    assert len(cases) > 100
    for c in cases:
        bits, cnt = ctx.cmp(BT[dtype], O.OPNAME[c["op"]], _dec(c, "src"), _dec(c, "a")[0], _dec(c, "b")[0], poison=32)
        tag = "%s/%s/%s" % (dtype, c["op"], c["name"])
        assert cnt == c["count"], tag
        assert bits.tobytes().hex() == c["result"], tag


SIZES = [1, 7, 8, 9, 63, 64, 65, 127, 1000, 2047, 2048, 2049, 4096, 5000, 16384 + 3, 65536, 3 * 65536 + 1234]


@pytest.mark.parametrize("dtype", sorted(BT))
def test_cmp_random_vs_oracle(ctx, dtype):
    rng = np.random.default_rng(hash(dtype) % 1000)
    bt = BT[dtype]
    for n in SIZES:
        x = rand_col(bt, n, rng)
        for op in OPS:
            picks = [x[rng.integers(0, n)], x[rng.integers(0, n)]]
            a, b = picks[0], picks[1]
            if op == FilterMode.Range and x.dtype.kind != "f" and a > b and rng.random() < 0.7:
                a, b = b, a  # keep some a > b cases: the wrap-around set (number.go:211-243)
            want_bits, want_cnt = O.cmp(int(bt), int(op), x, a, b)
            got_bits, got_cnt = ctx.cmp(bt, op, x, a, b, poison=32)
            assert got_cnt == want_cnt, (dtype, n, op, a, b)
            assert (got_bits == want_bits).all(), (dtype, n, op, a, b)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_cmp_float_special_operands(ctx, dtype):
    bt = BT[dtype]
    npdt = NP_DTYPE[bt]
    info = np.finfo(npdt)
    rng = np.random.default_rng(9)
    x = rand_col(bt, 4099, rng)
    vals = [np.nan, np.inf, -np.inf, 0.0, -0.0, info.tiny, info.smallest_subnormal, -info.smallest_subnormal,
            info.max, info.min, 5.0]
    for op in OPS:
        for a in vals:
            for b in (vals if op == FilterMode.Range else [0.0]):
                want_bits, want_cnt = O.cmp(int(bt), int(op), x, npdt(a), npdt(b))
                got_bits, got_cnt = ctx.cmp(bt, op, x, npdt(a), npdt(b))
                assert got_cnt == want_cnt and (got_bits == want_bits).all(), (dtype, op, a, b)


@pytest.mark.parametrize("dtype", ["i64", "u64", "i8", "u16"])
def test_cmp_int_extreme_operands(ctx, dtype):
    bt = BT[dtype]
    npdt = NP_DTYPE[bt]
    info = np.iinfo(npdt)
    rng = np.random.default_rng(10)
    x = rand_col(bt, 3001, rng)
    vals = [info.min, info.max, 0, 5, info.min + 1, info.max - 1]
    for op in OPS:
        for a in vals:
            for b in (vals if op == FilterMode.Range else [0]):
                want_bits, want_cnt = O.cmp(int(bt), int(op), x, npdt(a), npdt(b))
                got_bits, got_cnt = ctx.cmp(bt, op, x, npdt(a), npdt(b))
                assert got_cnt == want_cnt and (got_bits == want_bits).all(), (dtype, op, a, b)


def test_cmp_empty_and_nil(ctx):
    # cmp/tests/int64.go:212-213: l0 / nil -> count 0, nothing written
    bits, cnt = ctx.cmp(BlockType.Int64, FilterMode.Equal, np.zeros(0, np.int64), 5)
    assert cnt == 0 and bits.size == 0
