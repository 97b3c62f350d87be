"""GPU parity for the Int128 / Int256 matchers (knox_match_wide) through the C ABI: the
reference's own golden vectors (internal/cmp/tests/int128.go, int256.go via
tests/golden/cmp_wide_cases.json), random data against the oracle, masks, poison tail."""
import json
import os

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd.types import FilterMode
from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
WIDE = json.load(open(os.path.join(G, "cmp_wide_cases.json")))["cases"]
MODE = {"eq": FilterMode.Equal, "ne": FilterMode.NotEqual, "lt": FilterMode.Lt, "le": FilterMode.Le,
        "gt": FilterMode.Gt, "ge": FilterMode.Ge, "bw": FilterMode.Range}


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def upload_limbs(ctx, limbs):
    # X0 is []int64, the rest []uint64 (num.Int128Stride)
    return [ctx.upload(l.view(np.int64) if i == 0 else l.view(np.uint64), 0) for i, l in enumerate(limbs)]


@pytest.mark.parametrize("dtype", ["i128", "i256"])
def test_wide_golden_vectors(ctx, dtype):
    nl = 2 if dtype == "i128" else 4
    ran = 0
    for c in WIDE:
        if c["dtype"] != dtype or c["n"] == 0:
            continue
        src = np.array(c["src"], dtype=np.uint64).reshape(-1, nl)
        cols = upload_limbs(ctx, [np.ascontiguousarray(src[:, l]) for l in range(nl)])
        bits, cnt = ctx.match_wide(cols, MODE[c["op"]], c["a"], c["b"] if c["op"] == "bw" else None)
        got = bits.bytes(poison=32)  # asserts the 32 poison bytes behind the bitset are untouched
        assert got.tobytes().hex() == c["result"], (dtype, c["op"], c["name"])
        assert cnt == c["count"], (dtype, c["op"], c["name"])
        for o in cols:
            o.free()
        bits.free()
        ran += 1
    assert ran == 105  # 119 minus the empty l0 / nil cases


@pytest.mark.parametrize("nl", [2, 4])
def test_wide_random_vs_oracle(ctx, nl):
    rng = np.random.default_rng(40 + nl)
    for n in (1, 7, 2047, 2048, 2049, 10000, 3 * 2048 + 5):
        # few distinct values per limb so that every limb decides some comparisons
        limbs = [rng.integers(0, 3, size=n).astype(np.uint64) for _ in range(nl)]
        limbs[0] = (limbs[0].astype(np.int64) - 1).view(np.uint64)  # signed top limb: -1, 0, 1
        limbs[-1] = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64) * 2 + rng.integers(0, 2, size=n).astype(np.uint64)
        cols = upload_limbs(ctx, limbs)
        a = [int(l[n // 2]) for l in limbs]
        b = [int(l[n // 3]) for l in limbs]
        lo, hi = (a, b) if O.cmp_wide(O.OPNAME["le"], [np.array([x], dtype=np.uint64) for x in a], b)[1] else (b, a)
        mask_h = rng.integers(0, 256, size=O.bitfield_len(n)).astype(np.uint8)
        mask = ctx.bits(n).upload(mask_h)
        for op in ("eq", "ne", "lt", "le", "gt", "ge", "bw"):
            want, wcnt = O.cmp_wide(O.OPNAME[op], limbs, lo, hi)
            bits, cnt = ctx.match_wide(cols, MODE[op], lo, hi if op == "bw" else None)
            assert (bits.bytes(poison=32) == want).all(), (nl, n, op)
            assert cnt == wcnt
            # with a mask: hits & mask (count = popcount of the result, see knoxgpu.h)
            wm, _ = O.cmp_wide(O.OPNAME[op], limbs, lo, hi, mask=mask_h)
            bm, cm = ctx.match_wide(cols, MODE[op], lo, hi if op == "bw" else None, mask=mask)
            assert (bm.bytes(poison=32) == wm).all(), (nl, n, op, "mask")
            assert cm == O.popcount(wm, n)
            bits.free()
            bm.free()
        for o in cols:
            o.free()
        mask.free()


def test_wide_extremes_and_errors(ctx):
    M = 2**64 - 1
    vals = [[1 << 63, 0], [M, M], [0, 0], [0, 1], [0x7FFFFFFFFFFFFFFF, M]]  # MinInt128, -1, 0, 1, MaxInt128
    src = np.array(vals * 8, dtype=np.uint64)
    cols = upload_limbs(ctx, [np.ascontiguousarray(src[:, 0]), np.ascontiguousarray(src[:, 1])])
    bits, cnt = ctx.match_wide(cols, FilterMode.Lt, [0, 0])
    assert cnt == 16  # MinInt128 and -1
    bits, cnt = ctx.match_wide(cols, FilterMode.Range, [M, M], [0, 1])
    assert cnt == 24  # -1, 0, 1
    bits, cnt = ctx.match_wide(cols, FilterMode.Ge, [1 << 63, 0])
    assert cnt == 40
    with pytest.raises(K.KnoxError):  # three limbs
        ctx.match_wide(cols + [cols[0]], FilterMode.Lt, [0, 0, 0])
    f = ctx.upload(np.zeros(40, dtype=np.float64), 0)
    with pytest.raises(K.KnoxError):  # float limb
        ctx.match_wide([cols[0], f], FilterMode.Lt, [0, 0])
    with pytest.raises(K.KnoxError):  # IN is not a wide mode
        ctx.match_wide(cols, FilterMode.In, [0, 0])


@pytest.mark.parametrize("form", ["", "general"])
@pytest.mark.parametrize("nl", [2, 4])
def test_wide_in_not_in(ctx, nl, form, monkeypatch):
    """i128 / i256 InSet / NotInSet matchers (match_i128.go:318-430): membership, bits.Set semantics, mask -- through the prefix
    search in LDS (the default up to 4096 members) and through the full-compare binary search (KNOX_WIDE_IN_FORM=general)"""
    if form:
        monkeypatch.setenv("KNOX_WIDE_IN_FORM", form)
    else:
        monkeypatch.delenv("KNOX_WIDE_IN_FORM", raising=False)
    rng = np.random.default_rng(60 + nl)
    n = 5000 + 3
    limbs = [rng.integers(0, 3, size=n).astype(np.uint64) for _ in range(nl)]
    limbs[0] = (limbs[0].astype(np.int64) - 1).view(np.uint64)  # signed top limb: -1, 0, 1
    cols = upload_limbs(ctx, limbs)
    rows = [tuple(int(l[i]) for l in limbs) for i in range(n)]
    sets = [[rows[5], rows[77], rows[5], tuple([2**64 - 1] * nl)], [], [rows[i] for i in range(0, 200)], [tuple([7] * nl)]]
    mask_h = rng.integers(0, 256, size=O.bitfield_len(n)).astype(np.uint8)
    mask_h[-1] &= 0xFF >> (8 - n % 8)
    mask = ctx.bits(n).upload(mask_h)
    for st in sets:
        member = np.array([r in set(st) for r in rows])
        for negate in (False, True):
            want = np.packbits(member != negate, bitorder="little")
            got = ctx.match_wide_set(cols, st, negate=negate)
            assert (got.bytes(poison=32) == want).all(), (nl, len(st), negate)
            gm = ctx.match_wide_set(cols, st, negate=negate, mask=mask)
            assert (gm.bytes() == (want & mask_h)).all()
            got.free()
            gm.free()
    for o in cols:
        o.free()
    mask.free()
