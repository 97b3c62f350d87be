"""BASELINE config 0/1 plumbing: one 2^16-row int64 pack, cmp.Int64Less + bitset popcount.
CPU leg: the oracle (the reference's generic Go path restated).  GPU leg: the same call through
the finest ABI boundary (knox_cmp), which must agree byte for byte.
Inputs mirror the reference's bench generator: uniform [0, 2^60) (internal/tests/gen.go:19,267),
val = 2^59 (~50 %) and val = 127 (internal/cmp/tests/bench.go:41)."""
import numpy as np
import pytest

from oracle import oracle as O

N = 1 << 16


def _pack():
    return O.generate(O.I64, 0, 1, 0, N, lo=0, span=1 << 60)


@pytest.mark.parametrize("val", [1 << 59, 127])
def test_cpu_path_count_equals_popcount(val):
    x = _pack()
    bits, cnt = O.cmp(O.I64, O.LT, x, val, poison=32)
    assert bits.size == N // 8 == 8192
    assert cnt == O.popcount(bits, N) == int((x < val).sum())
    assert (np.unpackbits(bits, bitorder="little").astype(bool) == (x < val)).all()
    if val == 1 << 59:
        assert abs(cnt / N - 0.5) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("val", [1 << 59, 127])
def test_gpu_knox_cmp_matches_cpu_path(val):
    import knoxdb_amd as K
    x = _pack()
    want_bits, want_cnt = O.cmp(O.I64, O.LT, x, val)
    with K.Context(0) as ctx:
        bits, cnt = ctx.cmp(K.BlockType.Int64, K.FilterMode.Lt, x, val, poison=32)
        assert cnt == want_cnt and (bits == want_bits).all()
        d = ctx.bits(N).upload(bits)
        assert d.count() == want_cnt  # bitset popcount == returned count
        d.free()
