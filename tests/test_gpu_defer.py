"""The scans' deferred, phase-aligned bitset bursts (knox_scan.hip kDefer) change WHEN bitset words leave the
workgroup, never what they are: every scan below runs in three contexts -- bursts tile by tile (KNOX_DEFER=0 when the context is
opened, the form the golden-vector tests of tests/test_gpu_cmp.py pin against the oracle at small sizes), deferred but unaligned
(=2) and the product form -- on the same column, and the three bitsets, counts, per-pack counts and aggregates must be identical;
the sizes make a wave flush several times (more rounds than the stage holds tiles), flush a partial stage at the end, and leave
ragged rows to the tail kernel.  One case is also checked against the oracle outright."""
import os

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf, Or
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
from oracle import oracle as O

pytestmark = pytest.mark.gpu
R = ReducerFunc


def _ctx(mode):
    old = os.environ.get("KNOX_DEFER")
    if mode is None:
        os.environ.pop("KNOX_DEFER", None)
    else:
        os.environ["KNOX_DEFER"] = mode
    try:
        return K.Context(0)       # the variable is read when the context is opened
    finally:
        if old is None:
            os.environ.pop("KNOX_DEFER", None)
        else:
            os.environ["KNOX_DEFER"] = old


CASES = [
    # (dtype, rows, generator kind, lo, span, tree, aggs)
    (BlockType.Int64, (1 << 28) + 12345, 0, -(1 << 40), 1 << 41, lambda: And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [(R.Sum, 0), (R.Count, 0)]),
    (BlockType.Uint64, 1 << 28, 0, 0, 1 << 41, lambda: And(Leaf(0, FilterMode.Equal, 12345)), []),
    (BlockType.Int32, (1 << 28) + 77, 0, -(1 << 30), 1 << 31, lambda: And(Leaf(0, FilterMode.Range, -(1 << 28), 1 << 28)), [(R.Sum, 0)]),
    (BlockType.Int16, 1 << 28, 0, -(1 << 15), 1 << 16, lambda: And(Leaf(0, FilterMode.Lt, 100)), [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)]),
    (BlockType.Uint8, (1 << 28) + 3, 0, 0, 256, lambda: And(Leaf(0, FilterMode.Range, 64, 127)), [(R.Sum, 0)]),
    (BlockType.Float64, 1 << 27, 2, 0, 0, lambda: And(Leaf(0, FilterMode.Lt, 0.25)), [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)]),
]


def test_deferred_bursts_write_the_same_bitsets():
    ctxs = [(_ctx("0"), "tile by tile"), (_ctx("2"), "deferred, unaligned"), (_ctx(None), "product")]
    try:
        base = ctxs[0][0]
        for dtype, n, kind, lo, span, tree, aggs in CASES:
            pack = 1 << 20
            col = base.generate(dtype, kind, 11, n, lo=lo, span=span, pack_rows=pack)
            ref = None
            for ctx, name in ctxs:
                c = col if ctx is base else ctx.wrap(dtype, col.device_ptr, n, pack)
                bits = ctx.bits(n)
                res = ctx.scan(tree(), [c], aggs=aggs, bits_out=bits, pack_counts=(n % pack == 0))
                got = (bits.bytes().tobytes(), int(res.count), None if res.pack_counts is None else res.pack_counts.tobytes(),
                       [bytes(s) for s in res.states])
                if ref is None:
                    ref = got
                else:
                    assert got[0] == ref[0], (dtype, n, name, "bitset")
                    assert got[1] == ref[1] and got[2] == ref[2], (dtype, n, name, "counts")
                    assert got[3] == ref[3], (dtype, n, name, "aggregate states")
                bits.free()
                if c is not col:
                    c.free()
            col.free()
        # two leaves (the tree of BASELINE configs[2]) and three: the pipelined and the 3-workgroups-per-CU instantiations
        n, pack = 1 << 27, 1 << 20
        a = base.generate(BlockType.Float64, 1, 3, n, pack_rows=pack)
        b = base.generate(BlockType.Float64, 2, 4, n, pack_rows=pack)
        i = base.generate(BlockType.Int64, 0, 5, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
        for tree, cols_, aggs in ((And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5)), [a, b], [(R.Min, 0), (R.Max, 0), (R.Mean, 0)]),
                                  (Or(Leaf(0, FilterMode.Lt, 100.0), Leaf(1, FilterMode.Gt, 0.9)), [a, b], []),
                                  (And(Leaf(0, FilterMode.Ge, 256.0), Leaf(1, FilterMode.Lt, 0.5), Leaf(2, FilterMode.Gt, 0)), [a, b, i], [(R.Sum, 2)])):
            ref = None
            for ctx, name in ctxs:
                cs = cols_ if ctx is base else [ctx.wrap(c.dtype, c.device_ptr, n, pack) for c in cols_]
                bits = ctx.bits(n)
                res = ctx.scan(tree, cs, aggs=aggs, bits_out=bits, pack_counts=True)
                got = (bits.bytes().tobytes(), int(res.count), res.pack_counts.tobytes(), [bytes(s) for s in res.states])
                assert ref is None or got == ref, (name, len(cols_))
                ref = ref or got
                bits.free()
                if ctx is not base:
                    for c in cs:
                        c.free()
        for o in (a, b, i):
            o.free()
    finally:
        for ctx, _ in ctxs:
            ctx.close()


def test_a_scan_with_several_flush_phases_equals_the_oracle():
    """2^27 int64 rows: 256 rounds of 512 workgroups at 2 WG/CU, i.e. three flushes per wave -- the whole bitset against the
    oracle's cmp_bw over rows regenerated on the host, count and wrapping sum too"""
    n, pack = 1 << 27, 1 << 20
    lo, hi = -(1 << 38), 1 << 38
    with K.Context(0) as ctx:
        col = ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
        bits = ctx.bits(n)
        res = ctx.scan(And(Leaf(0, FilterMode.Range, lo, hi)), [col], aggs=[(R.Sum, 0)], bits_out=bits, pack_counts=True)
        x = O.generate_mt(O.I64, 0, 2, 0, n, lo=-(1 << 40), span=1 << 41)
        want, cnt = O.cmp(O.I64, O.RG, x, lo, hi)
        assert bits.bytes().tobytes() == want.tobytes()
        assert int(res.count) == int(cnt) == int(res.pack_counts.sum())
        sel = (x >= lo) & (x <= hi)
        assert int(res.states[0].isum) == int(x[sel].sum(dtype=np.int64).view(np.uint64))


def test_random_sizes_and_trees_match_the_tile_by_tile_form():
    """60 random scans (every numeric type, 1 .. 2^25 rows incl. ragged ones, one to three leaves, with and without aggregates and
    per-pack counts): the product context against a KNOX_DEFER=0 context on the same columns, bit for bit"""
    import random
    rng = random.Random(20261004)
    types = [(BlockType.Int64, -(1 << 40), 1 << 41), (BlockType.Uint64, 0, 1 << 41), (BlockType.Int32, -(1 << 30), 1 << 31), (BlockType.Uint32, 0, 1 << 31),
             (BlockType.Int16, -(1 << 15), 1 << 16), (BlockType.Uint16, 0, 1 << 16), (BlockType.Int8, -128, 256), (BlockType.Uint8, 0, 256),
             (BlockType.Float64, 0, 0), (BlockType.Float32, 0, 0)]
    ref_ctx, ctx = _ctx("0"), _ctx(None)
    try:
        for case in range(60):
            dtype, lo, span = rng.choice(types)
            is_f = dtype in (BlockType.Float64, BlockType.Float32)
            n = rng.choice([rng.randrange(1, 5000), rng.randrange(5000, 1 << 20), rng.randrange(1 << 20, 1 << 25), 1 << rng.randrange(16, 26)])
            pack = rng.choice([0, 4096, 1 << 16, 1 << 20])
            if pack and n % pack and rng.random() < 0.5:
                n = max(pack, n - n % pack)
            n_leaves = rng.choice([1, 1, 1, 2, 3])
            cols = [ref_ctx.generate(dtype, 2 if is_f else 0, 100 + case * 4 + k, n, lo=lo, span=span, pack_rows=pack) for k in range(n_leaves)]

            def leaf(k):
                if is_f:
                    return Leaf(k, rng.choice([FilterMode.Lt, FilterMode.Ge]), rng.random())
                a = lo + rng.randrange(span)
                b = lo + rng.randrange(span)
                return Leaf(k, FilterMode.Range, min(a, b), max(a, b)) if rng.random() < 0.7 else Leaf(k, rng.choice([FilterMode.Lt, FilterMode.Gt, FilterMode.NotEqual]), a)
            leaves = [leaf(k) for k in range(n_leaves)]
            tree = (And if rng.random() < 0.6 else Or)(*leaves)
            aggs = rng.choice([[], [(R.Sum, 0)], [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)], [(R.Count, 0)], [(R.First, 0), (R.Last, 0)]])
            want_packs = bool(pack) and n % pack == 0 and rng.random() < 0.7
            got = []
            for c in (ref_ctx, ctx):
                cs = cols if c is ref_ctx else [c.wrap(dtype, x.device_ptr, n, pack) for x in cols]
                bits = c.bits(n)
                res = c.scan(tree, cs, aggs=aggs, bits_out=bits, pack_counts=want_packs)
                got.append((bits.bytes().tobytes(), int(res.count), res.pack_counts.tobytes() if want_packs else None, [bytes(s) for s in res.states]))
                bits.free()
                if c is not ref_ctx:
                    for x in cs:
                        x.free()
            assert got[0] == got[1], (case, dtype, n, pack, n_leaves, aggs)
            for x in cols:
                x.free()
    finally:
        ref_ctx.close()
        ctx.close()


def test_launch_shape_and_parked_lds_agree_with_the_runtime():
    """knox_kernels.h scan_shape is the one table both the kernel's parked LDS words and run_fused's workgroups per CU come
    from; every deferring instantiation asks the runtime at its first launch whether that many workgroups are resident per
    CU.  Runs last in this file: the tests above have launched the plain, aggregate-heavy and multi-leaf forms of every
    width.  (The single-leaf forms of all ten types with every aggregate: tests/test_gpu_scan.py, counted by the same
    process-wide counter when the suite runs in one process, as the driver runs it.)"""
    from knoxdb_amd import _lib
    with K.Context(0) as c:
        for dtype in (BlockType.Int64, BlockType.Int32, BlockType.Int16, BlockType.Int8, BlockType.Float64, BlockType.Float32):
            is_f = dtype in (BlockType.Float64, BlockType.Float32)
            n = 1 << 20
            cols = [c.generate(dtype, 2 if is_f else 0, 7 + k, n, lo=0, span=100, pack_rows=1 << 16) for k in range(3)]
            lf = (lambda k: Leaf(k, FilterMode.Lt, 0.5)) if is_f else (lambda k: Leaf(k, FilterMode.Lt, 50))
            for nl in (1, 2, 3):
                for aggs in ([], [(R.Sum, 0)], [(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.First, 0)]):
                    bits = c.bits(n)
                    c.scan(And(*[lf(k) for k in range(nl)]), cols[:nl], aggs=aggs, bits_out=bits, pack_counts=True)
                    bits.free()
            for x in cols:
                x.free()
    assert _lib.lib().knox_scan_occupancy_mismatches() == 0
