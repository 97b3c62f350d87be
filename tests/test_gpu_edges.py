"""GPU edge cases and error behaviour of the C ABI: empty / ragged inputs, odd pack sizes, the
result ring, borrowed device memory on a caller's stream, error codes instead of crashes."""
import ctypes as C

import os

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf, Or, _lib
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
from oracle import oracle as O
from helpers import oracle_match, rand_col

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def test_empty_column_scan(ctx):
    col = ctx.upload(np.zeros(0, np.int64), 1024)
    bits = ctx.bits(0)
    res = ctx.scan(And(Leaf(0, FilterMode.Ge, 0)), [col], aggs=[(R.Sum, 0), (R.Min, 0)], bits_out=bits)
    assert res.count == 0 and bits.bytes().size == 0
    assert res.value(0) == (0, False, -1) and res.value(1)[1] is False   # Value() of an unseen reducer
    assert bits.count() == 0 and bits.indexes().size == 0
    col.free(); bits.free()


@pytest.mark.parametrize("pack_rows", [64, 192, 1000 * 64, 4096])
def test_odd_pack_sizes_and_short_last_pack(ctx, pack_rows):
    rng = np.random.default_rng(pack_rows)
    n = 7 * pack_rows + pack_rows // 2 + 3 if pack_rows > 64 else 1000
    n -= n % 1  # any n
    x = rand_col(BlockType.Int32, n, rng)
    col = ctx.upload(x, pack_rows)
    bits = ctx.bits(n)
    tree = And(Leaf(0, FilterMode.Gt, 3))
    n_packs = (n + pack_rows - 1) // pack_rows
    skip = (rng.random(n_packs) < 0.3).astype(np.uint8)
    want, cnt = O.cmp(O.I32, O.GT, x, 3)
    # zero the rows of skipped packs in the expectation only if they hold no match (contract); here we
    # give skip flags that are TRUE to the contract by pruning only packs without matches
    per_pack = [int(np.unpackbits(want, bitorder="little")[p * pack_rows:min(n, (p + 1) * pack_rows)].sum()) for p in range(n_packs)]
    skip = np.array([1 if (skip[p] and per_pack[p] == 0) else 0 for p in range(n_packs)], dtype=np.uint8)
    res = ctx.scan(tree, [col], aggs=[(R.Count, 0)], bits_out=bits, pack_counts=True, pack_skip=skip)
    assert (bits.bytes(poison=16) == want).all() and res.count == cnt
    assert res.pack_counts.tolist() == per_pack
    col.free(); bits.free()


@pytest.mark.parametrize("n,pack_rows", [(3 * 4096 + 100, 4096), (5 * 960 + 7, 960), (8 * 65536, 65536), (3 * 65536 + 5, 65536), (4 * 1024, 1024)])
def test_pruned_packs_never_match_whatever_the_pack_shape(ctx, n, pack_rows):
    """pack_skip[p] = 1 means "no row of pack p matches" (pack/stats/match.go:112-195 pruned it) for the bitset, the count,
    the per-pack counts AND the aggregates -- also when the packs are too small or too ragged for the fused kernel's
    pack-granular walk (a short last pack is the normal case in the reference)."""
    rng = np.random.default_rng(n + pack_rows)
    n_packs = (n + pack_rows - 1) // pack_rows
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    f = (rng.integers(0, 4096, n) / 16.0).astype(np.float64)
    skip = (rng.random(n_packs) < 0.4).astype(np.uint8)
    skip[min(1, n_packs - 1)] = 1
    live = ~np.repeat(skip.astype(bool), pack_rows)[:n]
    cv, cf = ctx.upload(v, pack_rows), ctx.upload(f, pack_rows)
    bits = ctx.bits(n)
    for tree, want in ((And(Leaf(0, FilterMode.Lt, 250)), (v < 250) & live),
                       (And(Leaf(0, FilterMode.Range, -500, 500), Leaf(1, FilterMode.Ge, 64.0)), (v >= -500) & (v <= 500) & (f >= 64.0) & live)):
        res = ctx.scan(tree, [cv, cf], aggs=[(R.Sum, 0), (R.Max, 1), (R.Count, 0)], bits_out=bits, pack_counts=True, pack_skip=skip)
        assert res.count == int(want.sum())
        assert (np.unpackbits(bits.bytes(), bitorder="little")[:n].astype(bool) == want).all()
        assert res.pack_counts.tolist() == [int(want[p * pack_rows:(p + 1) * pack_rows].sum()) for p in range(n_packs)]
        assert res.value(0)[0] == int(v[want].sum()) and res.value(2)[0] == int(want.sum())
        assert res.value(1)[0] == f[want].max() and res.value(1)[2] == int(np.flatnonzero(want & (f == f[want].max()))[0])
        # without a result bitset the contract is the same
        res2 = ctx.scan(tree, [cv, cf], aggs=[(R.Sum, 0)], pack_skip=skip)
        assert res2.count == int(want.sum()) and res2.value(0)[0] == int(v[want].sum())
    for o in (cv, cf, bits):
        o.free()


def test_result_ring_is_fifo_and_bounded(ctx):
    x = np.arange(100000, dtype=np.int64)
    col = ctx.upload(x, 1024)
    plans = [ctx.plan(And(Leaf(0, FilterMode.Lt, k)), [col], aggs=[(R.Count, 0)]) for k in (10, 200, 3000, 40000, 5)]
    for p in plans[:4]:
        ctx.scan_async(p)
    with pytest.raises(K.KnoxError) as e:   # a 5th uncollected scan
        ctx.scan_async(plans[4])
    assert e.value.code == -8
    with pytest.raises(K.KnoxError) as e:   # the synchronous form refuses to jump the queue
        ctx.scan(And(Leaf(0, FilterMode.Lt, 1)), [col])
    assert e.value.code == -8
    assert [ctx.scan_collect(p).count for p in plans[:4]] == [10, 200, 3000, 40000]
    with pytest.raises(K.KnoxError):        # nothing left to collect
        ctx.scan_collect(plans[0])
    col.free()


def test_error_codes_instead_of_crashes(ctx):
    L = _lib.lib()
    x = ctx.upload(np.arange(1000, dtype=np.int64), 128)
    y = ctx.upload(np.arange(999, dtype=np.int64), 128)
    with pytest.raises(K.KnoxError) as e:     # columns of different length
        ctx.scan(And(Leaf(0, FilterMode.Lt, 5), Leaf(1, FilterMode.Lt, 5)), [x, y])
    assert e.value.code == -1
    with pytest.raises(K.KnoxError) as e:     # bitset of the wrong size
        ctx.scan(And(Leaf(0, FilterMode.Lt, 5)), [x], bits_out=ctx.bits(999))
    assert e.value.code == -1
    plan = ctx.plan(And(Leaf(0, FilterMode.Lt, 5)), [x], aggs=[(R.Sum, 0)])
    plan[4][0].func = 42                      # unknown reducer function
    with pytest.raises(K.KnoxError) as e:
        ctx.scan_async(plan)
    assert e.value.code == -5
    a, b = ctx.bits(100), ctx.bits(101)
    with pytest.raises(K.KnoxError):          # size mismatch: error here, silent no-op in Go
        a.and_(b)
    h = C.c_void_p()
    assert L.knox_col_upload(ctx.h, 99, None, 0, 0, C.byref(h)) == -1          # unknown dtype
    assert L.knox_col_wrap(ctx.h, 1, C.c_void_p(8), 10, 0, C.byref(h)) == -1   # misaligned device pointer
    assert L.knox_cmp(None, 1, 1, None, 0, None, None, None) == -1
    assert L.knox_ctx_open(12345, C.byref(h)) == -1
    ptr = np.array([(0 << 32) | 9999], dtype=np.uint64)                        # (off,len) outside buf
    assert L.knox_col_upload_bytes(ctx.h, ptr.ctypes.data_as(C.c_void_p), 1, np.zeros(4, np.uint8).ctypes.data_as(C.c_void_p), 4, 0, C.byref(h)) == -7
    for o in (x, y, a, b):
        o.free()


_TORCH_SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import torch                      # torch's HIP runtime must initialise before libknoxgpu's
t = torch.arange(0, 1 << 20, dtype=torch.int64, device="cuda") * 3 - 1000
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
s = torch.cuda.current_stream()
c = K.Context(0, stream=s.cuda_stream)
col = c.wrap(BlockType.Int64, t.data_ptr(), t.numel(), 1 << 16, keep=t)
bits = c.bits(t.numel())
res = c.scan(And(Leaf(0, FilterMode.Range, 0, 999999)), [col], aggs=[(R.Sum, 0), (R.Count, 0)], bits_out=bits)
h = t.cpu().numpy()
sel = (h >= 0) & (h <= 999999)
assert res.count == int(sel.sum()) and int(res.value(0)[0]) == int(h[sel].sum())
assert (bits.bytes() == np.packbits(sel, bitorder="little")).all()
col.free(); bits.free(); c.close()
assert int(t[5].item()) == -985   # borrowed memory is never freed by knox
print("torch-borrowed ok")
"""


def test_borrowed_torch_memory_on_the_callers_stream():
    # own process: torch bundles its own HIP runtime, which has to come up before libknoxgpu's
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _TORCH_SCRIPT % root], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "torch-borrowed ok" in out.stdout, out.stdout + out.stderr


def _free_device_memory():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_many_alloc_free_cycles_do_not_leak(ctx):
    ctx.sync()
    free0 = _free_device_memory()
    for _ in range(30):
        col = ctx.generate(BlockType.Int64, 0, 1, 1 << 22, lo=0, span=100)
        bits = ctx.bits(1 << 22)
        ctx.scan(And(Leaf(0, FilterMode.Lt, 50)), [col], aggs=[(R.Sum, 0)], bits_out=bits)
        col.free(); bits.free()
    assert free0 - _free_device_memory() < (64 << 20)


def test_four_leaf_mixed_float_int_tree_all_widths(ctx):
    rng = np.random.default_rng(77)
    n = 50000 + 13
    for bts in ((BlockType.Int32, BlockType.Float32, BlockType.Uint32, BlockType.Int32),
                (BlockType.Int16, BlockType.Uint16, BlockType.Int16, BlockType.Uint16),
                (BlockType.Int8, BlockType.Uint8, BlockType.Uint8, BlockType.Int8),
                (BlockType.Float64, BlockType.Int64, BlockType.Uint64, BlockType.Float64)):
        cols = [rand_col(b, n, rng) for b in bts]
        L = Leaf
        tree = And(Or(L(0, FilterMode.Lt, cols[0][1]), L(1, FilterMode.Ge, cols[1][2])), Or(L(2, FilterMode.NotEqual, cols[2][3]), L(3, FilterMode.Range, cols[3][4], cols[3][4])))
        dev = [ctx.upload(c, 4096) for c in cols]
        bits = ctx.bits(n)
        res = ctx.scan(tree, dev, aggs=[(R.Count, 0)], bits_out=bits)
        want = oracle_match(tree, cols, n)
        assert (bits.bytes(poison=8) == want).all() and res.count == O.popcount(want, n), bts
        for d in dev:
            d.free()
        bits.free()


def test_two_threads_two_contexts_scan_concurrently():
    """(b) Threading: one knox_ctx per host thread, no shared mutable state in the library --
    many queries in flight like the reference's b.RunParallel (benchmark3_test.go:112-127)."""
    import threading
    rng = np.random.default_rng(77)
    n = 300000 + 7
    data = [rng.integers(-1000, 1000, size=n).astype(np.int64) for _ in range(4)]
    errors, results = [], [None] * 4

    def worker(i):
        try:
            with K.Context(0) as c:
                col = c.upload(data[i], 4096)
                out = []
                for rep in range(20):
                    lo = -500 + 10 * rep + i
                    bits = c.bits(n)
                    r = c.scan(And(Leaf(0, FilterMode.Range, lo, lo + 300)), [col], aggs=[(R.Sum, 0)], bits_out=bits, pack_counts=True)
                    out.append((r.count, int(r.value(0)[0]), int(r.pack_counts.sum()), bits.count()))
                    bits.free()
                col.free()
                results[i] = out
        except Exception as e:  # surfaced below: an exception in a thread must fail the test
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(4):
        for rep, (cnt, s, pc, bc) in enumerate(results[i]):
            lo = -500 + 10 * rep + i
            m = (data[i] >= lo) & (data[i] <= lo + 300)
            assert cnt == pc == bc == int(m.sum()) and s == int(data[i][m].sum()), (i, rep)


@pytest.mark.skipif(os.environ.get("KNOX_PLACEMENT") == "0", reason="placement switched off by the environment")
def test_arena_allocation_is_transparent_and_bounded():
    """All device memory of a context comes out of one arena (include/knoxgpu.h): results are the same with
    it on or off; with it on nothing is held outside the arena, the mapped bytes never exceed the budget, the bitset
    allocated FOR a column was verified against it, and a budget that is too small fails with NOMEM."""
    n = (1 << 27) + 4096 * 3 + 17  # > 1 GiB of int64: assembled from probed chunks
    x = (np.arange(n, dtype=np.int64) * 2654435761) % 1000003
    want = int(((x >= 1000) & (x <= 500000)).sum())
    for on in (True, False):
        with K.Context(0) as c:
            c.set_placement(on)
            if on:
                c.set_arena(16 << 30)
            col = c.upload(x, 1 << 20)
            bits = c.bits(n, for_col=col)  # knox_bits_alloc_for: a class the column does not lie in
            info = c.arena_info()
            if on and info["enabled"]:
                assert info["mapped_gib"] <= 16 and info["used_gib"] * 2**30 >= n * 8
                ratio, tried = bits.pair_check          # the (column, bitset) pair was measured when the bitset was chosen
                assert tried >= 1 and 0.7 < ratio < 1.1
            else:
                assert bits.pair_check == (0.0, 0) and info["mapped_gib"] == 0
            r = c.scan(And(Leaf(0, FilterMode.Range, 1000, 500000)), [col], aggs=[(R.Sum, 0)], bits_out=bits, pack_counts=True)
            assert r.count == want == bits.count() == int(r.pack_counts.sum())
            assert int(r.value(0)[0]) == int(x[(x >= 1000) & (x <= 500000)].sum())
            assert c.placement_info() == {"held_bytes": 0, "held_regions": 0}
            bits.free()
            col.free()
            assert c.arena_info()["used_gib"] * 2**30 < (200 << 20)     # only the context's scratch and the classes' 32 MiB reference windows (<= 4: Arena::kMaxClasses) are left
    with K.Context(0) as c:
        c.set_arena(4 << 30)
        small = c.upload(x[:1 << 20], 1 << 16)
        if c.arena_info()["enabled"]:
            with pytest.raises(K.KnoxError) as e:
                c.generate(K.BlockType.Int64, 0, 1, 1 << 30, pack_rows=1 << 20)      # 8 GiB do not fit a 4 GiB arena
            assert e.value.code == -4
            assert c.arena_info()["mapped_gib"] <= 4
        small.free()


@pytest.mark.skipif(os.environ.get("KNOX_PLACEMENT") == "0", reason="placement switched off by the environment")
def test_arena_reuses_freed_memory_and_trims_only_on_request():
    """freed columns go back to the arena's free lists, not to the driver (returning multi-GiB blocks slows live
    neighbours): the next column of that size lands on the same address; knox_ctx_arena_trim returns whole
    free chunks; knox_ctx_close returns everything"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    f0, f1, f2, f3, tot = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f0), C.byref(tot))
    n = 3 << 27                     # 3 GiB of int64: spans chunks
    with K.Context(0) as c:
        c.set_arena(32 << 30)
        col = c.generate(K.BlockType.Int64, 3, 0, n, lo=0, pack_rows=1 << 20)
        if not c.arena_info()["enabled"]:
            pytest.skip("no virtual-memory API on this device")
        mapped = c.arena_info()["mapped_gib"]
        c.sync()
        hip.hipMemGetInfo(C.byref(f1), C.byref(tot))
        col.free()
        hip.hipMemGetInfo(C.byref(f2), C.byref(tot))
        assert abs(f2.value - f1.value) < (8 << 20)                      # nothing went back to the driver
        col2 = c.generate(K.BlockType.Int64, 3, 0, n, lo=1, pack_rows=1 << 20)
        assert c.arena_info()["mapped_gib"] == mapped          # the same physical chunks again, nothing new from the driver
        r = c.scan(And(Leaf(0, FilterMode.Le, 10)), [col2], aggs=[(R.Sum, 0)])
        assert r.count == 10 and int(r.value(0)[0]) == sum(range(1, 11))
        col2.free()
        freed = c.arena_trim()
        assert freed >= (3 << 30) and abs(c.arena_info()["mapped_gib"] * 2**30 - (mapped * 2**30 - freed)) <= (4 << 20)
    hip.hipMemGetInfo(C.byref(f3), C.byref(tot))
    assert f0.value - f3.value < (256 << 20)                             # the closed context holds nothing


def test_pinned_host_memory_round_trip(ctx):
    """knox_host_alloc: page-locked memory is DMA'd from / into directly (no staging ring); results
    are the same as with pageable memory"""
    n = 3 * (1 << 20) + 11
    pinned = ctx.host_array(n, np.int64)
    pinned[:] = (np.arange(n, dtype=np.int64) * 7919) % 100003
    pageable = np.array(pinned)  # ordinary copy
    c1, c2 = ctx.upload(pinned, 1 << 16), ctx.upload(pageable, 1 << 16)
    tree = And(Leaf(0, FilterMode.Range, 500, 40000))
    b1, b2 = ctx.bits(n), ctx.bits(n)
    r1 = ctx.scan(tree, [c1], aggs=[(R.Sum, 0)], bits_out=b1)
    r2 = ctx.scan(tree, [c2], aggs=[(R.Sum, 0)], bits_out=b2)
    m = (pageable >= 500) & (pageable <= 40000)
    assert r1.count == r2.count == int(m.sum()) and r1.value(0)[0] == r2.value(0)[0] == int(pageable[m].sum())
    assert (b1.bytes() == b2.bytes()).all()
    back = ctx.host_array(n, np.int64)
    assert ctx.L.knox_col_download(ctx.h, c1.h, 0, n, back.ctypes.data_as(C.c_void_p)) == 0
    assert (back == pageable).all()
    for o in (c1, c2, b1, b2):
        o.free()
    ctx.host_free(pinned)
    ctx.host_free(back)
