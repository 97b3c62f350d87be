"""Ingest that overlaps the scans (include/knoxgpu.h knox_col_create / _append_async / _reset / _upload_async / _wait):
the device form of the reference's loadPack / filter.Match interleave (internal/pack/table/reader.go:329-334, :484).
The pipelined scan of a table that is streamed run by run through two device buffers must equal, bit for bit, the scan
of the same runs uploaded the blocking way -- bitsets, per-pack counts, counts, wrapping sums -- and the oracle."""
import ctypes as C

import numpy as np
import pytest

import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R
from oracle import oracle as O

pytestmark = pytest.mark.gpu

PACK = 1 << 16
A, B = -(1 << 38), 1 << 38


@pytest.fixture(scope="module")
def ctx():
    c = K.Context(0)
    yield c
    c.close()


def make_runs(rng, sizes):
    return [rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64) for n in sizes]


def resident(ctx, run, tree, aggs):
    """the blocking path: knox_col_upload, then one scan of the whole run"""
    col = ctx.upload(run, PACK)
    bits = ctx.bits(run.size)
    res = ctx.scan(tree, [col], aggs=aggs, bits_out=bits, pack_counts=True)
    out = (bits.bytes(poison=32).copy(), res.pack_counts.copy(), res.count, int(res.value(0)[0]))
    col.free()
    bits.free()
    return out


@pytest.mark.parametrize("pinned", [True, False], ids=["page-locked source", "pageable source"])
def test_pipelined_scan_equals_resident_scan(ctx, pinned):
    rng = np.random.default_rng(11)
    sizes = [1 << 21, 1 << 21, 3 * PACK, 1 << 21, (1 << 20) + 777]      # equal runs, a short one, a ragged last one
    runs = make_runs(rng, sizes)
    tree = And(Leaf(0, FilterMode.Range, A, B))
    aggs = [(R.Sum, 0), (R.Count, 0)]
    want = [resident(ctx, r, tree, aggs) for r in runs]
    cap = max(sizes)
    hosts = []
    for r in runs:                                                       # the table as the host holds it
        if pinned:
            h = ctx.host_array(r.size, np.int64)
            h[:] = r
        else:
            h = r.copy()
        hosts.append(h)
    cols = [ctx.col_create(BlockType.Int64, cap, PACK), ctx.col_create(BlockType.Int64, cap, PACK)]
    assert cols[0].rows == 0 and cols[0].ready
    cols[0].append_async(hosts[0])
    got = []
    for i, h in enumerate(hosts):
        cur, nxt = cols[i & 1], cols[(i + 1) & 1]
        if i + 1 < len(hosts):
            nxt.reset()                                                  # BEFORE this run's scan is enqueued (include/knoxgpu.h)
            nxt.append_async(hosts[i + 1])                               # copy stream: overlaps the scan below
        assert cur.rows == h.size
        bits = ctx.bits(h.size)
        plan = ctx.plan(tree, [cur], aggs=aggs)
        ctx.scan_async(plan, bits_out=bits, pack_counts=True)            # the scan stream waits for cur's upload on the device
        res = ctx.scan_collect(plan, (h.size + PACK - 1) // PACK)
        got.append((bits.bytes(poison=32).copy(), res.pack_counts.copy(), res.count, int(res.value(0)[0])))
        bits.free()
    for i, (g, w) in enumerate(zip(got, want)):
        assert (g[0] == w[0]).all(), i
        assert (g[1] == w[1]).all() and g[2] == w[2] and g[3] == w[3], i
    # and the oracle, on the ragged run (cmp_bw -> Indexes -> SumReducer)
    wb, wc, ws = O.scan_i64_bw_sum(runs[-1], PACK, A, B)
    assert (got[-1][0] == wb).all() and got[-1][2] == wc and got[-1][3] == ws
    for c in cols:
        c.wait()
        assert c.ready
        c.free()
    if pinned:
        for h in hosts:
            ctx.host_free(h)


def test_scan_right_behind_its_own_upload(ctx):
    """append_async(A); scan_async(A): nothing but the device-side wait orders the scan behind the copy"""
    rng = np.random.default_rng(5)
    n = (1 << 22) + 12345
    x = rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64)
    pin = ctx.host_array(n, np.int64)
    pin[:] = x
    col = ctx.upload_async(pin, PACK)
    bits = ctx.bits(n)
    res = ctx.scan(And(Leaf(0, FilterMode.Range, A, B)), [col], aggs=[(R.Sum, 0), (R.Min, 0), (R.Max, 0)], bits_out=bits, pack_counts=True)
    wb, wc, ws = O.scan_i64_bw_sum(x, PACK, A, B)
    assert (bits.bytes(poison=32) == wb).all() and res.count == wc and int(res.value(0)[0]) == ws
    sel = x[(x >= A) & (x <= B)]
    assert int(res.value(1)[0]) == int(sel.min()) and int(res.value(2)[0]) == int(sel.max())
    # the other readers wait the same way: reduce over a fresh asynchronous upload, download of one
    col2 = ctx.upload_async(pin, PACK)
    st = ctx.reduce(col2, bits)
    assert st.count == wc and int(K.agg_value(st, R.Sum)[0]) == ws
    col3 = ctx.upload_async(pin, PACK)
    assert (col3.download() == x).all()
    for o in (col, col2, col3, bits):
        o.free()
    ctx.host_free(pin)


def test_append_pack_by_pack_and_capacity(ctx):
    """a run assembled pack by pack (the reference's unit of loading) equals one upload; capacity is enforced; reset refills"""
    rng = np.random.default_rng(9)
    n_packs = 9
    x = rng.integers(0, 1000, size=n_packs * PACK - 100, dtype=np.int32)
    col = ctx.col_create(BlockType.Int32, n_packs * PACK, PACK)
    for p in range(n_packs):
        col.append_async(np.ascontiguousarray(x[p * PACK:(p + 1) * PACK]))   # pageable pieces: not referenced after return
    assert col.rows == x.size
    tree = And(Leaf(0, FilterMode.Lt, 500))
    bits = ctx.bits(x.size)
    res = ctx.scan(tree, [col], aggs=[(R.Sum, 0)], bits_out=bits, pack_counts=True)
    wb, wc = O.cmp(O.I32, O.LT, x, 500)
    assert (bits.bytes(poison=32) == wb).all() and res.count == wc and int(res.value(0)[0]) == int(x[x < 500].sum())
    assert ctx.L.knox_col_append_async(ctx.h, col.h, x.ctypes.data_as(C.c_void_p), 101) == -7      # KNOX_E_RANGE: over capacity
    col.reset()
    assert col.rows == 0
    y = (x[: 3 * PACK] + 7).astype(np.int32)
    col.append_async(y)
    b2 = ctx.bits(y.size)
    r2 = ctx.scan(tree, [col], aggs=[(R.Sum, 0)], bits_out=b2)
    wb2, wc2 = O.cmp(O.I32, O.LT, y, 500)
    assert (b2.bytes(poison=32) == wb2).all() and r2.count == wc2
    # not an appendable column / wrong dtype -> KNOX_E_INVALID
    plain = ctx.upload(y, PACK)
    assert ctx.L.knox_col_append_async(ctx.h, plain.h, y.ctypes.data_as(C.c_void_p), 1) == -1
    for o in (col, bits, b2, plain):
        o.free()


def test_packed_columns_streamed_on_the_copy_stream(ctx):
    """bit-packed containers appended asynchronously to two packed columns in turn (the cold-scan form: the data crosses PCIe
    packed), scanned as they are: every run's bitset, per-pack counts, count and wrapping sum equal knox_packed_upload +
    knox_packed_scan of the same containers and the oracle's evaluation of the decoded values"""
    rng = np.random.default_rng(21)
    pack, bits_per = 8192, 13
    base = 1_700_000_000_000
    runs = []
    for n_packs, last in ((6, pack), (3, pack), (5, pack - 1000)):          # the third run ends in a short pack
        vals = [base + rng.integers(0, 1 << bits_per, size=(last if p + 1 == n_packs else pack), dtype=np.int64) for p in range(n_packs)]
        runs.append((vals, [O.bitpack_container_store(O.I64, v) for v in vals]))
    a, b = base + 1000, base + 5000
    cols = [ctx.packed_create(BlockType.Int64, pack, 8, 8 * pack * 2), ctx.packed_create(BlockType.Int64, pack, 8, 8 * pack * 2)]
    pins = []
    def pinned(conts):                                                       # containers in page-locked memory: DMA'd in place
        out = []
        for c in conts:
            h = ctx.host_array(len(c), np.uint8)
            h[:] = np.frombuffer(c, dtype=np.uint8)
            pins.append(h)
            out.append(h)
        return out
    cols[0].append_async(pinned(runs[0][1]))
    for i, (vals, conts) in enumerate(runs):
        cur, nxt = cols[i & 1], cols[(i + 1) & 1]
        if i + 1 < len(runs):
            nxt.reset()
            nxt.append_async(pinned(runs[i + 1][1]) if i == 0 else runs[i + 1][1])   # page-locked and pageable sources
        x = np.concatenate(vals)
        assert cur.rows == x.size
        bits = ctx.bits(x.size)
        cnt, st, pc = cur.scan(FilterMode.Range, a, b, want_sum=True, bits_out=bits, pack_counts=True)
        ref = ctx.packed_upload(BlockType.Int64, conts, pack)
        rbits = ctx.bits(x.size)
        rcnt, rst, rpc = ref.scan(FilterMode.Range, a, b, want_sum=True, bits_out=rbits, pack_counts=True)
        wb, wc = O.cmp(O.I64, O.RG, x, a, b)
        assert cnt == rcnt == wc and (pc == rpc).all() and int(pc.sum()) == wc
        assert (bits.bytes(poison=32) == wb).all() and (rbits.bytes(poison=32) == wb).all()
        assert int(K.agg_value(st, R.Sum)[0]) == int(K.agg_value(rst, R.Sum)[0]) == int(x[(x >= a) & (x <= b)].sum())
        for o in (bits, rbits, ref):
            o.free()
    # capacity: a ninth pack does not fit
    cols[0].reset()
    cols[0].append_async(runs[0][1] + runs[1][1][:2])
    import ctypes as C
    one = np.frombuffer(runs[0][1][0], dtype=np.uint8)
    ptrs, lens = (C.c_void_p * 1)(one.ctypes.data), (C.c_size_t * 1)(one.size)
    assert ctx.L.knox_packed_append_async(ctx.h, cols[0].h, ptrs, lens, 1) == -7       # KNOX_E_RANGE
    for c in cols:
        c.wait()
        c.free()
    for h in pins:
        ctx.host_free(h)
