#!/usr/bin/env python3
"""Cold scan: a table that is NOT resident is streamed run by run from page-locked host memory (knox_host_alloc)
through two device buffers while the previous run is scanned -- the device form of the reference's loadPack /
filter.Match interleave (internal/pack/table/reader.go:329-334, :484).  Reports rows/s and GB/s with H2D INCLUDED
beside the plain DMA rate of the same buffers and the resident (HBM) scan rate; checks every run's count / sum /
bitset checksum against the resident scan of the same run.

  python tools/bench_stream.py [--runs 4] [--run-rows-log2 30] [--table-runs 4] [--pageable]

--table-runs runs are generated on the device and downloaded into the host table; --runs passes over that table are
streamed back (runs > table-runs wraps around), e.g. --runs 16 --table-runs 4 = a 2^34-row scan from 32 GiB of host
memory through 16 GiB of device buffers."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc as R

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=8)
ap.add_argument("--table-runs", type=int, default=4)
ap.add_argument("--run-rows-log2", type=int, default=30)
ap.add_argument("--pageable", action="store_true", help="source in ordinary (pageable) memory: goes through the copy stream's pinned ring")
ap.add_argument("--packed", type=int, default=0, metavar="BITS", help="the table holds bit-packed containers of BITS bits per value, appended to packed "
                "columns (knox_packed_append_async) and scanned as they are: the cold scan crosses PCIe at BITS / 64 of the bytes")
args = ap.parse_args()


def put_uvarint(x):
    """pkg/num/varint.go: SQLite4-style prefix varint (what the container headers use)"""
    if x <= 240:
        return bytes([x])
    if x <= 2287:
        return bytes([(x - 240) // 256 + 241, (x - 240) % 256])
    if x <= 67823:
        return bytes([249, (x - 2288) // 256, (x - 2288) % 256])
    nb = (x.bit_length() + 7) // 8
    return bytes([247 + nb]) + x.to_bytes(nb, "big")


def bitpack_container(values, base, bits):
    """[4][uvarint For][uvarint log2][uvarint N][bits LSB first in little-endian 64-bit words] (int_bitpack.go:93-133)"""
    d = (values - base).astype(np.uint64)
    m = ((d[:, None] >> np.arange(bits, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8).reshape(-1)
    payload = np.packbits(m, bitorder="little").tobytes()
    payload += b"\0" * (-len(payload) % 8)
    return bytes([4]) + put_uvarint(base) + put_uvarint(bits) + put_uvarint(values.size) + payload


def packed_mode():
    """cold scan of bit-packed data: --table-runs runs of 2^k rows, each pack one container of `bits` bits per value"""
    bits = args.packed
    n, pack = 1 << args.run_rows_log2, 1 << 20
    n_packs = n // pack
    a, b = base_v + (1 << bits) // 4, base_v + (1 << bits) // 2
    rng = np.random.default_rng(5)
    uniq = [bitpack_container(base_v + rng.integers(0, 1 << bits, size=pack, dtype=np.int64), base_v, bits) for _ in range(8)]
    clen = len(uniq[0])
    table = []
    for r in range(args.table_runs):                       # one page-locked buffer per run, the containers back to back
        h = ctx.host_array(n_packs * clen, np.uint8)
        for p in range(n_packs):
            h[p * clen:(p + 1) * clen] = np.frombuffer(uniq[(p + r) % len(uniq)], dtype=np.uint8)
        table.append([h[p * clen:(p + 1) * clen] for p in range(n_packs)])
    want = []
    for r in range(args.table_runs):                       # the resident answers: knox_packed_upload + knox_packed_scan
        pk = ctx.packed_upload(BlockType.Int64, [bytes(c) for c in table[r][:8]] * (n_packs // 8), pack)
        cnt, st, _ = pk.scan(FilterMode.Range, a, b, want_sum=True)
        want.append((cnt, int(K.agg_value(st, R.Sum)[0])))
        pk.free()
    cols = [ctx.packed_create(BlockType.Int64, pack, n_packs, n_packs * clen), ctx.packed_create(BlockType.Int64, pack, n_packs, n_packs * clen)]
    bitss = [ctx.bits(n), ctx.bits(n)]
    for _ in range(2):                                     # the plain DMA rate of these buffers
        t0 = time.perf_counter()
        for i in range(args.table_runs):
            c = cols[i & 1]
            c.reset(); c.append_async(table[i]); c.wait()
        dma_s = (time.perf_counter() - t0) / args.table_runs

    def pipelined(k):
        ok = True
        cols[0].reset(); cols[0].append_async(table[0])
        for i in range(k):
            cur, nxt = i & 1, (i + 1) & 1
            if i + 1 < k:
                cols[nxt].reset()
                cols[nxt].append_async(table[(i + 1) % args.table_runs])
            cnt, st, pc = cols[cur].scan(FilterMode.Range, a, b, want_sum=True, bits_out=bitss[cur], pack_counts=True)
            w = want[i % args.table_runs]
            ok = ok and cnt == w[0] and int(K.agg_value(st, R.Sum)[0]) == w[1] and int(pc.sum()) == cnt
        return ok
    pipelined(args.table_runs)
    t0 = time.perf_counter(); ok = pipelined(args.runs); pipe_s = (time.perf_counter() - t0) / args.runs
    gb = n_packs * clen / 1e9
    print(json.dumps({"mode": "bit-packed containers, %d bits per value" % bits, "run_rows": n, "runs_streamed": args.runs, "packed_bytes_per_run": n_packs * clen,
                      "dma_only_gb_per_s": gb / dma_s,
                      "pipelined": {"s_per_run": pipe_s, "rows_per_s": n / pipe_s, "gb_per_s_h2d_inclusive_packed_bytes": gb / pipe_s,
                                    "equivalent_gb_per_s_of_int64": n * 8 / 1e9 / pipe_s, "fraction_of_dma_rate": dma_s / pipe_s, "results_equal_resident": bool(ok)}}))

n, pack = 1 << args.run_rows_log2, 1 << 20
n_packs = n // pack
a, b = -(1 << 38), 1 << 38
ctx = K.Context(0)
ctx.set_arena(64 << 30)
tree = And(Leaf(0, FilterMode.Range, a, b))
aggs = [(R.Sum, 0), (R.Count, 0)]

base_v = 1_700_000_000_000
if args.packed:
    packed_mode()
    sys.exit(0)

# the host table (page-locked, as a Go host's block arena over knox_host_alloc would be) + the resident answers
t0 = time.perf_counter()
table, want = [], []
dev = ctx.col_create(BlockType.Int64, n, pack)
for r in range(args.table_runs):
    g = ctx.generate(BlockType.Int64, 0, 2, n, first_row=r * n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
    h = np.empty(n, np.int64) if args.pageable else ctx.host_array(n, np.int64)
    ctx.L.knox_col_download(ctx.h, g.h, 0, n, h.ctypes.data)
    bits = ctx.bits(n, for_col=g)
    res = ctx.scan(tree, [g], aggs=aggs, bits_out=bits, pack_counts=True)
    want.append((res.count, int(res.value(0)[0]), int(np.bitwise_xor.reduce(bits.bytes().view(np.uint64)))))
    table.append(h)
    g.free(); bits.free()
dev.free()
print("[setup] %d runs of 2^%d int64 rows in %s host memory: %.1f s" % (args.table_runs, args.run_rows_log2, "pageable" if args.pageable else "page-locked", time.perf_counter() - t0), file=sys.stderr)

cols = [ctx.col_create(BlockType.Int64, n, pack), ctx.col_create(BlockType.Int64, n, pack)]
for i, c in enumerate(cols):          # filled once so that the bitsets below are verified against real streams
    c.append_async(table[i % args.table_runs]); c.wait()
bitss = [ctx.bits(n, for_col=cols[0]), ctx.bits(n, for_col=cols[1])]
plans = [ctx.plan(tree, [c], aggs=aggs) for c in cols]

# 1. the plain DMA rate of these buffers: upload only
for _ in range(2):
    t0 = time.perf_counter()
    for i in range(args.table_runs):
        c = cols[i & 1]
        c.reset(); c.append_async(table[i]); c.wait()
    dma_s = (time.perf_counter() - t0) / args.table_runs
# 2. the resident scan rate: the run is already in HBM
ctx.scan_async(plans[(args.table_runs - 1) & 1], bits_out=bitss[(args.table_runs - 1) & 1], pack_counts=True)
ctx.scan_collect(plans[0], n_packs)
t0 = time.perf_counter()
for _ in range(10):
    ctx.scan_async(plans[(args.table_runs - 1) & 1], bits_out=bitss[(args.table_runs - 1) & 1], pack_counts=True)
    ctx.scan_collect(plans[0], n_packs)
scan_s = (time.perf_counter() - t0) / 10

# 3. serial: upload, wait, scan, collect -- what a blocking knox_col_upload gives
def serial(k):
    out = []
    for i in range(k):
        c = cols[0]
        c.reset(); c.append_async(table[i % args.table_runs]); c.wait()
        ctx.scan_async(plans[0], bits_out=bitss[0], pack_counts=True)
        res = ctx.scan_collect(plans[0], n_packs)
        out.append((res.count, int(res.value(0)[0])))
    return out
t0 = time.perf_counter(); serial(min(args.runs, 4)); serial_s = (time.perf_counter() - t0) / min(args.runs, 4)

# 4. pipelined: run i+1 uploads while run i scans
def pipelined(k, check):
    ok = True
    cols[0].reset(); cols[0].append_async(table[0])
    for i in range(k):
        cur, nxt = i & 1, (i + 1) & 1
        if i + 1 < k:
            cols[nxt].reset()
            cols[nxt].append_async(table[(i + 1) % args.table_runs])
        ctx.scan_async(plans[cur], bits_out=bitss[cur], pack_counts=True)
        res = ctx.scan_collect(plans[cur], n_packs)
        w = want[i % args.table_runs]
        ok = ok and res.count == w[0] and int(res.value(0)[0]) == w[1] and int(res.pack_counts.sum()) == w[0]
        if check:
            ok = ok and int(np.bitwise_xor.reduce(bitss[cur].bytes().view(np.uint64))) == w[2]
    return ok
ok_full = pipelined(args.table_runs, True)            # untimed: every run's bitset checksum against the resident scan
t0 = time.perf_counter(); ok = pipelined(args.runs, False); pipe_s = (time.perf_counter() - t0) / args.runs
gb = n * 8 / 1e9
print(json.dumps({"run_rows": n, "runs_streamed": args.runs, "table_gib": args.table_runs * n * 8 / 2**30, "device_buffers_gib": 2 * n * 8 / 2**30,
                  "source": "pageable" if args.pageable else "page-locked (knox_host_alloc)",
                  "dma_only_gb_per_s": gb / dma_s, "resident_scan_ms": scan_s * 1e3, "resident_scan_rows_per_s": n / scan_s,
                  "serial_upload_then_scan": {"s_per_run": serial_s, "rows_per_s": n / serial_s, "gb_per_s_h2d_inclusive": gb / serial_s},
                  "pipelined": {"s_per_run": pipe_s, "rows_per_s": n / pipe_s, "gb_per_s_h2d_inclusive": gb / pipe_s,
                                "fraction_of_dma_rate": dma_s / pipe_s, "results_equal_resident": bool(ok), "bitset_checksums_equal_resident": bool(ok_full)}}))
