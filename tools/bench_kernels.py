#!/usr/bin/env python3
"""Steady-state timing of every kernel family (HIP events around back-to-back launches),
with achieved GB/s against the algorithmic bytes of each."""
import argparse, ctypes as C, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf, Or
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R

ap = argparse.ArgumentParser()
ap.add_argument("--rows-log2", type=int, default=28)
args = ap.parse_args()
n, pack = 1 << args.rows_log2, 1 << 20
ctx = K.Context(0)
L = ctx.L

def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps

def scan_ms(plan, bits, steps=30):
    L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(plan, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits)
        ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    L.knox_prof_enable(ctx.h, 0)
    v = [ms[i] for i in range(k.value)]
    return statistics.median(v[len(v) // 3:])

rows = []
def rec(name, ms, nbytes):
    rows.append((name, ms, nbytes / ms / 1e6))
    print("%-58s %9.4f ms %9.1f GB/s  (%.0f%% of 8 TB/s)" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))

bits = None
for bt, w, lo, span, a, b in ((B.Int64, 8, -(1 << 40), 1 << 41, -(1 << 38), 1 << 38), (B.Int32, 4, -(1 << 30), 1 << 31, -(1 << 28), 1 << 28),
                              (B.Int16, 2, -(1 << 15), 1 << 16, -(1 << 13), 1 << 13), (B.Uint8, 1, 0, 256, 64, 127)):
    col = ctx.generate(bt, 0, 2, n, lo=lo, span=span, pack_rows=pack)
    bits = ctx.bits(n, for_col=col)   # a bitset of the column's placement class
    plan = ctx.plan(And(Leaf(0, F.Range, a, b)), [col], aggs=[(R.Sum, 0)])
    rec("scan %s Between + sum + bitset" % bt.name, scan_ms(plan, bits), n * (w + 0.125))
    plan = ctx.plan(And(Leaf(0, F.Range, a, b)), [col], aggs=[(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0), (R.First, 0), (R.Last, 0)])
    rec("scan %s Between + ALL aggs + bitset" % bt.name, scan_ms(plan, bits), n * (w + 0.125))
    col.free(); bits.free()
for bt, w in ((B.Float64, 8), (B.Float32, 4)):
    col = ctx.generate(bt, 2, 4, n, pack_rows=pack)
    bits = ctx.bits(n, for_col=col)
    plan = ctx.plan(And(Leaf(0, F.Lt, 0.25)), [col], aggs=[(R.Sum, 0), (R.Min, 0), (R.Max, 0), (R.Mean, 0)])
    rec("scan %s Lt + sum/min/max/mean + bitset" % bt.name, scan_ms(plan, bits), n * (w + 0.125))
    col.free(); bits.free()
c1 = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
c2 = ctx.generate(B.Int64, 0, 3, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
c3 = ctx.generate(B.Float64, 2, 4, n, pack_rows=pack)
bits = ctx.bits(n, for_col=c1)
plan = ctx.plan(And(Leaf(0, F.Gt, 0), Or(Leaf(1, F.Lt, 0), Leaf(2, F.Ge, 0.5))), [c1, c2, c3], aggs=[(R.Sum, 0)])
rec("scan 3 leaves A>0 AND (B<0 OR C>=.5) + sum(A)", scan_ms(plan, bits), n * (24 + 0.125))
plan = ctx.plan(And(Leaf(0, F.Gt, 0)), [c1, c2], aggs=[(R.Sum, 1)])
rec("scan 1 leaf + sum of a separate column", scan_ms(plan, bits), n * (16 + 0.125))
st_ms = timeit(lambda: ctx.reduce(c1, bits), reps=10, warm=3)
rec("knox_reduce masked ALL aggs int64 (incl. sync+D2H)", st_ms, n * 8.125)
c2.free(); c3.free()
# bitset ops
b2 = ctx.bits(n)
ctx.scan(And(Leaf(0, F.Gt, 5)), [c1], bits_out=b2)
nb = n / 8
rec("bits and", timeit(lambda: bits.and_(b2)), 3 * nb)
rec("bits or", timeit(lambda: bits.or_(b2)), 3 * nb)
rec("bits neg", timeit(lambda: bits.neg()), 2 * nb)
rec("bits and_flag (incl. flag D2H sync)", timeit(lambda: bits.and_flag(b2), reps=10), 3 * nb)
rec("bits popcount (incl. D2H sync)", timeit(lambda: bits.count(), reps=10), nb)
rec("bits segment_counts 2^20 (incl. D2H)", timeit(lambda: bits.segment_counts(pack), reps=10), nb)
ctx.scan(And(Leaf(0, F.Range, -(1 << 34), 1 << 34)), [c1], bits_out=bits)   # ~1.6 % density
k = bits.count()
t0 = time.perf_counter(); idx = bits.indexes(); t1 = time.perf_counter()
rec("bits indexes, %.1f%% density, %d hits (incl. D2H to a fresh numpy array)" % (100.0 * k / n, k), (t1 - t0) * 1e3, nb + 4 * k)
pin = ctx.host_array(k + 8, np.uint32)
bits.indexes(out=pin)
t0 = time.perf_counter(); idx = bits.indexes(out=pin); t1 = time.perf_counter()
rec("bits indexes, same, into knox_host_alloc memory (DMA in place)", (t1 - t0) * 1e3, nb + 4 * k)
ctx.host_free(pin)
c1.free()
# bytes EQ
L32 = 32
nrow = 1 << 24
rng = np.random.default_rng(1)
keys = rng.integers(0, 256, size=(4096, L32), dtype=np.uint8)
pick = rng.integers(0, 4096, size=nrow)
buf = keys[pick].reshape(-1)
ptr = (np.arange(nrow, dtype=np.uint64) * np.uint64(L32)) << np.uint64(32) | np.uint64(L32)
bc = ctx.upload_bytes(ptr, buf, pack)
bb = ctx.bits(nrow)
plan = ctx.plan(And(Leaf(0, F.Equal, keys[7].tobytes())), [bc], aggs=[])
def bytes_scan():
    ctx.scan_async(plan, bits_out=bb); ctx.scan_collect(plan, 0)
rec("bytes EQ 32-byte keys, 2^24 rows (whole scan call)", timeit(bytes_scan, reps=10, warm=3), nrow * (8 + L32 + 0.125))
# bytes REGEXP on the same 32-byte binary keys (mostly invalid UTF-8: every byte is decoded) and on ASCII log-like rows
rxb = ctx.bits(nrow)
for pat in ("Hel.*", "^.{3}a", "(?i)[a-f0-9]{4}-\\d+$"):
    rec("bytes REGEXP %-22s 32-byte random keys, 2^24 rows" % repr(pat), timeit(lambda: ctx.match_regexp(bc, pat, bits=rxb), reps=5, warm=2), nrow * (8 + L32 + 0.125))
words = [b"GET /index.html 200", b"POST /api/v1/items 201", b"GET /static/app.js 304", b"Hello World", b"error: timeout id=ab12-77", b"WARN disk 93%"]
lens = np.array([len(w) for w in words])
pickw = rng.integers(0, len(words), size=nrow)
offs = np.concatenate([[0], np.cumsum(lens[pickw])[:-1]]).astype(np.uint64)
ptrw = (offs << np.uint64(32)) | lens[pickw].astype(np.uint64)
table = np.zeros((len(words), lens.max()), dtype=np.uint8)
for i, w in enumerate(words):
    table[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
sel = np.arange(lens.max())[None, :] < lens[pickw][:, None]
bufw = np.concatenate([table[pickw][sel], np.zeros(8, np.uint8)])
wc = ctx.upload_bytes(ptrw, bufw, pack)
wbytes = nrow * 8.125 + int(lens[pickw].sum())
for pat in ("Hel.*", "^GET .* 304$", "(?i)[a-f0-9]{4}-\\d+$", "\\b\\d{3}\\b"):
    rec("bytes REGEXP %-22s ASCII log rows (avg %.0f B), 2^24 rows" % (repr(pat), lens[pickw].mean()), timeit(lambda: ctx.match_regexp(wc, pat, bits=rxb), reps=5, warm=2), wbytes)
# literal per-window reducers (one lane per window)
nt = 1 << 24
tcol = ctx.upload((1_700_000_000 + np.cumsum(rng.integers(0, 3, size=nt))).astype(np.int64), pack)
vcol = ctx.upload(rng.standard_normal(nt), pack)
t_host_from = 1_700_000_000
for nbk in (1 << 10, 1 << 16):
    stepw = int(2 * nt / nbk) + 1
    for fn in (R.Mean, R.StdJoin):
        rec("literal window reducer %-8s 2^24 rows -> %d windows (whole call)" % (fn.name, nbk),
            timeit(lambda: ctx.reduce_buckets_sequential(tcol, vcol, fn, t_host_from, stepw, nbk, assume_sorted=True), reps=3, warm=1), nt * 16)
