#!/usr/bin/env python3
"""Diagnostic (not product): how fine-grained are the placement classes inside one 2 GiB hipMemCreate chunk?
Every chunk is cut into 16 windows of 128 MiB; each window is scanned (as an int64 column) into the representative
bitset windows of the first chunks that turned out to be of different classes; one character per window:
  '0'..'9' = slow with representative k (same class), '.' = fast with all, '?' = in between.
usage: placement_finemap.py [n_chunks=32] [win_mib=128]"""
import ctypes as C, os, statistics, sys
os.environ["KNOX_PLACEMENT"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
WIN = (int(sys.argv[2]) if len(sys.argv) > 2 else 128) << 20
CH = 2 << 30
hip = C.CDLL("libamdhip64.so")


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", Loc), ("win32", C.c_void_p),
                ("compression", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def chk(rc, what):
    if rc != 0:
        raise SystemExit("%s failed: hipError %d" % (what, rc))


ctx = K.Context(0)
prop = Prop(1, 0, Loc(1, 0), None, 0, 0, 0)
acc = Access(Loc(1, 0), 3)
va = C.c_void_p()
chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(N * CH), C.c_size_t(1 << 30), None, C.c_ulonglong(0)), "reserve")
for i in range(N):
    h = C.c_void_p()
    chk(hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0)), "create")
    chk(hip.hipMemMap(C.c_void_p(va.value + i * CH), C.c_size_t(CH), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
chk(hip.hipMemSetAccess(va, C.c_size_t(N * CH), C.byref(acc), C.c_size_t(1)), "access")
chk(hip.hipMemset(va, 0, C.c_size_t(N * CH)), "memset")
chk(hip.hipDeviceSynchronize(), "sync")
ctx.L.knox_prof_enable(ctx.h, 1)
tree = And(Leaf(0, FilterMode.Range, -5, 5))
PACK = 1 << 20


def median_ms(plan, bits, reps):
    for _ in range(reps):
        ctx.scan_async(plan, bits_out=bits, pack_counts=False)
        ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = sorted(ms[i] for i in range(k.value))
    return statistics.median(v[: max(1, len(v) // 2 + 1)])


def ratios(col_addr, nbytes, bit_addrs, reps=12):
    rows = nbytes // 8
    c = ctx.wrap(BlockType.Int64, col_addr, rows, PACK)
    p_ro, p = ctx.plan(tree, [c], aggs=[]), ctx.plan(tree, [c], aggs=[(ReducerFunc.Sum, 0)])
    ro = median_ms(p_ro, None, reps)
    out = []
    for ba in bit_addrs:
        b = K.Bits(ctx, rows, wrap_ptr=ba)
        out.append(ro / median_ms(p, b, reps))
        b.free()
    c.free()
    return out


def rep_addr(chunk):   # a 16 MiB bitset window in the middle of the chunk's first 128 MiB
    return va.value + chunk * CH + (64 << 20)


for _ in range(4):
    ratios(va.value, 1 << 30, [rep_addr(1)])
# representatives: greedily, a chunk whose first GiB is fast with every representative so far
reps = [0]
for i in range(1, N):
    r = ratios(va.value + i * CH + (1 << 30), 1 << 30, [rep_addr(k) for k in reps])
    if min(r) > 0.955 and len(reps) < 6:
        reps.append(i)
print("representatives (chunk index): %s" % reps)
print("chunk: 16 windows of %d MiB   |   first GiB / second GiB vs every representative" % (WIN >> 20))
for i in range(N):
    line = ""
    for w in range(CH // WIN):
        r = ratios(va.value + i * CH + w * WIN, WIN, [rep_addr(k) for k in reps], reps=16)
        slow = [k for k, x in enumerate(r) if x < 0.915]
        mid = [k for k, x in enumerate(r) if 0.915 <= x < 0.95]
        line += str(slow[0]) if len(slow) == 1 and not mid else "." if not slow and not mid else "?"
    g = [ratios(va.value + i * CH + h * (1 << 30), 1 << 30, [rep_addr(k) for k in reps]) for h in (0, 1)]
    print("%3d: %s   |   %s  /  %s" % (i, line, " ".join("%.2f" % x for x in g[0]), " ".join("%.2f" % x for x in g[1])))
