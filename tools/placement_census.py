#!/usr/bin/env python3
"""Diagnostic (not product): census of the placement classes over (nearly) the whole HBM, and the cost of VMM mappings.

1. takes physical memory in 2 GiB hipMemCreate chunks until `max_gib` is reached, classifies every chunk -- both
   halves -- by pair probes against one representative per class (a chunk is of class c when scanning it into a bitset
   inside c's representative chunk is slow: tools/placement_vmm.py showed slow <=> same class), prints the class
   sequence in allocation order and the bytes per class;
2. builds an 8 GiB VA-contiguous column out of four chunks of one class and times the headline scan with the bitset in
   (a) another class, (b) the same class; the same with plain hipMalloc memory beside it (is a VMM mapping slower?).

usage: placement_census.py [max_gib=256]
"""
import ctypes as C, os, statistics, sys, time
os.environ["KNOX_PLACEMENT"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

MAX_GIB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
free_b, total_b = C.c_size_t(), C.c_size_t()
chk(hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)), "meminfo")
n_max = min(MAX_GIB << 30, free_b.value - (12 << 30)) // CH
print("device memory: %.1f GiB free of %.1f GiB; taking up to %d chunks of 2 GiB" % (free_b.value / 2**30, total_b.value / 2**30, n_max))
va = C.c_void_p()
chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(n_max * CH), C.c_size_t(1 << 30), None, C.c_ulonglong(0)), "reserve")
t0 = time.perf_counter()
N = 0
for i in range(n_max):
    h = C.c_void_p()
    if hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0)) != 0:
        break
    chk(hip.hipMemMap(C.c_void_p(va.value + i * CH), C.c_size_t(CH), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
    N += 1
chk(hip.hipMemSetAccess(va, C.c_size_t(N * CH), C.byref(acc), C.c_size_t(1)), "set access")
t_map = time.perf_counter() - t0
chk(hip.hipMemset(va, 0, C.c_size_t(N * CH)), "memset")
chk(hip.hipDeviceSynchronize(), "sync")
print("%d chunks created + mapped in %.2f s (%.2f ms per chunk)" % (N, t_map, t_map / max(1, N) * 1e3))

PACK = 1 << 20
ctx.L.knox_prof_enable(ctx.h, 1)
tree = And(Leaf(0, FilterMode.Range, -5, 5))


def median_ms(plan, bits, reps=8):
    for _ in range(reps):
        ctx.scan_async(plan, bits_out=bits, pack_counts=False)
        ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = sorted(ms[i] for i in range(k.value))
    return statistics.median(v[: max(1, len(v) // 2 + 1)])


def ratio(col_addr, bits_addr, rows=1 << 27):
    c = ctx.wrap(BlockType.Int64, col_addr, rows, PACK)
    b = K.Bits(ctx, rows, wrap_ptr=bits_addr)
    r = median_ms(ctx.plan(tree, [c], aggs=[]), None) / median_ms(ctx.plan(tree, [c], aggs=[(ReducerFunc.Sum, 0)]), b)
    c.free()
    b.free()
    return r


for _ in range(4):
    ratio(va.value, va.value + CH - (64 << 20))   # clock ramp
reps = []       # chunk index of each class's representative
cls = []
t0 = time.perf_counter()
for i in range(N):
    halves = []
    for half in (0, 1):
        got = None
        for c, r in enumerate(reps):
            if ratio(va.value + i * CH + half * (CH // 2), va.value + r * CH + CH - (64 << 20), 1 << 26) < 0.915:
                got = c
                break
        if got is None and half == 0:
            reps.append(i)
            got = len(reps) - 1
        halves.append(got)
    cls.append(halves[0] if halves[0] == halves[1] else -1)
t_cls = time.perf_counter() - t0
print("classified in %.2f s (%.1f ms per chunk); %d classes, representatives %s" % (t_cls, t_cls / max(1, N) * 1e3, len(reps), reps))
print("class sequence in allocation order ('?' = the two halves differ):")
print("".join("?" if c < 0 else chr(ord("A") + c) for c in cls))
for c in range(len(reps)):
    print("  class %s: %3d chunks = %5.1f GiB" % (chr(ord("A") + c), cls.count(c), cls.count(c) * 2.0))
print("  mixed  : %3d chunks" % cls.count(-1))

# ---- an 8 GiB column out of four same-class chunks (VA-contiguous by mapping), bitset in another / the same class
by_class = {c: [i for i in range(N) if cls[i] == c] for c in range(len(reps))}
big = max(by_class, key=lambda c: len(by_class[c]))
other = next(c for c in by_class if c != big and by_class[c])
# a run of 5 consecutive chunks of class `big` in the mapping is VA-contiguous already
run = next((s for s in range(N - 4) if all(cls[s + k] == big for k in range(5))), None)
ROWS = 1 << 30
if run is not None:
    col = ctx.wrap(BlockType.Int64, va.value + run * CH, ROWS, PACK)
    p_ro, p = ctx.plan(tree, [col], aggs=[]), ctx.plan(tree, [col], aggs=[(ReducerFunc.Sum, 0)])
    for name, jb in (("other class", by_class[other][0]), ("same class", run + 4)):
        b = K.Bits(ctx, ROWS, wrap_ptr=va.value + jb * CH + (256 << 20))
        for _ in range(3):
            median_ms(p, b, 12)
        ro, mix = median_ms(p_ro, None, 12), median_ms(p, b, 12)
        print("VMM 8 GiB column (chunks %d..%d, class %s), bitset in %s: read-only %.4f ms = %.0f GB/s, scan %.4f ms = %.0f GB/s, ratio %.3f"
              % (run, run + 3, chr(ord("A") + big), name, ro, ROWS * 8 / ro / 1e6, mix, ROWS * 8.125 / mix / 1e6, ro / mix))
# plain hipMalloc memory beside it (whatever classes the driver gives)
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
pm, pb = C.c_void_p(), C.c_void_p()
if hip.hipMalloc(C.byref(pm), (8 << 30) + 512) == 0 and hip.hipMalloc(C.byref(pb), (128 << 20) + 512) == 0:
    hip.hipMemset(pm, 0, C.c_size_t(8 << 30))
    hip.hipMemset(pb, 0, C.c_size_t(128 << 20))
    col = ctx.wrap(BlockType.Int64, pm.value, ROWS, PACK)
    b = K.Bits(ctx, ROWS, wrap_ptr=pb.value)
    p_ro, p = ctx.plan(tree, [col], aggs=[]), ctx.plan(tree, [col], aggs=[(ReducerFunc.Sum, 0)])
    for _ in range(3):
        median_ms(p, b, 12)
    ro, mix = median_ms(p_ro, None, 12), median_ms(p, b, 12)
    print("hipMalloc 8 GiB column + hipMalloc bitset: read-only %.4f ms = %.0f GB/s, scan %.4f ms = %.0f GB/s, ratio %.3f"
          % (ro, ROWS * 8 / ro / 1e6, mix, ROWS * 8.125 / mix / 1e6, ro / mix))
    for name, jb in (("VMM class " + chr(ord("A") + big), by_class[big][0]), ("VMM class " + chr(ord("A") + other), by_class[other][0])):
        b2 = K.Bits(ctx, ROWS, wrap_ptr=va.value + jb * CH + (256 << 20))
        mix = median_ms(p, b2, 12)
        print("hipMalloc column, bitset in %s: scan %.4f ms = %.0f GB/s, ratio %.3f" % (name, mix, ROWS * 8.125 / mix / 1e6, ro / mix))
