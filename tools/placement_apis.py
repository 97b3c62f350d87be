#!/usr/bin/env python3
"""Diagnostic (not product): does any allocation API / flag hand out memory from ANOTHER placement class than the
process's first hipMemCreate chunks?  (Consecutive chunks come from one long run of one class; a write pool in another
class needs memory from somewhere else.)  Each candidate buffer is used as the bitset of a scan over a 1 GiB VMM column
and as a 1 GiB column scanned into a VMM bitset window; fast (>= 0.94 of read-only) = another class."""
import ctypes as C, os, statistics, sys
os.environ["KNOX_PLACEMENT"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
GIB = 1 << 30
hip = C.CDLL("libamdhip64.so")
class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", Loc), ("win32", C.c_void_p), ("compression", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]
class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]
ctx = K.Context(0)
acc = Access(Loc(1, 0), 3)
def vmm(nbytes, handle_type=0):
    prop = Prop(1, handle_type, Loc(1, 0), None, 0, 0, 0)
    h = C.c_void_p(); va = C.c_void_p()
    if hip.hipMemCreate(C.byref(h), C.c_size_t(nbytes), C.byref(prop), C.c_ulonglong(0)) != 0: return None
    if hip.hipMemAddressReserve(C.byref(va), C.c_size_t(nbytes), C.c_size_t(GIB), None, C.c_ulonglong(0)) != 0: return None
    if hip.hipMemMap(va, C.c_size_t(nbytes), C.c_size_t(0), h, C.c_ulonglong(0)) != 0: return None
    if hip.hipMemSetAccess(va, C.c_size_t(nbytes), C.byref(acc), C.c_size_t(1)) != 0: return None
    return va.value
def plain(nbytes):
    p = C.c_void_p()
    return p.value if hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0 else None
def ext(nbytes, flags):
    p = C.c_void_p()
    return p.value if hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(flags)) == 0 else None
def async_(nbytes):
    p = C.c_void_p()
    rc = hip.hipMallocAsync(C.byref(p), C.c_size_t(nbytes), None)
    hip.hipDeviceSynchronize()
    return p.value if rc == 0 else None
ctx.L.knox_prof_enable(ctx.h, 1)
tree = And(Leaf(0, FilterMode.Range, -5, 5))
def median_ms(plan, bits, reps=12):
    for _ in range(reps):
        ctx.scan_async(plan, bits_out=bits, pack_counts=False); ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = sorted(ms[i] for i in range(k.value)); return statistics.median(v[: max(1, len(v) // 2 + 1)])
def ratio(col_addr, bits_addr):
    rows = 1 << 27
    c = ctx.wrap(BlockType.Int64, col_addr, rows, 1 << 20); b = K.Bits(ctx, rows, wrap_ptr=bits_addr)
    r = median_ms(ctx.plan(tree, [c], aggs=[]), None) / median_ms(ctx.plan(tree, [c], aggs=[(ReducerFunc.Sum, 0)]), b)
    c.free(); b.free(); return r
base = [vmm(GIB) for _ in range(3)]
for a in base: hip.hipMemset(C.c_void_p(a), 0, C.c_size_t(GIB))
hip.hipDeviceSynchronize()
for _ in range(4): ratio(base[1], base[0])
print("baseline: VMM chunk 1 -> window in VMM chunk 0: %.3f ; chunk 2 -> chunk 0: %.3f ; chunk 1 -> chunk 2: %.3f" % (ratio(base[1], base[0]), ratio(base[2], base[0]), ratio(base[1], base[2])))
# small physical allocations: served from the driver's leftover fragments (wherever those lie) before the big free run?
cols = [vmm(GIB) for _ in range(4)]
for a_ in cols: hip.hipMemset(C.c_void_p(a_), 0, C.c_size_t(GIB))
hip.hipDeviceSynchronize()
print("four more 1 GiB chunks as columns into window 0: " + " ".join("%.3f" % ratio(c, base[0]) for c in cols))
for size_mib in (32, 128, 256):
    line = []
    for i in range(12):
        p = vmm(size_mib << 20)
        if p is None:
            line.append("fail"); continue
        hip.hipMemset(C.c_void_p(p), 0, C.c_size_t(size_mib << 20)); hip.hipDeviceSynchronize()
        line.append("%.3f" % ratio(cols[-1], p))
    print("12 x hipMemCreate(%3d MiB) as the bitset of the last 1 GiB column: %s" % (size_mib, " ".join(line)))
line = []
for i in range(12):
    p = plain(64 << 20)
    hip.hipMemset(C.c_void_p(p), 0, C.c_size_t(64 << 20)); hip.hipDeviceSynchronize()
    line.append("%.3f" % ratio(cols[-1], p))
print("12 x hipMalloc(64 MiB) as the bitset of the last 1 GiB column: %s" % " ".join(line))
