#!/usr/bin/env python3
"""Diagnostic (not product): what IS the "placement class" of DESIGN.md 7?

Physical memory is taken chunk by chunk through HIP's virtual-memory API (hipMemCreate: one physical allocation per
chunk, mapped where we choose), so every chunk is a known physical unit, and every (column chunk i, bitset chunk j)
pair is timed with the real scan kernel: 1 GiB of int64 rows from chunk i, the 16 MiB bitset inside chunk j.

  matrix     ratio mix / read-only for all pairs          -> is it a binary class?  is it symmetric / transitive?
  offsets    the bitset window at other offsets of chunk j -> is a physical chunk uniform?
  int32      the same bytes scanned as int32 rows          -> does the read/write address RATIO matter?
  shift      column start shifted inside its chunk         -> does the relative alignment matter?

usage: placement_vmm.py [n_chunks=48] [chunk_gib=2]
"""
import ctypes as C, os, statistics, sys
os.environ["KNOX_PLACEMENT"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
CH = (int(sys.argv[2]) if len(sys.argv) > 2 else 2) << 30
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
gran = C.c_size_t()
chk(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "granularity")
print("allocation granularity (recommended): %d KiB; %d chunks of %d GiB" % (gran.value >> 10, N, CH >> 30))
va = C.c_void_p()
chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(N * CH), C.c_size_t(0), None, C.c_ulonglong(0)), "reserve")
handles = []
for i in range(N):
    h = C.c_void_p()
    rc = hip.hipMemCreate(C.byref(h), C.c_size_t(CH), C.byref(prop), C.c_ulonglong(0))
    if rc != 0:
        print("hipMemCreate stopped at chunk %d (hipError %d)" % (i, rc))
        N = i
        break
    chk(hip.hipMemMap(C.c_void_p(va.value + i * CH), C.c_size_t(CH), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
    handles.append(h)
acc = Access(Loc(1, 0), 3)
chk(hip.hipMemSetAccess(va, C.c_size_t(N * CH), C.byref(acc), C.c_size_t(1)), "set access")
chk(hip.hipMemset(va, 0, C.c_size_t(N * CH)), "memset")
chk(hip.hipDeviceSynchronize(), "sync")

ROWS = 1 << 27                      # 1 GiB of int64 rows, 16 MiB of bitset
PACK = 1 << 20
ctx.L.knox_prof_enable(ctx.h, 1)


def median_ms(plan, bits, reps=8):
    for _ in range(reps):
        ctx.scan_async(plan, bits_out=bits, pack_counts=False)
        ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = sorted(ms[i] for i in range(k.value))
    return statistics.median(v[: max(1, len(v) // 2 + 1)])   # lower half: the clock-ramped launches


def col_at(addr, dtype=BlockType.Int64, rows=ROWS):
    return ctx.wrap(dtype, addr, rows, PACK)


def bits_at(addr, rows=ROWS):
    return K.Bits(ctx, rows, wrap_ptr=addr)


tree = And(Leaf(0, FilterMode.Range, -5, 5))
cols = [col_at(va.value + i * CH) for i in range(N)]
bitsets = [bits_at(va.value + j * CH + CH - (64 << 20)) for j in range(N)]   # the last 64 MiB of the chunk hold the windows
plans_ro = [ctx.plan(tree, [c], aggs=[]) for c in cols]
plans = [ctx.plan(tree, [c], aggs=[(ReducerFunc.Sum, 0)]) for c in cols]
for _ in range(3):
    median_ms(plans[0], bitsets[0], 16)   # clock ramp
ro = [median_ms(p, None) for p in plans_ro]
print("read-only GB/s per chunk: " + " ".join("%4.0f" % (ROWS * 8 / (m * 1e6)) for m in ro))
M = np.zeros((N, N))
for i in range(N):
    for j in range(N):
        M[i, j] = ro[i] / median_ms(plans[i], bitsets[j])
print("\nmatrix: rows = column chunk, columns = bitset chunk; '#' ratio >= 0.93 (fast), '.' < 0.90 (slow), '+' between")
for i in range(N):
    print("%3d %s  min %.3f max %.3f" % (i, "".join("#" if x >= 0.93 else "." if x < 0.90 else "+" for x in M[i]), M[i].min(), M[i].max()))
fast = M >= 0.915
# binary-class model: class of a chunk as a bitset = its column of the matrix; as a column = its row
row_classes = {}
for i in range(N):
    row_classes.setdefault(tuple(fast[i]), []).append(i)
col_classes = {}
for j in range(N):
    col_classes.setdefault(tuple(fast[:, j]), []).append(j)
print("\ndistinct row patterns (chunks that behave alike as COLUMN): %d -> %s" % (len(row_classes), list(row_classes.values())))
print("distinct column patterns (chunks that behave alike as BITSET): %d -> %s" % (len(col_classes), list(col_classes.values())))
print("diagonal (column and bitset in the SAME physical chunk): " + "".join("#" if fast[i, i] else "." for i in range(N)))
print("symmetric: %s" % bool((fast == fast.T).all()))
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "placement_vmm_matrix.npy"), M)

# ---- is one physical chunk uniform?  bitset window at several offsets of chunk j, column = chunk 0 and one chunk of the other kind
others = [i for i in range(N) if tuple(fast[i]) != tuple(fast[0])]
probe_cols = [0] + others[:1]
print("\noffsets: bitset window at offset k/8 of chunk j (k = 0..7), columns %s" % probe_cols)
for j in range(min(N, 12)):
    line = []
    for ci in probe_cols:
        r = []
        for k in range(8):
            b = bits_at(va.value + j * CH + k * (CH // 8) + (CH // 8) - (32 << 20))
            r.append(ro[ci] / median_ms(plans[ci], b))
        line.append("".join("#" if x >= 0.915 else "." for x in r))
    print("  chunk %2d: %s" % (j, "  ".join(line)))

# ---- the same bytes as int32 rows (twice the bits per byte of column): same classes?
print("\nint32 view (2^28 rows over the same 1 GiB, 32 MiB bitset): row = column chunk, first %d bitset chunks" % min(N, 16))
t32 = And(Leaf(0, FilterMode.Range, -5, 5))
for i in range(min(N, 8)):
    c32 = col_at(va.value + i * CH, BlockType.Int32, 1 << 28)
    p_ro = ctx.plan(t32, [c32], aggs=[])
    p = ctx.plan(t32, [c32], aggs=[(ReducerFunc.Sum, 0)])
    r0 = median_ms(p_ro, None)
    row = [r0 / median_ms(p, bits_at(va.value + j * CH + CH - (64 << 20), 1 << 28)) for j in range(min(N, 16))]
    print("%3d %s   int64 row: %s" % (i, "".join("#" if x >= 0.915 else "." for x in row), "".join("#" if x else "." for x in fast[i][:16])))

# ---- relative alignment: shift the column start inside its chunk
print("\nshift: column = chunk i + s bytes, bitset = chunk j window; pairs (0, j_fast), (0, j_slow)")
jf = next((j for j in range(N) if fast[0, j]), None)
js = next((j for j in range(N) if not fast[0, j]), None)
for s in (0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 256 << 20, 512 << 20):
    c = col_at(va.value + s)
    p_ro, p = ctx.plan(tree, [c], aggs=[]), ctx.plan(tree, [c], aggs=[(ReducerFunc.Sum, 0)])
    r0 = median_ms(p_ro, None)
    out = []
    for j in (jf, js):
        out.append("%.3f" % (r0 / median_ms(p, bitsets[j])) if j is not None else "  -  ")
    print("  shift %9d: fast-pair %s  slow-pair %s" % (s, out[0], out[1]))
