#!/usr/bin/env python3
"""Diagnostic: does the cache policy of the column loads change the read+write placement effect?

The product library generates `ncol` 2^lg-row int64 columns; every diagnostics build
(knoxdb_amd/diag/libknoxgpu_p<k>.so, `make -C knoxdb_amd/csrc diag`) then scans THE SAME device
buffers (knox_col_wrap) so that all policies see identical physical placements.
  p0 product (nontemporal builtin)   p1 plain   p2 sc0 sc1   p3 sc0 sc1 nt   p4 sc1   p5 sc1 nt   p6 sc0   p7 nt (asm)
usage: load_policy_matrix.py [lg=28] [ncol=12]
"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
import knoxdb_amd._lib as LL
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n, pack = 1 << lg, 1 << 20
base = LL.lib()
ctx0 = K.Context(0)
cols0 = [ctx0.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack) for _ in range(ncol)]
ptrs = [c.device_ptr for c in cols0]
ctx0.sync()


def load(path):
    L = C.CDLL(path)
    for name, (res, args) in LL.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L


def run(ctx, col, bits, steps, with_bits=True):
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [col], aggs=[(ReducerFunc.Sum, 0)])
    b = bits if with_bits else None
    ctx.scan_async(plan, bits_out=b)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=b)
        ctx.scan_collect(plan, 0)
    res = ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)()
    k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = [ms[i] for i in range(k.value)]
    return n * (8.125 if with_bits else 8.0) / statistics.median(v[len(v) // 2:]) / 1e6, res.count


diag = os.path.join(os.path.dirname(LL.SO_PATH), "diag")
libs = [(0, base)] + [(p, load(os.path.join(diag, "libknoxgpu_p%d.so" % p))) for p in range(1, 8)
                      if os.path.exists(os.path.join(diag, "libknoxgpu_p%d.so" % p))]
print("policy | " + " ".join("c%-5d" % i for i in range(ncol)) + " |    min    med | read-only med")
want = None
for p, L in libs:
    LL._lib = L
    ctx = K.Context(0)
    ctx.L.knox_prof_enable(ctx.h, 1)
    cols = [ctx.wrap(BlockType.Int64, ptr, n, pack) for ptr in ptrs]
    bits = ctx.bits(n)
    run(ctx, cols[0], bits, 40)
    row, ro = [], []
    for c in cols:
        g, cnt = run(ctx, c, bits, 14)
        want = cnt if want is None else want
        assert cnt == want, "policy %d changed the result" % p
        row.append(g)
    for c in cols[:4]:
        ro.append(run(ctx, c, bits, 14, with_bits=False)[0])
    print("p%d     | " % p + " ".join("%6.0f" % x for x in row) + " | %6.0f %6.0f | %6.0f" % (min(row), statistics.median(row), statistics.median(ro)))
    for c in cols:
        c.free()
    bits.free()
    ctx.close()
LL._lib = base
for c in cols0:
    c.free()
ctx0.close()
