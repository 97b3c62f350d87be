#!/usr/bin/env python3
"""Per-launch kernel time series of the headline scan (diagnostic): shows warm-up / DPM effects."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--sync", type=int, default=1)
ap.add_argument("--packs", type=int, default=1)
ap.add_argument("--sleep-us", type=int, default=0)
ap.add_argument("--pipe", type=int, default=0)
ap.add_argument("--torch", type=int, default=0)
a = ap.parse_args()
n = 1 << 30
if a.torch:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
ctx = K.Context(0)
col = ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20)
bits = ctx.bits(n)
plan = ctx.plan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [col], aggs=[(ReducerFunc.Sum, 0)])
ctx.L.knox_prof_enable(ctx.h, 1)
t0 = time.perf_counter()
if a.pipe:
    ctx.scan_async(plan, bits_out=bits, pack_counts=bool(a.packs))
    for i in range(a.steps - 1):
        ctx.scan_async(plan, bits_out=bits, pack_counts=bool(a.packs))
        ctx.scan_collect(plan, 1024 if a.packs else 0)
    ctx.scan_collect(plan, 1024 if a.packs else 0)
else:
    inflight = 0
    for i in range(a.steps):
        ctx.scan_async(plan, bits_out=bits, pack_counts=bool(a.packs))
        inflight += 1
        if a.sync or inflight == 4:
            ctx.scan_collect(plan, 1024 if a.packs else 0)
            inflight -= 1
        if a.sleep_us:
            time.sleep(a.sleep_us * 1e-6)
    while inflight:
        ctx.scan_collect(plan, 1024 if a.packs else 0)
        inflight -= 1
ctx.sync()
wall = time.perf_counter() - t0
ms = (C.c_float * 512)()
k = C.c_int()
ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
v = [ms[i] for i in range(k.value)]
print("sync=%d packs=%d steps=%d wall/step=%.4f ms" % (a.sync, a.packs, a.steps, wall / a.steps * 1e3))
import statistics
print("median %.4f  min %.4f  first10 %s  last10 %s" % (statistics.median(v), min(v), " ".join("%.3f" % x for x in v[:10]), " ".join("%.3f" % x for x in v[-10:])))
