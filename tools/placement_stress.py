#!/usr/bin/env python3
"""Diagnostic: fill the device with 8 GiB columns with placement on: time per allocation, class of each
column, held memory; everything must come back after the frees."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd.types import BlockType

hip = C.CDLL("libamdhip64.so")
free0, free1, tot = C.c_size_t(), C.c_size_t(), C.c_size_t()
ctx = K.Context(0)
hip.hipMemGetInfo(C.byref(free0), C.byref(tot))
print("free %.1f GiB of %.1f" % (free0.value / 2**30, tot.value / 2**30), flush=True)
cols = []
n = 1 << 30
while True:
    t0 = time.perf_counter()
    try:
        c = ctx.generate(BlockType.Int64, 3, 0, n, lo=0, pack_rows=1 << 20)
    except K.KnoxError as e:
        print("stop: %s after %d columns (%.1f s)" % (e, len(cols), time.perf_counter() - t0), flush=True)
        break
    ctx.sync()
    g, k, r = c.placement
    info = ctx.placement_info()
    cols.append(c)
    print("col %2d: %6.2f s  probe %5.0f GB/s  mix/ro %.3f  candidates %2d  held %5.1f GiB" %
          (len(cols), time.perf_counter() - t0, g, r, k, info["held_bytes"] / 2**30), flush=True)
    if len(cols) >= 40:
        break
for c in cols:
    c.free()
ctx.close()
hip.hipMemGetInfo(C.byref(free1), C.byref(tot))
print("free after close %.1f GiB (before %.1f)" % (free1.value / 2**30, free0.value / 2**30))
