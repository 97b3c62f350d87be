#!/usr/bin/env python3
"""Round-trip latency of the finest boundary, knox_cmp (the cmp func vars of internal/cmp/cmp.go:6-114: host slice in,
host bitset out, synchronous), for int64 Between at n = 2^12 ... 2^24 rows, beside the AVX2 port of the reference's
kernel (oracle/knox_cpu_simd.c, ONE thread: a cmp call runs on the calling goroutine) on the same box.

  pageable     the source is ordinary host memory (a Go slice): host copy into the pinned ring, DMA, scan from HBM
  page-locked  the source lies in knox_host_alloc memory (what the Go arena hook hands out): the kernel reads it in place
  cpu_avx2     compare + movemask + popcount (+ a masked sum: the port fuses it; the compare alone is not slower)

The crossover of the page-locked column against cpu_avx2 is the row floor of the Go rebinding (integration/go/cmp_knoxgpu.go
knoxgpu.CmpMinRows); below it bind1 / bind2 call the kernel that was installed before.  Prints a table."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import knoxdb_amd as K
from knoxdb_amd.types import BlockType, FilterMode
from oracle import oracle as O          # the checker / CPU baseline: a tool, not the product

ctx = K.Context(0)
rng = np.random.default_rng(1)
a, b = -(1 << 38), 1 << 38
print("%8s %14s %14s %14s %12s %12s" % ("rows", "pageable us", "page-locked us", "cpu_avx2 us", "gpu/cpu (pl)", "G rows/s (pl)"))
floor = None
for lg in range(12, 25):
    n = 1 << lg
    src = rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64)
    pin = ctx.host_array(n, np.int64)
    pin[:] = src
    reps = 200 if lg <= 18 else 40 if lg <= 21 else 10

    def t(fn):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e6

    nb = (n + 7) >> 3
    bits = np.zeros(nb + 32, dtype=np.uint8)
    av, bv = np.array([a], dtype=np.int64), np.array([b], dtype=np.int64)
    L, h = ctx.L, ctx.h
    import ctypes as C
    call = lambda arr: L.knox_cmp(h, int(BlockType.Int64), int(FilterMode.Range), arr.ctypes.data_as(C.c_void_p), n, av.ctypes.data_as(C.c_void_p),
                                  bv.ctypes.data_as(C.c_void_p), bits.ctypes.data_as(C.c_void_p))
    c1 = call(src)
    want, wc = O.cmp(O.I64, O.RG, src, a, b)
    assert c1 == wc and (bits[:nb] == want).all(), "pageable result != oracle"
    bits[:] = 0
    c2 = call(pin)
    assert c2 == wc and (bits[:nb] == want).all(), "page-locked result != oracle"
    us_page, us_pin = t(lambda: call(src)), t(lambda: call(pin))
    cb = np.zeros(nb, dtype=np.uint8)
    us_cpu = t(lambda: O.simd_scan_i64_bw_sum(src, n, a, b, 1, bits=cb, level=2))
    if floor is None and us_pin < us_cpu:
        floor = lg
    print("%8s %14.1f %14.1f %14.1f %12.2f %12.2f" % ("2^%d" % lg, us_page, us_pin, us_cpu, us_pin / us_cpu, n / us_pin / 1e3), flush=True)
    ctx.host_free(pin)
print("first size at which the page-locked GPU call beats one AVX2 core: %s" % ("2^%d rows" % floor if floor else "none up to 2^24 rows"))
