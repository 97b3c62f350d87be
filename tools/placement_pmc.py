#!/usr/bin/env python3
"""Diagnostic workload for PMC comparison of slow vs fast column placements: N columns, S scans each.
Prints per-column median kernel ms (line 'COL i ms'); run under rocprofv3 --pmc to get per-dispatch counters."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType, FilterMode, ReducerFunc
ncol, steps = int(sys.argv[1]), int(sys.argv[2])
n = 1 << 28
ctx = K.Context(0)
cols = [ctx.generate(BlockType.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=1 << 20) for _ in range(ncol)]
bits = ctx.bits(n)
ctx.L.knox_prof_enable(ctx.h, 1)
for ci, col in enumerate(cols):
    plan = ctx.plan(And(Leaf(0, FilterMode.Range, -(1 << 38), 1 << 38)), [col], aggs=[(ReducerFunc.Sum, 0)])
    for i in range(steps):
        ctx.scan_async(plan, bits_out=bits); ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k))
    v = [ms[i] for i in range(k.value)]
    print("COL %d %.4f" % (ci, statistics.median(v[len(v) // 2:])))
