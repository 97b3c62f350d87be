#!/usr/bin/env python3
"""The bytes matchers other than EQ on the cfg-4 column shape (32-byte keys, one pool per pack): kernel-ish whole-call
times of LT / RANGE / IN / REGEXP / NE at 2^lg rows.  usage: bench_bytes.py [lg=27]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import FilterMode as F

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n, pack = 1 << lg, 1 << 20
ctx = K.Context(0)
col = ctx.generate_bytes(7, n, pack)
bits = ctx.bits(n)
ptr, buf = ctx.download_bytes_pack(col, 0)          # keys that occur in the column: rows of pack 0
row = lambda i: bytes(buf[i * 32:(i + 1) * 32])
key, key2 = row(5), row(9)
lo, hi = min(key, key2), max(key, key2)
keys = sorted(set(row(i) for i in range(0, 4096, 7)))


def timed(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for name, tree in (("EQ", And(Leaf(0, F.Equal, key))), ("NE", And(Leaf(0, F.NotEqual, key))), ("LT", And(Leaf(0, F.Lt, key))),
                   ("RANGE", And(Leaf(0, F.Range, lo, hi))), ("IN (%d keys)" % len(keys), And(Leaf(0, F.In, keys)))):
    ms = timed(lambda: ctx.scan(tree, [col], bits_out=bits))
    print("%-16s %8.3f ms  %7.1f GB/s (40.125 B/row)" % (name, ms, n * 40.125 / ms / 1e6), flush=True)
ms = timed(lambda: ctx.match_regexp(col, "^\\x00.*\\xff$", bits=bits))
print("%-16s %8.3f ms  %7.1f GB/s" % ("REGEXP", ms, n * 40.125 / ms / 1e6), flush=True)
