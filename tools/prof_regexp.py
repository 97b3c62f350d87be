#!/usr/bin/env python3
"""Kernel-only view of the bytes matchers (run under rocprofv3 --kernel-trace --stats): 2^24 log-like rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import FilterMode as F

ctx = K.Context(0)
nrow = 1 << 24
rng = np.random.default_rng(1)
words = [b"GET /index.html 200", b"POST /api/v1/items 201", b"GET /static/app.js 304", b"Hello World", b"error: timeout id=ab12-77", b"WARN disk 93%"]
lens = np.array([len(w) for w in words])
pick = rng.integers(0, len(words), size=nrow)
offs = np.concatenate([[0], np.cumsum(lens[pick])[:-1]]).astype(np.uint64)
ptr = (offs << np.uint64(32)) | lens[pick].astype(np.uint64)
table = np.zeros((len(words), lens.max()), dtype=np.uint8)
for i, w in enumerate(words):
    table[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
sel = np.arange(lens.max())[None, :] < lens[pick][:, None]
buf = np.concatenate([table[pick][sel], np.zeros(8, np.uint8)])
col = ctx.upload_bytes(ptr, buf, 1 << 20)
bits = ctx.bits(nrow)
for pat in ("Hel.*", "^GET .* 304$", "(?i)[a-f0-9]{4}-\\d+$", "\\b\\d{3}\\b"):
    for _ in range(5):
        ctx.match_regexp(col, pat, bits=bits)
for _ in range(5):
    ctx.scan(And(Leaf(0, F.Equal, b"Hello World")), [col], bits_out=bits)
ctx.sync()
print("bytes per row incl. ptr:", 8 + lens[pick].mean())
