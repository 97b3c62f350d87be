#!/usr/bin/env python3
"""Generate XXH3-64 (seed 0) golden vectors for byte strings with the python `xxhash`
package (an independent binding of upstream xxHash), because the reference's []byte hash is
the un-vendored dependency github.com/zeebo/xxh3 v1.1.0 (internal/hash/hash.go:26, go.mod:18)
and the reference's own tests pin only the 4- and 8-byte inputs (internal/hash/xxh3_test.go).

Input rule (reproduced by the test, no payload stored): data(len) = bytes((i*131 + 7) & 0xff
for i in range(len)).  Output: tests/golden/xxh3_bytes.json  {len: digest}.
Run from the repo root:  python tests/golden/make_xxh3_bytes_golden.py
"""
import json
import os

import xxhash

HERE = os.path.dirname(os.path.abspath(__file__))


def data(n):
    return bytes((i * 131 + 7) & 0xFF for i in range(n))


lens = list(range(0, 260)) + [511, 512, 513, 1023, 1024, 1025, 2048, 4095, 4096, 4097, 10000]
out = {str(n): xxhash.xxh3_64_intdigest(data(n)) for n in lens}
with open(os.path.join(HERE, "xxh3_bytes.json"), "w") as fh:
    json.dump({"generator": "python xxhash %s (xxh3_64_intdigest, seed 0)" % xxhash.VERSION,
               "input_rule": "bytes((i*131+7)&0xff for i in range(len))", "digests": out}, fh)
print("wrote", len(out), "digests")
