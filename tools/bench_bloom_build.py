import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import knoxdb_amd as K
from knoxdb_amd.types import BlockType as B
ctx = K.Context(0)
n, pack = 1 << 30, 1 << 20
col = ctx.generate(B.Uint64, 0, 5, n, pack_rows=pack)
for i in range(3):
    t0 = time.perf_counter(); b = ctx.bloom_build(col, 2); ctx.sync(); t1 = time.perf_counter()
    print("uint64 build %d: %.1f ms" % (i, (t1 - t0) * 1e3)); b.free()
col.free()
bc = ctx.generate_bytes(7, n, pack)
for i in range(3):
    t0 = time.perf_counter(); b = ctx.bloom_build(bc, 2); ctx.sync(); t1 = time.perf_counter()
    print("bytes build %d: %.1f ms" % (i, (t1 - t0) * 1e3)); b.free()
