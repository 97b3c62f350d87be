import ctypes as C, os, statistics, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R
n, pack = 1 << 30, 1 << 20
ctx = K.Context(0)
def rate(bits, plan, bpr, steps=14):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(plan, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits); ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k)); ctx.L.knox_prof_enable(ctx.h, 0)
    v = [ms[i] for i in range(k.value)]
    return n * bpr / statistics.median(v[len(v) // 2:]) / 1e6
col = ctx.generate(B.Int64, 0, 2, n, lo=-(1 << 40), span=1 << 41, pack_rows=pack)
bits = ctx.bits(n, for_col=col)
plan = ctx.plan(And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), [col], aggs=[(R.Sum, 0)])
print("pair check", bits.pair_check, ctx.arena_info())
ro = ctx.plan(And(Leaf(0, F.Range, -(1 << 38), 1 << 38)), [col], aggs=[(R.Sum, 0)])
print("default %.0f  read-only %.0f" % (rate(bits, plan, 8.125, 40), rate(None, ro, 8.0, 20)))
for it in (1, 2, 4):
    os.environ["KNOX_ITERS"] = str(it)
    print("iters", it, " ".join("b%d:%5.0f" % (b, (ctx.set_tuning(b), rate(bits, plan, 8.125))[1]) for b in (2, 4, 16, 64, 256, 1024, 4096)))
