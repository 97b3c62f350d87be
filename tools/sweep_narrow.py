import ctypes as C, os, statistics, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R
n, pack = 1 << 30, 1 << 20
ctx = K.Context(0)
def rate(col, bits, plan, bpr, steps=14):
    ctx.L.knox_prof_enable(ctx.h, 1)
    ctx.scan_async(plan, bits_out=bits)
    for _ in range(steps - 1):
        ctx.scan_async(plan, bits_out=bits); ctx.scan_collect(plan, 0)
    ctx.scan_collect(plan, 0)
    ms = (C.c_float * 512)(); k = C.c_int()
    ctx.L.knox_prof_read_each(ctx.h, ms, 512, C.byref(k)); ctx.L.knox_prof_enable(ctx.h, 0)
    v = [ms[i] for i in range(k.value)]
    return n * bpr / statistics.median(v[len(v) // 2:]) / 1e6
ref = ctx.generate(B.Int64, 0, 2, 1 << 28, lo=0, span=100, pack_rows=pack)
bits = ctx.bits(n, for_col=ref)
for bt, w, lo, span, a, b in ((B.Uint8, 1, 0, 256, 64, 127), (B.Int16, 2, -(1 << 15), 1 << 16, -(1 << 13), 1 << 13), (B.Int32, 4, -(1 << 30), 1 << 31, -(1 << 28), 1 << 28)):
    col = ctx.generate(bt, 0, 2, n, lo=lo, span=span, pack_rows=pack)
    plan = ctx.plan(And(Leaf(0, F.Range, a, b)), [col], aggs=[(R.Sum, 0)])
    rate(col, bits, plan, w + .125, 40)
    for it in (1, 2, 4):
        os.environ["KNOX_ITERS"] = str(it)
        row = []
        for bpc in (2, 3, 4, 6, 8):
            ctx.set_tuning(bpc)
            row.append(rate(col, bits, plan, w + .125))
        print(bt.name, "iters", it, " ".join("b%d:%5.0f" % (b_, r) for b_, r in zip((2, 3, 4, 6, 8), row)), "")
    col.free()
