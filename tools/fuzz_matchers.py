#!/usr/bin/env python3
"""Randomised differential run of the matchers whose kernels were rewritten in round 4 -- numeric IN / NOT IN (every lookup tier),
bit-packed scans (LDS-staged and general form), bytes IN (prefix search), time windows (the three launch forms; out-of-order clocks
anywhere in int64 with steps up to 2^44, with and without the compaction pass, against integer floor division) -- against the
oracle, with random sizes, widths, set sizes, value ranges and ragged tails.  Not part of the test suite (minutes of oracle time);
run after touching one of those kernels.  usage: fuzz_matchers.py [seconds=60] [seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import knoxdb_amd as K
from knoxdb_amd import And, Leaf
from knoxdb_amd.types import BlockType as B, FilterMode as F, ReducerFunc as R
from oracle import oracle as O
from helpers import oracle_match

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = K.Context(0)
NP = {B.Int64: np.int64, B.Uint64: np.uint64, B.Int32: np.int32, B.Uint32: np.uint32, B.Int16: np.int16, B.Uint16: np.uint16, B.Int8: np.int8,
      B.Uint8: np.uint8, B.Float64: np.float64, B.Float32: np.float32}
OT = {B.Int64: O.I64, B.Uint64: O.U64, B.Int32: O.I32, B.Uint32: O.U32, B.Int16: O.I16, B.Uint16: O.U16, B.Int8: O.I8, B.Uint8: O.U8}
t_end = time.time() + budget
n_cases = {"inset": 0, "packed": 0, "bytes_in": 0, "windows": 0, "shuffled": 0}


def rand_values(bt, n):
    dt = NP[bt]
    if np.dtype(dt).kind == "f":
        x = (rng.integers(-5000, 5000, size=n) / 8.0).astype(dt)
        x[rng.integers(0, n, 3)] = np.nan
        x[rng.integers(0, n, 3)] = -0.0
        return x
    info = np.iinfo(dt)
    kind = rng.integers(0, 3)
    if kind == 0:
        return rng.integers(info.min, info.max, size=n, endpoint=True, dtype=dt)
    if kind == 1 and np.dtype(dt).itemsize == 8:
        return (rng.integers(0, 1 << 62, size=n, dtype=np.int64)).astype(dt)
    lo = int(info.min) + int(rng.integers(0, min(int(info.max) - int(info.min) - 1000, 2 ** 62))) if int(info.max) - int(info.min) > 2000 else int(info.min)
    span = int(min(int(info.max) - lo, 10 ** int(rng.integers(1, 7))))
    return (lo + rng.integers(0, span + 1, size=n)).astype(dt)


while time.time() < t_end:
    which = rng.integers(0, 5)
    if which == 0:
        bt = list(NP)[rng.integers(0, len(NP))]
        n = int(rng.integers(1, 40000))
        x = rand_values(bt, n)
        nk = int([0, 1, 7, 8, 9, 200, 256, 257, 4096, 4097, 7000][rng.integers(0, 11)])
        pool = x[~np.isnan(x)] if x.dtype.kind == "f" else x
        members = rng.choice(pool, size=nk) if nk and pool.size else np.zeros(0, dtype=x.dtype)
        if rng.integers(0, 2) and nk:
            members = np.concatenate([members, rand_values(bt, 5)]).astype(x.dtype)
        col = ctx.upload(x, 4096)
        for mode in (F.In, F.NotIn):
            want = oracle_match(And(Leaf(0, mode, members)), [x], n)
            bits = ctx.bits(n)
            res = ctx.scan(And(Leaf(0, mode, members)), [col], aggs=[(R.Count, 0)], bits_out=bits)
            assert (bits.bytes(poison=32) == want).all() and res.count == O.popcount(want, n), ("inset", bt, n, nk, mode, seed)
            bits.free()
        col.free()
        n_cases["inset"] += 1
    elif which == 1:
        bt = [B.Int64, B.Uint64, B.Int32, B.Uint16][rng.integers(0, 4)]
        dt = NP[bt]
        info = np.iinfo(dt)
        width_bits = 8 * np.dtype(dt).itemsize
        bits_w = int(rng.integers(0, min(width_bits, 40) + 1))
        pack = 4096 * int(rng.integers(1, 4))
        n_packs = int(rng.integers(1, 4))
        last = int(rng.integers(1, pack + 1))
        hi = int(info.max) - ((1 << bits_w) - 1)
        base = int(info.min) + (int(rng.integers(0, min(hi - int(info.min), 2 ** 62) + 1)) if hi > int(info.min) else 0)
        packs = []
        for p in range(n_packs):
            m = pack if p < n_packs - 1 else last
            v = (np.full(m, base, dtype=object) + rng.integers(0, 1 << bits_w, size=m, dtype=np.uint64).astype(object)) if bits_w else np.full(m, base, dtype=object)
            packs.append(np.array([int(z) for z in v], dtype=dt))
        conts = [O.bitpack_container_store(OT[bt], p) for p in packs]
        whole = np.concatenate(packs)
        n = whole.size
        pk = ctx.packed_upload(bt, conts, pack)
        for _ in range(3):
            mode = [F.Equal, F.NotEqual, F.Lt, F.Le, F.Gt, F.Ge, F.Range][rng.integers(0, 7)]
            a = int(whole[rng.integers(0, n)]) + int(rng.integers(-2, 3))
            b = int(whole[rng.integers(0, n)]) + int(rng.integers(-2, 3))
            a = min(max(a, int(info.min)), int(info.max)); b = min(max(b, int(info.min)), int(info.max))
            want, wcnt = O.cmp(OT[bt], int(mode), whole, a, b if mode == F.Range else 0)
            for form in ("", "global"):
                if form: os.environ["KNOX_PACKED_FORM"] = form
                else: os.environ.pop("KNOX_PACKED_FORM", None)
                bits = ctx.bits(n)
                bits.one()
                cnt, st, _ = pk.scan(mode, a, b if mode == F.Range else None, want_sum=True, bits_out=bits)
                assert (bits.bytes(poison=32) == want).all() and cnt == wcnt, ("packed", bt, bits_w, pack, n_packs, last, mode, a, b, form, seed)
                sel = np.unpackbits(want, bitorder="little")[:n].astype(bool)
                wsum = int(whole[sel].astype(object).sum()) % (1 << width_bits)
                got = int(K.agg_value(st, R.Sum)[0]) % (1 << width_bits)
                assert got == wsum, ("packed sum", bt, bits_w, mode, form, seed)
                bits.free()
        os.environ.pop("KNOX_PACKED_FORM", None)
        pk.free()
        n_cases["packed"] += 1
    elif which == 2:
        n = int(rng.integers(1, 20000))
        alpha = int(rng.integers(1, 5))
        keys = [bytes(rng.integers(0, alpha, size=int(rng.integers(0, 24)), dtype=np.uint8)) for _ in range(int(rng.integers(1, 600)))]
        vals = [keys[i] for i in rng.integers(0, len(keys), size=n)]
        buf = np.frombuffer(b"".join(vals) + b"\x00" * 8, dtype=np.uint8)
        lens = np.array([len(v) for v in vals], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        ptr = (offs << np.uint64(32)) | lens
        col = ctx.upload_bytes(ptr, buf, 1024)
        st = [keys[i] for i in rng.integers(0, len(keys), size=int(rng.integers(0, 300)))] + [b"zz", b""][: int(rng.integers(0, 3))]
        for mode in (F.In, F.NotIn):
            want = oracle_match(And(Leaf(0, mode, st)), [(ptr, buf)], n)
            bits = ctx.bits(n)
            ctx.scan(And(Leaf(0, mode, st)), [col], bits_out=bits)
            assert (bits.bytes(poison=32) == want).all(), ("bytes_in", n, len(st), mode, seed)
            bits.free()
        col.free()
        n_cases["bytes_in"] += 1
    elif which == 4:
        # out-of-order time: the window index (a double-precision quotient estimate + multiply-back, knox_sort.hip window_index) against
        # Python's integer floor division over steps from 1 to 2^44 and clocks anywhere in int64; with / without the compaction pass
        n = int(rng.integers(100, 200000))
        step = int(2 ** rng.uniform(0, 44)) + int(rng.integers(0, 3))
        nb = int(rng.integers(1, 5000)) if rng.random() < 0.85 else int(rng.integers(15000, 90000))     # (many windows: several ranges of tables, or the sort)
        t_from = int(rng.integers(-(1 << 62), (1 << 62) - step * nb - 1))
        span = step * nb
        lo, hi = max(t_from - span // 4 - 5, -(1 << 63)), min(t_from + span + span // 4 + 5, (1 << 63) - 1)
        t = rng.integers(lo, hi, n, dtype=np.int64, endpoint=True)
        edge = rng.integers(0, nb + 1, size=min(n, 64)) * step + t_from + rng.integers(-1, 2, size=min(n, 64))   # timestamps on and beside window edges
        t[rng.integers(0, n, size=edge.size)] = np.clip(edge, lo, hi)
        bt = [B.Int64, B.Int16, B.Uint8, B.Uint32, B.Float64, B.Float32][rng.integers(0, 6)]
        v = rand_values(bt, n)
        if v.dtype.kind == "f":
            v[rng.random(n) < 0.1] = 0.0
            v[rng.random(n) < 0.1] = -0.0
            v[rng.random(n) < 0.05] = np.nan
        sel = rng.random(n) < rng.random()
        tcol, vcol = ctx.upload(t, 4096), ctx.upload(v, 4096)
        mask = ctx.bits(n).upload(np.packbits(sel, bitorder="little"))
        key = (t.astype(object) - t_from) // step
        inside = sel & np.array([0 <= k < nb for k in key])
        rows = np.flatnonzero(inside)
        ks = np.array([int(key[r]) for r in rows], dtype=np.int64)
        want_cnt = np.bincount(ks, minlength=nb)
        first = np.full(nb, -1, dtype=np.int64)
        for r, k in zip(rows[::-1], ks[::-1]):
            first[k] = r
        got = []
        for compact, form in (("0", None), ("1", None), ("1", "rows")):
            os.environ["KNOX_WINDOW_COMPACT"] = compact
            os.environ.pop("KNOX_COMPACT_FORM", None)
            if form:
                os.environ["KNOX_COMPACT_FORM"] = form
            st = ctx.reduce_buckets(tcol, vcol, t_from, step, nb, mask=mask)
            assert [s.count for s in st] == want_cnt.tolist(), ("shuffled counts", n, step, nb, t_from, compact, form, seed)
            assert [s.first_row for s in st] == first.tolist(), ("shuffled first rows", n, step, nb, t_from, compact, form, seed)
            vals, seen = ctx.reduce_buckets_values(tcol, vcol, R.Last, t_from, step, nb, mask=mask)
            last = (vals.tobytes(), seen.tobytes())      # (views of the context's result buffer: copied before the next *_view call)
            got.append((bytes(ctx.reduce_buckets_view(tcol, vcol, t_from, step, nb, mask=mask)),) + last)
        os.environ.pop("KNOX_WINDOW_COMPACT", None)
        os.environ.pop("KNOX_COMPACT_FORM", None)
        assert got[0] == got[1] == got[2], ("shuffled forms", n, step, nb, t_from, seed)
        # Value() without the sort (window_direct_kernel) against the sorted path, every reducer that has the form
        for f in (R.Sum, R.Count, R.Mean, R.Min, R.Max, R.First, R.Last) + (R.Var, R.Std):
            for m in (mask, None):
                os.environ["KNOX_WINDOW_DIRECT"] = "0"
                a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                want = (a.copy(), b.copy())
                os.environ.pop("KNOX_WINDOW_DIRECT")
                a, b = ctx.reduce_buckets_values(tcol, vcol, f, t_from, step, nb, mask=m)
                if v.dtype.kind == "f" and f in (R.Sum, R.Mean, R.Var, R.Std):      # (a NaN's payload is not part of Value()'s contract; its presence and every sign are)
                    same = np.array_equal(a, want[0], equal_nan=True) and (np.signbit(a) == np.signbit(want[0]))[~np.isnan(want[0])].all()
                else:
                    same = a.tobytes() == want[0].tobytes()
                assert same and b.tobytes() == want[1].tobytes(), ("direct values", n, step, nb, t_from, bt, f, m is not None, seed)
        for o in (tcol, vcol, mask):
            o.free()
        n_cases["shuffled"] += 1
    else:
        n = int(rng.integers(100, 300000))
        step = int(rng.integers(1, 400))
        dtm = rng.integers(0, 3, n)
        t = (1_000_000 + np.cumsum(dtm)).astype(np.int64)
        t_from = int(t[0]) - int(rng.integers(0, 5))
        nb = int((t[-1] - t_from) // step) + 1
        bt = [B.Int64, B.Float64, B.Int16, B.Uint8, B.Uint32, B.Float32][rng.integers(0, 6)]
        v = rand_values(bt, n)
        if v.dtype.kind == "f":
            v = np.where(np.isnan(v), 1.5, v).astype(v.dtype)
        tcol, vcol = ctx.upload(t, 4096), ctx.upload(v, 4096)
        mh = np.packbits(rng.random(n) < rng.random(), bitorder="little")
        mask = ctx.bits(n).upload(mh)
        got = {}
        for form in ("block", "wave", "group"):
            os.environ["KNOX_WINDOW_FORM"] = form
            got[form] = bytes(ctx.reduce_buckets_view(tcol, vcol, t_from, step, nb, mask=mask))
            vals, seen = ctx.reduce_buckets_values(tcol, vcol, R.Sum, t_from, step, nb, mask=mask)
            got[form + "v"] = (vals.tobytes(), seen.tobytes())
        os.environ.pop("KNOX_WINDOW_FORM", None)
        assert got["block"] == got["wave"] == got["group"], ("windows forms", n, step, bt, seed)
        assert got["blockv"] == got["wavev"] == got["groupv"], ("windows values", n, step, bt, seed)
        for o in (tcol, vcol, mask):
            o.free()
        n_cases["windows"] += 1
print("fuzz ok:", n_cases, "seed", seed, flush=True)
