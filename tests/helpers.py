"""Shared test helpers: the oracle-side evaluation of a filter tree (the reference's
MatchAnd / MatchOr data flow, internal/operator/filter/match_core.go:45-215) and value codecs."""
import numpy as np

from oracle import oracle as O
from knoxdb_amd import And, Leaf, Or
from knoxdb_amd.types import NP_DTYPE, BlockType, FilterMode


def oracle_leaf(leaf, host_cols):
    col = host_cols[leaf.col]
    if isinstance(col, tuple) and leaf.mode in (FilterMode.In, FilterMode.NotIn):
        return O.bytes_inset(leaf.mode == FilterMode.NotIn, col[0], col[1], list(leaf.a))
    if isinstance(col, tuple):  # bytes column: (ptr, buf)
        bits, _ = O.bytes_match(int(leaf.mode), col[0], col[1], bytes(leaf.a), val2=bytes(leaf.b) if leaf.b is not None else None)
        return bits
    dt = int({v: k for k, v in NP_DTYPE.items()}[col.dtype.type])
    if leaf.mode in (FilterMode.In, FilterMode.NotIn):
        return O.inset(dt, leaf.mode == FilterMode.NotIn, col, np.asarray(leaf.a, dtype=col.dtype))
    if leaf.mode == FilterMode.True_:
        return O.bits_one(col.size)
    if leaf.mode == FilterMode.False_:
        return np.zeros(O.bitfield_len(col.size), np.uint8)
    bits, _ = O.cmp(dt, int(leaf.mode), col, leaf.a, leaf.b)
    return bits


def oracle_match(node, host_cols, n):
    """filter.Match on one materialized 'pack' of n rows -> bitset bytes."""
    if isinstance(node, Leaf):
        return oracle_leaf(node, host_cols)
    if isinstance(node, And):
        bits = O.bits_one(n)  # MatchAnd starts with bits.One()
        for ch in node.children:
            scratch = oracle_match(ch, host_cols, n)
            bits, any_, _ = O.bits_flag_op("and", bits, scratch, n)  # bits.AndFlag(scratch)
            if not any_:
                break  # early exit changes work, never the result
        return bits
    bits = np.zeros(O.bitfield_len(n), np.uint8)  # MatchOr starts from zero
    for ch in node.children:
        bits = O.bits_op("or", bits, oracle_match(ch, host_cols, n), n)
    return bits


def special_floats(dtype, n, rng):
    """random floats salted with NaN / +-Inf / +-0 / denormals / extremes"""
    info = np.finfo(dtype)
    x = (rng.standard_normal(n) * 100).astype(dtype)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, info.tiny, -info.tiny, info.max, info.min,
                         info.smallest_subnormal, -info.smallest_subnormal, 5.0, -5.12], dtype=dtype)
    k = max(1, n // 7)
    if n:
        pos = rng.integers(0, n, size=k)
        x[pos] = specials[rng.integers(0, specials.size, size=k)]
    return x


def rand_col(bt, n, rng):
    bt = BlockType(bt)
    npdt = NP_DTYPE[bt]
    if bt in (BlockType.Float64, BlockType.Float32):
        return special_floats(npdt, n, rng)
    info = np.iinfo(npdt)
    x = rng.integers(-50 if info.min < 0 else 0, 50, size=n).astype(npdt)
    k = max(1, n // 9)
    if n:
        pos = rng.integers(0, n, size=k)
        ext = np.array([info.min, info.max, info.min + 1, info.max - 1, 0, 5], dtype=npdt)
        x[pos] = ext[rng.integers(0, ext.size, size=k)]
    return x


# ---- the reference's StringPool matcher vectors (tests/golden/bytes_cases.json <- pkg/stringx/match_test.go) ----
def bytes_golden_cases():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bytes_cases.json")))["cases"]


def bytes_golden_pool(case, first=0, last=None, dedup=False):
    """rows [first, last) of a golden case as one StringPool (ptr, buf): every row the 8-byte big-endian string
    Uint64Bytes(v) (match_test.go:137-141), appended back to back as StringPool.Append does (pool.go:205-225);
    dedup=True lets equal rows share the first occurrence's bytes (a layout Range / sub-pools produce)"""
    import numpy as np
    rows = case["rows_u64_be"][first:last]
    buf, ptr, seen = bytearray(), [], {}
    for v in rows:
        s = int(v).to_bytes(8, "big")
        if dedup and s in seen:
            ptr.append(seen[s])
            continue
        p = (len(buf) << 32) | len(s)
        seen[s] = p
        ptr.append(p)
        buf += s
    return np.array(ptr, dtype=np.uint64), np.frombuffer(bytes(buf) + b"\x00" * 3, dtype=np.uint8)
