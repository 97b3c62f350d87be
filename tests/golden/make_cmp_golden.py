#!/usr/bin/env python3
"""Extract the reference's compare-kernel golden vectors into a JSON data fixture.

Reads (as text; the reference cannot be compiled here, there is no Go toolchain)
    /root/reference/internal/cmp/tests/{int8..int64,uint8..uint64,float32,float64}.go
and evaluates the *data* they hold: the 32-value source vectors (`*_s0..s4`), the match
scalars (`*_mat_*`), the expected bitset bytes (`*_res_*`) and the case tables
(`Int64EqualCases` ...), including the tiling rule of the `mk*` helpers
(internal/cmp/tests/int64.go:165-207): repeat src / result up to the requested length,
clear the unused bits of the last byte, count = popcount(result).

Output: tests/golden/cmp_cases.json -- a list of
    {dtype, op, name, n, src, a, b, result (hex), count}
Integers are stored as JSON ints; floats as their IEEE bit patterns (ints) so that NaN,
+-Inf, -0 and denormals survive exactly.  Only data is written, no reference source text.

Run from the repo root:  python tests/golden/make_cmp_golden.py
"""
import json
import math
import os
import re
import struct
import sys

REF = "/root/reference/internal/cmp/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cmp_cases.json")

FILES = {
    "i8": ("int8.go", "int8"), "i16": ("int16.go", "int16"), "i32": ("int32.go", "int32"),
    "i64": ("int64.go", "int64"), "u8": ("uint8.go", "uint8"), "u16": ("uint16.go", "uint16"),
    "u32": ("uint32.go", "uint32"), "u64": ("uint64.go", "uint64"),
    "f32": ("float32.go", "float32"), "f64": ("float64.go", "float64"),
}
OPS = {"Equal": "eq", "NotEqual": "ne", "Less": "lt", "LessEqual": "le",
       "Greater": "gt", "GreaterEqual": "ge", "Between": "bw"}


class GoMath:
    MaxInt8, MinInt8 = 127, -128
    MaxInt16, MinInt16 = 32767, -32768
    MaxInt32, MinInt32 = 2**31 - 1, -2**31
    MaxInt64, MinInt64 = 2**63 - 1, -2**63
    MaxUint8, MaxUint16, MaxUint32, MaxUint64 = 255, 65535, 2**32 - 1, 2**64 - 1
    MaxFloat32 = float.fromhex("0x1.fffffep127")
    SmallestNonzeroFloat32 = float.fromhex("0x1p-149")
    MaxFloat64 = sys.float_info.max
    SmallestNonzeroFloat64 = float.fromhex("0x1p-1074")

    @staticmethod
    def Inf(sign):
        return math.inf if sign >= 0 else -math.inf

    @staticmethod
    def NaN():
        return math.nan


class GoDiv:
    """Wrapper so that `/` on Go untyped constants truncates for ints, divides for floats."""

    def __init__(self, v):
        self.v = v

    def __truediv__(self, o):
        o = o.v if isinstance(o, GoDiv) else o
        if isinstance(self.v, int) and isinstance(o, int):
            q = abs(self.v) // abs(o)
            return GoDiv(q if (self.v >= 0) == (o >= 0) else -q)
        return GoDiv(self.v / o)

    def __neg__(self):
        return GoDiv(-self.v)


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def unwrap(v):
    if isinstance(v, GoDiv):
        return v.v
    if isinstance(v, list):
        return [unwrap(x) for x in v]
    return v


def go_eval(expr, env, is_float):
    """Evaluate the tiny Go expression subset used by the fixtures."""
    e = expr.strip()
    # append(a, b...) -> (a + b)
    while True:
        m = re.search(r"append\(([^()]*?),\s*([^()]*?)\.\.\.\)", e)
        if not m:
            break
        e = e[:m.start()] + "(%s + %s)" % (m.group(1), m.group(2)) + e[m.end():]
    e = re.sub(r"make\(\[\]\w+,\s*0\)", "[]", e)
    e = re.sub(r"\bnil\b", "[]", e)
    # slice literals
    e = re.sub(r"\[\]\w+\{", "[", e)
    e = e.replace("}", "]")
    # float32(x) conversions: round to f32
    e = re.sub(r"\bfloat32\(", "F32(", e)
    # math constants participate in Go-style constant division
    e = re.sub(r"math\.(Max\w+|Min\w+|Smallest\w+)", r"D(math.\1)", e)
    ns = {"math": GoMath, "D": GoDiv,
          "F32": lambda x: struct.unpack("<f", struct.pack("<f", unwrap(x)))[0]}
    ns.update(env)
    return unwrap(eval(e, {"__builtins__": {}}, ns))


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def parse_var_block(text, is_float):
    m = re.search(r"\nvar \(\n(.*?)\n\)\n", text, flags=re.S)
    body = m.group(1)
    env = {}
    # statements: name [type] = expr ; expr may span lines when it is a {...} literal
    pos = 0
    stmt_re = re.compile(r"\s*(\w+)(?:\s+\w+)?\s*=\s*", re.S)
    while True:
        m = stmt_re.match(body, pos)
        if not m:
            break
        name = m.group(1)
        i = m.end()
        if body.startswith("[]", i):
            j = body.index("}", i) + 1
        else:
            j = body.find("\n", i)
            if j < 0:
                j = len(body)
        env[name] = go_eval(body[i:j], env, is_float)
        pos = j
    return env


def bitfield_len(n):
    return (n + 7) >> 3


def mk(src, result, length):
    assert len(src) % 8 == 0 and len(result) == bitfield_len(len(src))
    s = list(src)
    while length > len(s):
        s = s + s
    s = s[:length]
    r = list(result)
    while bitfield_len(length) > len(r):
        r = r + r
    r = r[:bitfield_len(length)]
    if length % 8:
        r[-1] &= 0xFF >> (8 - length % 8)
    return s, r, sum(bin(b).count("1") for b in r)


def enc(v, dtype):
    if dtype == "f64":
        return struct.unpack("<Q", struct.pack("<d", v))[0]
    if dtype == "f32":
        return struct.unpack("<I", struct.pack("<f", v))[0]
    return int(v)


def main():
    cases = []
    for dtype, (fname, gotype) in FILES.items():
        is_float = dtype[0] == "f"
        text = strip_comments(open(os.path.join(REF, fname)).read())
        env = parse_var_block(text, is_float)
        for m in re.finditer(r"\nvar (\w+?)(Equal|NotEqual|LessEqual|Less|GreaterEqual|Greater|Between)"
                             r"Cases = \[\]MatchTest\[\w+\]\{\n(.*?)\n\}\n", text, flags=re.S):
            op = OPS[m.group(2)]
            for line in m.group(3).split("\n"):
                line = line.strip().rstrip(",")
                if not line:
                    continue
                if line.startswith("{"):
                    f = split_top(line[1:-1])
                    name = json.loads(f[0])
                    src = go_eval(f[1], env, is_float)
                    a, b = go_eval(f[2], env, is_float), go_eval(f[3], env, is_float)
                    res = go_eval(f[4], env, is_float)
                    cnt = int(f[5])
                else:
                    inner = line[line.index("(") + 1:line.rindex(")")]
                    f = split_top(inner)
                    name = json.loads(f[0])
                    src0 = go_eval(f[1], env, is_float)
                    a, b = go_eval(f[2], env, is_float), go_eval(f[3], env, is_float)
                    res0 = go_eval(f[4], env, is_float)
                    src, res, cnt = mk(src0, res0, int(f[5]))
                if is_float and dtype == "f32":
                    a = struct.unpack("<f", struct.pack("<f", a))[0] if not math.isnan(a) else a
                    b = struct.unpack("<f", struct.pack("<f", b))[0] if not math.isnan(b) else b
                cases.append({
                    "dtype": dtype, "op": op, "name": name, "n": len(src),
                    "src": [enc(v, dtype) for v in src],
                    "a": enc(a, dtype), "b": enc(b, dtype),
                    "result": bytes(res).hex(), "count": cnt,
                })
    with open(OUT, "w") as fh:
        json.dump({"source": "internal/cmp/tests/*.go (data only)", "cases": cases}, fh,
                  separators=(",", ":"))
    by = {}
    for c in cases:
        by[c["dtype"]] = by.get(c["dtype"], 0) + 1
    print("wrote %d cases -> %s  %s" % (len(cases), OUT, by))


if __name__ == "__main__":
    main()
