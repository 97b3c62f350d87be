#!/usr/bin/env python3
"""Extract the reference's Int128 / Int256 compare golden vectors into a JSON data fixture.

Reads (as text) /root/reference/internal/cmp/tests/int128.go and int256.go and evaluates the data
they hold: the 32-value source vectors (`i128_s1..s3`), the match values (`*_mat_*`), the
expected bitset bytes (`*_res_*`) and the case tables (`Int128EqualCases` ...) with the tiling
rule of `mkI128` / `mkI256` (int128.go:165-208): repeat src / result to the requested length,
clear the unused bits of the last byte, count = popcount(result).

Values are written as limb lists, MOST significant limb first (num.Int128 = [2]uint64{hi, lo},
num.Int256 = [4]uint64, pkg/num/int128.go:22-42, int256.go:21-55), every limb as an unsigned
64-bit JSON int.  Only data is written.

Run from the repo root:  python tests/golden/make_cmp_wide_golden.py
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_cmp_golden import GoMath, mk, split_top, strip_comments  # noqa: E402

REF = "/root/reference/internal/cmp/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cmp_wide_cases.json")
OPS = {"Equal": "eq", "NotEqual": "ne", "Less": "lt", "LessEqual": "le", "Greater": "gt", "GreaterEqual": "ge", "Between": "bw"}
M64 = (1 << 64) - 1


def u64(x):
    return int(x) & M64


def sign_of(x):
    return M64 if int(x) < 0 else 0


# constructors of pkg/num (int128.go:33-42,127-129; int256.go:33-55,150-158), on limb lists
def i128_from_i64(x):
    return [sign_of(x), u64(x)]


def i128_from_2(a, b):
    return [u64(a), u64(b)]


def i256_from_i128(v):
    s = M64 if v[0] >> 63 else 0
    return [s, s, v[0], v[1]]


def i256_from_i64(x):
    s = sign_of(x)
    return [s, s, s, u64(x)]


NS = {
    "math": GoMath,
    "Int128FromInt64": i128_from_i64, "Int128From2Int64": i128_from_2,
    "MaxInt128": [0x7FFFFFFFFFFFFFFF, M64], "MinInt128": [1 << 63, 0], "ZeroInt128": [0, 0],
    "Int256FromInt64": i256_from_i64, "Int256From2Int64": lambda a, b: i256_from_i128(i128_from_2(a, b)),
    "Int256FromInt128": i256_from_i128,
    "MaxInt256": [0x7FFFFFFFFFFFFFFF, M64, M64, M64], "MinInt256": [1 << 63, 0, 0, 0], "ZeroInt256": [0, 0, 0, 0],
}


def go_eval(expr, env):
    e = expr.strip()
    e = re.sub(r"make\(\[\]\w+,\s*0\)", "[]", e)
    e = re.sub(r"\bnil\b", "[]", e)
    e = re.sub(r"\[\]\w+\{", "[", e)
    e = e.replace("{", "[").replace("}", "]")
    ns = dict(NS)
    ns.update(env)
    return eval(e, {"__builtins__": {}}, ns)


def parse_data_block(text):
    """the `var ( ... )` block that holds the vectors (the first one only aliases constructors)"""
    for m in re.finditer(r"\nvar \(\n(.*?)\n\)\n", text, flags=re.S):
        if re.search(r"_s1\s*=", m.group(1)):
            body = m.group(1)
            break
    else:
        raise SystemExit("data block not found")
    env, pos = {}, 0
    stmt_re = re.compile(r"\s*(\w+)\s*=\s*", re.S)
    while True:
        m = stmt_re.match(body, pos)
        if not m:
            break
        name, i = m.group(1), m.end()
        if body.startswith("[]", i):
            # slice literal, possibly with nested {..} elements: find the matching brace
            depth, j = 0, body.index("{", i)
            while True:
                depth += body[j] == "{"
                depth -= body[j] == "}"
                j += 1
                if depth == 0:
                    break
        else:
            j = body.find("\n", i)
            j = len(body) if j < 0 else j
        env[name] = go_eval(body[i:j], env)
        pos = j
    return env


def main():
    cases = []
    for dtype, fname, limbs in (("i128", "int128.go", 2), ("i256", "int256.go", 4)):
        text = strip_comments(open(os.path.join(REF, fname)).read())
        env = parse_data_block(text)
        for m in re.finditer(r"\nvar Int\d+(Equal|NotEqual|LessEqual|Less|GreaterEqual|Greater|Between)"
                             r"Cases = \[\]Int\d+MatchTest\{\n(.*?)\n\}\n", text, flags=re.S):
            op = OPS[m.group(1)]
            for line in m.group(2).split("\n"):
                line = line.strip().rstrip(",")
                if not line:
                    continue
                if line.startswith("{"):
                    f = split_top(line[1:-1])
                    name, src = json.loads(f[0]), go_eval(f[1], env)
                    a, b, res, cnt = go_eval(f[2], env), go_eval(f[3], env), go_eval(f[4], env), int(f[5])
                else:
                    f = split_top(line[line.index("(") + 1:line.rindex(")")])
                    name = json.loads(f[0])
                    a, b = go_eval(f[2], env), go_eval(f[3], env)
                    src, res, cnt = mk(go_eval(f[1], env), go_eval(f[4], env), int(f[5]))
                assert all(len(v) == limbs for v in src) and len(a) == limbs and len(b) == limbs
                cases.append({"dtype": dtype, "op": op, "name": name, "n": len(src), "src": [[u64(x) for x in v] for v in src],
                              "a": [u64(x) for x in a], "b": [u64(x) for x in b], "result": bytes(res).hex(), "count": cnt})
    with open(OUT, "w") as fh:
        json.dump({"source": "internal/cmp/tests/int128.go, int256.go (data only)", "limb_order": "most significant first",
                   "cases": cases}, fh, separators=(",", ":"))
    by = {}
    for c in cases:
        by[c["dtype"]] = by.get(c["dtype"], 0) + 1
    print("wrote %d cases -> %s  %s" % (len(cases), OUT, by))


if __name__ == "__main__":
    main()
