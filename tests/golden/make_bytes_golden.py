#!/usr/bin/env python3
"""Extract the reference's StringPool matcher golden vectors into a JSON data fixture.

Reads (as text) /root/reference/pkg/stringx/match_test.go and evaluates the data it holds:

  * the three 32-value source vectors `u64_s0..s2`, the match values `u64_<op>_mat_<k>[a|b]` and the expected
    bitset bytes `u64_<op>_res_<k>` (match_test.go:370-470);
  * the seven case tables `Bytes{Equal,NotEqual,Less,LessEqual,Greater,GreaterEqual,Between}Cases`
    (match_test.go:232-368) with the construction rule of `mkBytes` (match_test.go:174-227): every source value
    becomes the 8-byte BIG-endian string `Uint64Bytes(v)` (:137-141), the pool is the source repeated to the requested
    length (`Range(0, k).AppendTo(pool)` doubles a periodic prefix, so row i = src[i % len(src)]), the result bytes are
    repeated to BitFieldLen(length), the unused bits of the last byte cleared, count = popcount(result);
  * the literal first case of every table: `{"l0", 0, NewStringPool(0), []byte{5}, ...}` -- an empty pool.

What the driver `testCompareCases` (match_test.go:48-78) asserts per case is what the fixture's consumers assert:
`set.Count() == count`, the result bytes, and 32 poison bytes (0xfa) behind them untouched.

Only data is written: per case the row values (u64, to be written big-endian), the match value(s) as hex strings, the
expected bytes as hex, the count.  Run from the repo root:  python tests/golden/make_bytes_golden.py
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_cmp_golden import GoMath, split_top, strip_comments  # noqa: E402

REF = "/root/reference/pkg/stringx/match_test.go"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bytes_cases.json")
TABLES = {"Equal": "eq", "NotEqual": "ne", "Less": "lt", "LessEqual": "le", "Greater": "gt", "GreaterEqual": "ge", "Between": "bw"}


def go_eval(expr, env):
    """the small expression language of the data block: integer / byte lists, math.MaxUintN, append(a, b...)"""
    e = expr.strip()
    m = re.fullmatch(r"append\((.+?),\s*(.+?)\.\.\.\)", e, flags=re.S)
    if m:
        return list(go_eval(m.group(1), env)) + list(go_eval(m.group(2), env))
    e = re.sub(r"\[\]\w+\{", "[", e).replace("{", "[").replace("}", "]")
    ns = {"math": GoMath}
    ns.update(env)
    return eval(e, {"__builtins__": {}}, ns)


def parse_vars(text):
    """`name [type] = value` entries of the var ( ... ) block that holds u64_s0 (the first one is BitFieldLen)"""
    env = {}
    for blk in re.finditer(r"\nvar \(\n(.*?)\n\)\n", text, flags=re.S):
        if "u64_s0" not in blk.group(1):
            continue
        body = blk.group(1)
        # an entry runs from `name ... =` to the next entry start; lists span lines
        ents = list(re.finditer(r"^\s*(u64_\w+)\s*(?:uint64\s*)?=\s*", body, flags=re.M))
        for i, m in enumerate(ents):
            end = ents[i + 1].start() if i + 1 < len(ents) else len(body)
            env[m.group(1)] = go_eval(body[m.end():end].strip().rstrip(","), env)
    assert len(env["u64_s0"]) == len(env["u64_s1"]) == len(env["u64_s2"]) == 32
    return env


def bitfield_len(n):
    return (n + 7) >> 3


def mk_bytes(name, src, match, match2, result, length):
    assert len(src) % 8 == 0 and len(result) == bitfield_len(len(src)), name       # the two panics of mkBytes
    rows = [src[i % len(src)] for i in range(length)]
    n = bitfield_len(length)
    res = list(result)
    while n > len(res):
        res = res + res
    res = res[:n]
    if length % 8:
        res[-1] &= 0xFF >> (8 - length % 8)
    be = lambda v: int(v).to_bytes(8, "big").hex()
    return {"name": name, "n": length, "rows_u64_be": [int(v) for v in rows], "match": be(match), "match2": be(match2),
            "result": bytes(res).hex(), "count": sum(bin(b).count("1") for b in res)}


def main():
    text = strip_comments(open(REF).read())
    env = parse_vars(text)
    cases = []
    for table, op in TABLES.items():
        m = re.search(r"var Bytes%sCases = \[\]BytesMatchTest\{\n(.*?)\n\}\n" % table, text, flags=re.S)
        assert m, table
        for line in m.group(1).split("\n"):
            line = line.strip().rstrip(",")
            if not line:
                continue
            if line.startswith("{"):
                # {"l0", 0, NewStringPool(0), []byte{5}, []byte{} | []byte{10}, []byte{}, 0}
                f = split_top(line[1:-1])
                assert f[0] == '"l0"' and f[1].strip() == "0" and f[2].strip() == "NewStringPool(0)" and f[6].strip() == "0", line
                lit = lambda s: bytes(go_eval(s, {})).hex()
                cases.append({"op": op, "name": "l0", "n": 0, "rows_u64_be": [], "match": lit(f[3]), "match2": lit(f[4]),
                              "result": lit(f[5]), "count": 0})
                continue
            mm = re.fullmatch(r"mkBytes\((.*)\)", line, flags=re.S)
            assert mm, line
            f = split_top(mm.group(1))
            assert len(f) == 6, line
            c = mk_bytes(json.loads(f[0]), go_eval(f[1], env), go_eval(f[2], env), go_eval(f[3], env), go_eval(f[4], env), int(f[5]))
            c["op"] = op
            cases.append(c)
    per = {op: sum(c["op"] == op for c in cases) for op in TABLES.values()}
    assert all(v == 15 for v in per.values()), per
    json.dump({"source": "pkg/stringx/match_test.go:25-33,174-227,232-470", "poison": 32, "cases": cases}, open(OUT, "w"), separators=(",", ":"))
    print("wrote %d cases -> %s" % (len(cases), OUT))


if __name__ == "__main__":
    main()
