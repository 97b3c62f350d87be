#!/usr/bin/env python3
"""Extract the reference's bitset / hash known-answer data into JSON fixtures.

Sources (read as text, data only):
  internal/bitset/tests/pop.go:18-47      PopCases   (source bytes, size, expected count)
  internal/bitset/tests/run.go:32-520     RunTestcases (bitset bytes, size, expected Idx rows,
                                          expected forward runs)
  internal/bitset/tests/bench.go:8-15     Patterns   (fill bytes used by the property tests)
  internal/bitset/generic/bitset_test.go:24-28  bitsetSizes
  internal/hash/xxh3_test.go:14-31        xxh3 u64/u32 known answers

Output: tests/golden/bitset_cases.json, tests/golden/xxh3_kat.json
Run from the repo root:  python tests/golden/make_bitset_golden.py
"""
import json
import os
import re

REF = "/root/reference/internal"
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def bitfield_len(n):
    return (n + 7) >> 3


def bytemask(size):
    return 0xFF >> (7 - ((size - 1) & 7))


def fill_bitset(_buf, size, val):
    # internal/bitset/tests/gen.go:28-38 (data-generation rule restated)
    buf = [val & 0xFF] * bitfield_len(size)
    if buf:
        buf[-1] &= bytemask(size)
    return buf


def fill_index(start, length):
    return list(range(start, start + length))


class _Bytes:
    @staticmethod
    def Repeat(b, n):
        return list(b) * n


def go_eval(expr):
    e = expr.strip()
    while True:
        m = re.search(r"append\(((?:[^()]|\([^()]*\))*?),\s*((?:[^()]|\([^()]*\))*?)\.\.\.\)", e)
        if not m:
            break
        e = e[:m.start()] + "(%s + %s)" % (m.group(1), m.group(2)) + e[m.end():]
    # append(slice, byte(x)) -> slice + [x]
    while True:
        m = re.search(r"append\(((?:[^()]|\((?:[^()]|\([^()]*\))*\))*),\s*byte\((\w+)\)\)", e)
        if not m:
            break
        e = e[:m.start()] + "(%s + [%s])" % (m.group(1), m.group(2)) + e[m.end():]
    e = re.sub(r"\[\]\[2\]int\{", "[", e)
    e = re.sub(r"\[\]\w+\{", "[", e)
    e = e.replace("{", "[").replace("}", "]")
    e = re.sub(r"\bnil\b", "[]", e)
    ns = {"fillIndex": fill_index, "FillBitset": fill_bitset, "bytes": _Bytes}
    return eval(e, {"__builtins__": {}}, ns)


def split_fields(body):
    """Split 'Name: x,\n Buf: y,\n ...' at top-level commas into a dict."""
    out, depth, cur = [], 0, ""
    for ch in body:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    d = {}
    for f in out:
        f = f.strip()
        if not f:
            continue
        k, v = f.split(":", 1)
        d[k.strip()] = v.strip()
    return d


def top_level_structs(block):
    """Yield the text inside each top-level {...} of a Go slice literal body."""
    depth, start = 0, None
    for i, ch in enumerate(block):
        if ch == "{":
            if depth == 0:
                start = i + 1
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                yield block[start:i]


def main():
    pop_txt = strip_comments(open(f"{REF}/bitset/tests/pop.go").read())
    body = re.search(r"var PopCases = \[\]PopTest\{(.*)\n\}", pop_txt, flags=re.S).group(1)
    pops = []
    for s in top_level_structs(body):
        d = split_fields(s)
        pops.append({"name": json.loads(d["Name"]), "source": bytes(go_eval(d["Source"])).hex(),
                     "result": bytes(go_eval(d["Result"])).hex(),
                     "size": go_eval(d["Size"]), "count": go_eval(d["Count"])})

    run_txt = strip_comments(open(f"{REF}/bitset/tests/run.go").read())
    body = re.search(r"var RunTestcases = \[\]BitsetRunTestcase\{(.*?)\n\}\n", run_txt, flags=re.S).group(1)
    runs = []
    for s in top_level_structs(body):
        d = split_fields(s)
        buf = go_eval(d["Buf"])
        runs.append({"name": json.loads(d["Name"]), "buf": bytes(buf).hex(), "size": go_eval(d["Size"]),
                     "idx": go_eval(d["Idx"]), "runs": go_eval(d["Runs"])})

    b_txt = strip_comments(open(f"{REF}/bitset/tests/bench.go").read())
    patterns = go_eval(re.search(r"var Patterns = (\[\]byte\{.*?\})", b_txt, flags=re.S).group(1))
    g_txt = strip_comments(open(f"{REF}/bitset/generic/bitset_test.go").read())
    sizes = go_eval(re.search(r"var bitsetSizes = (\[\]int\{.*?\})", g_txt, flags=re.S).group(1))

    with open(os.path.join(HERE, "bitset_cases.json"), "w") as fh:
        json.dump({"source": "internal/bitset/tests/{pop,run,bench}.go, generic/bitset_test.go (data only)",
                   "pop": pops, "run": runs, "patterns": patterns, "sizes": sizes}, fh, separators=(",", ":"))
    print("bitset: %d pop cases, %d run cases, patterns=%s sizes=%s" % (len(pops), len(runs), patterns, sizes))

    # xxh3 known answers: input = little-endian ints from bytes {i..i+7}
    h_txt = strip_comments(open(f"{REF}/hash/xxh3_test.go").read())
    m = re.search(r"xxh3Uint32Result\s*=\s*\[\]uint64\{(.*?)\}", h_txt, flags=re.S)
    u32 = [int(x) for x in re.findall(r"\d+", m.group(1))]
    m = re.search(r"xxh3Uint64Result\s*=\s*\[\]uint64\{(.*?)\}", h_txt, flags=re.S)
    u64 = [int(x) for x in re.findall(r"\d+", m.group(1))]
    with open(os.path.join(HERE, "xxh3_kat.json"), "w") as fh:
        json.dump({"source": "internal/hash/xxh3_test.go:14-31 (data only)",
                   "input_rule": "case i hashes the little-endian integer made of bytes i, i+1, ...",
                   "u32": u32, "u64": u64}, fh)
    print("xxh3: %d u32, %d u64 known answers" % (len(u32), len(u64)))


if __name__ == "__main__":
    main()
