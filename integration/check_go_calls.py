#!/usr/bin/env python3
"""Type-check the cgo call sites of integration/go/**/*.go against include/knoxgpu.h.

There is no Go toolchain in the build image, so the Go side of the boundary cannot be compiled here.  This script is the
next best thing and runs in the CPU test suite (tests/test_go_boundary.py):

  * parses every function declaration of include/knoxgpu.h (name, return type, parameter types);
  * finds every `C.knox_*(...)` call in the Go sources, splits its arguments, infers the cgo type of each argument
    from the conversions / declarations in the enclosing function (`C.int(x)`, `(*C.uint8_t)(p)`, `unsafe.Pointer(p)`,
    `var out C.knox_scan_out` ... `&out`, struct fields `c.p`, `nil`) and compares arity and types with the header;
  * checks that every call that reaches a context (first parameter knox_ctx*, or the free of a column / bitset / filter)
    in the cgo package sits behind that context's mutex: the library's contexts are not re-entrant;
  * checks the struct fields the Go code touches (`cn[i].val_len`, `out.rows_scanned`, `ca[i]._func` ...) exist in the
    header's structs, and that the enum constants restated in Go have the header's values.

An argument whose type cannot be inferred is an ERROR (the Go source is kept in forms this script can read), so a
clean run means every C call was checked.  Exit code 0 = clean.  `--list` prints the checked calls."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "knoxgpu.h")
GO_DIR = os.path.join(ROOT, "integration", "go")


# ---------------------------------------------------------------------------------------------------------------------
# the header
# ---------------------------------------------------------------------------------------------------------------------
def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def norm_c(t):
    """'const knox_col *const *' -> 'knox_col**';  'uint8_t id[128]' -> 'uint8_t*'."""
    t = t.strip()
    arr = "[" in t
    t = re.sub(r"\[[^\]]*\]", "", t)
    t = re.sub(r"\bconst\b", " ", t)
    t = re.sub(r"\b(struct|enum)\b", " ", t)
    stars = t.count("*") + (1 if arr else 0)
    words = t.replace("*", " ").split()
    # drop the parameter name: the last word if more than one word and it is not a type keyword
    base = words
    if len(words) > 1 and words[-1] not in ("int", "char", "long", "double", "float", "void", "unsigned"):
        base = words[:-1]
    return " ".join(base) + "*" * stars


def split_top(s, sep=","):
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        out.append("".join(cur))
    return [x.strip() for x in out]


def parse_header(path):
    src = strip_comments(open(path).read())
    funcs, structs, enums = {}, {}, {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = {}
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mt = re.match(r"(.*?)([\w\s,\*\[\]]+)$", decl, flags=re.S)
            # 'int32_t write_buffers, chunks, probes' / 'const uint8_t *val' / 'uint64_t a, b'
            head = decl.split(",")[0]
            base = norm_c(head)
            names = [re.sub(r"[\*\s]|\[[^\]]*\]", "", head.split()[-1])] + [re.sub(r"[\*\s]|\[[^\]]*\]", "", x) for x in decl.split(",")[1:]]
            for nm in names:
                fields[nm] = base
        structs[m.group(3)] = fields
    for m in re.finditer(r"enum\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
        val = -1
        for item in split_top(m.group(2)):
            if not item:
                continue
            if "=" in item:
                k, v = item.split("=")
                val = int(v.strip(), 0)
            else:
                k, val = item, val + 1
            enums[k.strip()] = val
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    body = re.sub(r"enum\s+\w+\s*\{.*?\}\s*;", " ", body, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(knox_\w+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        ps = [] if params.strip() in ("", "void") else [norm_c(p) for p in split_top(params)]
        funcs[name] = (norm_c(ret + " x"), ps)
    return funcs, structs, enums


# ---------------------------------------------------------------------------------------------------------------------
# the Go side
# ---------------------------------------------------------------------------------------------------------------------
GO_SCALARS = {"int", "size_t", "int64_t", "uint64_t", "int32_t", "uint32_t", "uint8_t", "double", "float"}


def go_ctype(t):
    """'**C.knox_col' -> 'knox_col**';  'C.int' -> 'int'."""
    t = t.strip()
    stars = len(t) - len(t.lstrip("*"))
    t = t.lstrip("*")
    if t == "unsafe.Pointer":
        return "void*"
    if not t.startswith("C."):
        return None
    return t[2:] + "*" * stars


def func_bodies(src):
    """Yields (header_line, body_text) of every top-level Go func."""
    for m in re.finditer(r"^func\s[^\n{]*\{", src, flags=re.M):
        start = m.end()
        depth, i = 1, start
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        yield m.group(0), src[start:i]


class Scope:
    def __init__(self, structs_go, header_line, body):
        self.vars = {}
        self.structs_go = structs_go  # Go struct name -> {field: cgo type}
        self.objs = {}  # Go variable -> Go struct name (receivers / params of wrapper types)
        for nm, ty in re.findall(r"(\w+)\s+\*?(Ctx|Column|Bits|Bloom|State|Packed|Encoded)\b", header_line):
            self.objs[nm] = ty
        for names, ty in re.findall(r"\b([\w, ]+?)\s+\*?(Ctx|Column|Bits|Bloom|State|Packed|Encoded)\b", header_line):
            for nm in names.split(","):
                self.objs[nm.strip()] = ty
        # parameters of cgo types: `rc C.int`
        for nm, ty in re.findall(r"(\w+)\s+(\*{0,2}C\.\w+)", header_line):
            self.vars[nm] = go_ctype(ty)
        # unsafe.Pointer parameters and locals are C's void*
        for nm in re.findall(r"(\w+)\s+unsafe\.Pointer\b", header_line) + re.findall(r"\bvar\s+(\w+)\s+unsafe\.Pointer\b", body):
            self.vars[nm] = "void*"
        # the owning context of a wrapper object: `c := col.c`
        for nm in re.findall(r"\b(\w+)\s*:=\s*\w+\.c\b(?!\.)", body):
            self.objs[nm] = "Ctx"
        # locals of wrapper types: `var out State`
        for nm, ty in re.findall(r"\bvar\s+(\w+)\s+\*?(Ctx|Column|Bits|Bloom|State|Packed|Encoded)\b", body):
            self.objs[nm] = ty
        # var a, b C.type / var a *C.type
        for names, ty in re.findall(r"\bvar\s+([\w, ]+?)\s+(\*{0,2}C\.\w+)", body):
            for nm in names.split(","):
                self.vars[nm.strip()] = go_ctype(ty)
        # a, b := C.t1(x), C.t2(y)   /   a := C.t(x)
        for lhs, rhs in re.findall(r"^\s*([\w, ]+?)\s*:=\s*(C\.\w+\(.*)$", body, flags=re.M):
            names = [x.strip() for x in lhs.split(",")]
            convs = split_top(rhs)
            if len(names) == len(convs):
                for nm, cv in zip(names, convs):
                    m = re.match(r"C\.(\w+)\(", cv)
                    if m and m.group(1) in GO_SCALARS:
                        self.vars[nm] = m.group(1)
        # x := make([]T, n) with T a cgo type
        for nm, ty in re.findall(r"(\w+)\s*:=\s*make\(\[\]\s*(\*{0,2}C\.\w+)", body):
            self.vars[nm] = "[]" + go_ctype(ty)
        # x := C.CBytes(v)
        for nm in re.findall(r"(\w+)\s*:=\s*C\.CBytes\(", body):
            self.vars[nm] = "void*"
        # for _, p := range slice
        for nm, sl in re.findall(r"for\s+\w+,\s*(\w+)\s*:=\s*range\s+(\w+)", body):
            if self.vars.get(sl, "").startswith("[]"):
                self.vars[nm] = self.vars[sl][2:]

    def typeof(self, e):
        e = e.strip()
        if e == "nil":
            return "nil"
        m = re.match(r"^\((\*{1,2}C\.\w+)\)\(", e)  # (*C.uint8_t)(...)
        if m:
            return go_ctype(m.group(1))
        m = re.match(r"^C\.(\w+)\(", e)  # C.int(...)
        if m:
            return m.group(1)
        if e.startswith("unsafe.Pointer("):
            return "void*"
        if e.startswith("&"):
            inner = self.typeof(e[1:])
            return None if inner in (None, "nil") else inner + "*"
        m = re.match(r"^(\w+)\[[^\]]+\]$", e)  # cols[i]
        if m and self.vars.get(m.group(1), "").startswith("[]"):
            return self.vars[m.group(1)][2:]
        m = re.match(r"^(\w+)\.(\w+)$", e)  # c.p / s.c / col.p
        if m and m.group(1) in self.objs:
            return self.structs_go.get(self.objs[m.group(1)], {}).get(m.group(2))
        if re.match(r"^\w+$", e):
            return self.vars.get(e)
        return None


def go_structs(src):
    out = {}
    for m in re.finditer(r"type\s+(\w+)\s+struct\s*\{(.*?)\}", src, flags=re.S):
        fields = {}
        for line in re.split(r"[\n;]", m.group(2)):
            mm = re.match(r"\s*(\w+)\s+(\*{0,2}C\.\w+)", line)
            if mm:
                fields[mm.group(1)] = go_ctype(mm.group(2))
        out[m.group(1)] = fields
    return out


def compatible(go_t, c_t):
    if go_t == "nil":
        return c_t.endswith("*")
    if go_t == c_t:
        return True
    if c_t == "void*" and go_t == "void*":
        return True
    return False


def find_calls(body):
    for m in re.finditer(r"\bC\.(knox_\w+)\(", body):
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(body[i], 0)
            i += 1
        yield m.group(1), split_top(body[m.end():i - 1]), body.count("\n", 0, m.start())


LOCKED_FREES = {"knox_col_free", "knox_bits_free", "knox_bloom_free", "knox_packed_free", "knox_enc_free"}

GO_ENUMS = {  # Go constant name -> header enum name (restated values must agree)
    "blockInt64": "KNOX_INT64", "blockInt32": "KNOX_INT32", "blockInt16": "KNOX_INT16", "blockInt8": "KNOX_INT8",
    "blockUint64": "KNOX_UINT64", "blockUint32": "KNOX_UINT32", "blockUint16": "KNOX_UINT16", "blockUint8": "KNOX_UINT8",
    "blockFloat64": "KNOX_FLOAT64", "blockFloat32": "KNOX_FLOAT32",
    "modeEq": "KNOX_EQ", "modeNe": "KNOX_NE", "modeGt": "KNOX_GT", "modeGe": "KNOX_GE", "modeLt": "KNOX_LT",
    "modeLe": "KNOX_LE", "modeRange": "KNOX_RANGE",
    "NodeLeaf": "KNOX_NODE_LEAF", "NodeAnd": "KNOX_NODE_AND", "NodeOr": "KNOX_NODE_OR",
    "RSum": "KNOX_R_SUM", "RFirst": "KNOX_R_FIRST", "RLast": "KNOX_R_LAST", "RMin": "KNOX_R_MIN", "RMax": "KNOX_R_MAX",
    "RMean": "KNOX_R_MEAN", "RVar": "KNOX_R_VAR", "RStd": "KNOX_R_STD", "RCount": "KNOX_R_COUNT",
}


def go_consts(src):
    """`a, b, c = 1, 2, 3` inside const blocks and `const x = 5`."""
    out = {}
    for lhs, rhs in re.findall(r"^\s*(?:const\s+)?([A-Za-z_][\w, ]*?)\s*=\s*([-\d, ]+)\s*$", src, flags=re.M):
        names, vals = [x.strip() for x in lhs.split(",")], [x.strip() for x in rhs.split(",")]
        if len(names) == len(vals) and all(re.match(r"^-?\d+$", v) for v in vals):
            out.update({n: int(v) for n, v in zip(names, vals)})
    return out


def structure_errors(text):
    """What a missing toolchain still allows: brackets of a Go file balance outside strings, runes and comments, and no
    statement opener (`if` / `for` / `switch` / `func` ... `{`) lost its brace to a trailing comment."""
    errs, stack, i, n, line = [], [], 0, len(text), 1
    pairs = {")": "(", "]": "[", "}": "{"}
    while i < n:
        c = text[i]
        if c == "\n":
            line += 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
            continue
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            line += text.count("\n", i, j if j >= 0 else n)
            i = n if j < 0 else j + 2
            continue
        elif c == "`":
            j = text.find("`", i + 1)
            line += text.count("\n", i, j if j >= 0 else n)
            i = n if j < 0 else j + 1
            continue
        elif c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            i = j + 1
            continue
        elif c in "([{":
            stack.append((c, line))
        elif c in ")]}":
            if not stack or stack[-1][0] != pairs[c]:
                errs.append("line %d: unbalanced `%s`" % (line, c))
                return errs
            stack.pop()
        i += 1
    if stack:
        errs.append("line %d: `%s` is never closed" % (stack[-1][1], stack[-1][0]))
    for ln, l in enumerate(text.split("\n"), 1):
        code = l.split("//")[0].rstrip()
        if re.match(r"^\s*(if|for|switch|func|} else if|} else)\b", code) and "//" in l and not code.endswith(("{", ",", "(", "||", "&&")) and l.split("//", 1)[1].rstrip().endswith("{"):
            errs.append("line %d: the opening brace of this statement sits inside a comment" % ln)
    return errs


def check(verbose=False):
    funcs, structs, enums = parse_header(HEADER)
    errors, checked, bound = [], 0, set()
    go_files = []
    for d, _, fs in os.walk(GO_DIR):
        go_files += [os.path.join(d, f) for f in fs if f.endswith(".go")]
    for path in sorted(go_files):
        for e in structure_errors(open(path).read()):
            errors.append(f"{os.path.relpath(path, ROOT)}: {e}")
        src = re.sub(r"//[^\n]*", "", open(path).read().replace("/*\n#cgo", "/*cgo"))
        src = re.sub(r"/\*cgo.*?\*/", "", src, flags=re.S)
        rel = os.path.relpath(path, ROOT)
        sg = go_structs(src)
        # struct fields of C structs touched from Go: x.field where x's element type is a header struct
        for header_line, body in func_bodies(src):
            sc = Scope(sg, header_line, body)
            for name, args, line in find_calls(body):
                bound.add(name)
                if name not in funcs:
                    errors.append(f"{rel}: {name}: not declared in include/knoxgpu.h")
                    continue
                ret, params = funcs[name]
                if len(args) != len(params):
                    errors.append(f"{rel}: {name}: {len(args)} arguments, header takes {len(params)}")
                    continue
                for i, (a, p) in enumerate(zip(args, params)):
                    t = sc.typeof(a)
                    if t is None:
                        errors.append(f"{rel}: {name}: argument {i + 1} `{a}`: type not inferable (header: {p})")
                    elif not compatible(t, p):
                        errors.append(f"{rel}: {name}: argument {i + 1} `{a}` is {t}, header takes {p}")
                checked += 1
                # the context is not re-entrant (include/knoxgpu.h "Conventions"; one goroutine per query in the reference):
                # every call that reaches a knox_ctx -- through its first parameter, or by freeing an object whose free
                # synchronises the context's stream -- must sit behind the context's mutex
                needs_lock = (params[:1] == ["knox_ctx*"] and name not in ("knox_ctx_open", "knox_ctx_open_stream")) or name in LOCKED_FREES
                if needs_lock and os.path.basename(path) == "knoxgpu.go":
                    head = "\n".join(body.split("\n")[:line + 1])
                    if not re.search(r"\bmu\.Lock\(\)", head):
                        errors.append(f"{rel}: {name}: called without the context's lock (c.mu.Lock() must precede it in the same function)")
                if verbose:
                    print(f"ok  {rel}: {name}({', '.join(params)}) -> {ret}")
            # field accesses on C structs: name[i].field / name.field for variables of header struct types
            for var, ty in sc.vars.items():
                base = (ty or "").lstrip("[]").rstrip("*")
                if base in structs:
                    for fld in set(re.findall(r"\b%s(?:\[[^\]]+\])?\.(\w+)\b" % re.escape(var), body)):
                        f = fld[1:] if fld.startswith("_") else fld  # cgo spells C fields that are Go keywords `_func`
                        if f not in structs[base]:
                            errors.append(f"{rel}: {var}.{fld}: struct {base} has no such field")
        # the context pool's rule (knoxgpu.Pool): a context taken with Get() goes back with a deferred Put() of the SAME
        # variable in the same function, right after the Get -- otherwise an early return (every fallback path has one)
        # leaks the context and the next Get blocks for good
        for header_line, body in func_bodies(src):
            for m in re.finditer(r"\b(\w+)\s*:=\s*(\w+)\.Get\(\)", body):
                var, pool = m.group(1), m.group(2)
                rest = body[m.end():]
                nxt = rest.lstrip().split("\n", 1)[0].strip()
                if nxt != f"defer {pool}.Put({var})":
                    ln = body[:m.start()].count("\n") + 1
                    errors.append(f"{rel}: `{var} := {pool}.Get()` (function line {ln}) is not followed by `defer {pool}.Put({var})`")
        for gname, val in go_consts(src).items():
            if gname in GO_ENUMS and enums.get(GO_ENUMS[gname]) != val:
                errors.append(f"{rel}: const {gname} = {val}, header says {GO_ENUMS[gname]} = {enums.get(GO_ENUMS[gname])}")
    return errors, checked, sorted(bound), sorted(set(funcs) - bound)


def main():
    errors, checked, bound, unbound = check(verbose="--list" in sys.argv)
    for e in errors:
        print("ERROR", e)
    print(f"{checked} C call sites checked, {len(bound)} of {len(bound) + len(unbound)} entry points bound from Go, {len(errors)} errors")
    if "--unbound" in sys.argv:
        print("not bound from Go:", ", ".join(unbound))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
