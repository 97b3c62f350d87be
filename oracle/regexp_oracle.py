"""CPU oracle for the bytes REGEXP matcher (internal/operator/filter/match_bytes.go:543-609).

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

The reference calls Go's standard-library regexp (RE2 syntax) -- a dependency that is not part of the
reference tree (go.mod `go 1.24`; package regexp, regexp/syntax, unicode/utf8).  Its published semantics
are restated here on top of an INDEPENDENT engine, CPython's `re`:
  * rows are decoded the way regexp's inputBytes.step does: utf8.DecodeRune, an invalid or truncated
    encoding is ONE rune U+FFFD of width 1 (`go_decode` below; Python's own errors="replace" differs),
  * the pattern is translated token by token from Go syntax to Python syntax (`translate`): `$` without
    (?m) and `\\z` become `\\Z` (Python's `$` also matches before a final newline, Go's does not),
    `\\x{10FFFF}` and `(?<name>` are spelled the Python way, POSIX classes are expanded, and re.ASCII makes
    \\d \\s \\w \\b ASCII-only as in Go,
  * what cannot be expressed (\\Q..\\E, (?U), flag groups in mid-pattern, [[:^class:]]) raises
    NotImplementedError -- the tests pin those by hand-worked cases instead.
Known gaps of this checker, avoided by the generators: Python's \\s includes \\v (Go's does not); under
re.ASCII|re.I `k`/`s` do not match U+212A / U+017F (Go's simple folding does).

Patterns with \\p{..} classes or non-ASCII case folding go through `regexp_match_unicode`: the same translation onto
the third-party `regex` module (Unicode 17.0 tables; CPython's `re` has neither \\p nor tables newer than 13.0), with
\\d \\s \\w \\b spelled out as ASCII sets because that module has no way to keep them ASCII while folding case by
Unicode rules.  Two places where that module differs from Go and that the generators avoid / the model below handles:
it folds i / I with the Turkic U+0130 / U+0131 (Go's SimpleFold does not), and under IGNORECASE it reads \\p{Lu} as
"any cased letter" (Go: the table plus the runes whose fold orbit enters it, `fold_class_model`).

Pinned by: the reference's only regexp vector (operator/filter/match_test.go:244-283: "Hel.*" matches
"Hello", not "World") and the hand-worked cases in tests/test_regexp.py taken from the RE2 syntax page.
"""
import re

POSIX = {
    "alnum": "0-9A-Za-z", "alpha": "A-Za-z", "ascii": "\\x00-\\x7f", "blank": "\\t ", "cntrl": "\\x00-\\x1f\\x7f",
    "digit": "0-9", "graph": "!-~", "lower": "a-z", "print": " -~", "punct": "!-/:-@\\[-`{-~", "space": "\\t-\\r ",
    "upper": "A-Z", "word": "0-9A-Za-z_", "xdigit": "0-9A-Fa-f",
}


def go_decode(b):
    """bytes -> str of the runes Go's regexp sees (unicode/utf8.DecodeRune)."""
    out, i, n = [], 0, len(b)
    while i < n:
        b0 = b[i]
        if b0 < 0x80:
            out.append(chr(b0)); i += 1; continue
        if b0 < 0xC2 or b0 > 0xF4:
            out.append("\ufffd"); i += 1; continue
        lo, hi = 0x80, 0xBF
        if b0 < 0xE0:
            size = 2
        elif b0 < 0xF0:
            size = 3
            if b0 == 0xE0: lo = 0xA0
            if b0 == 0xED: hi = 0x9F
        else:
            size = 4
            if b0 == 0xF0: lo = 0x90
            if b0 == 0xF4: hi = 0x8F
        ok = i + size <= n and lo <= b[i + 1] <= hi and all(0x80 <= b[i + k] <= 0xBF for k in range(2, size))
        if not ok:
            out.append("\ufffd"); i += 1; continue
        out.append(b[i:i + size].decode("utf-8")); i += size
    return "".join(out)


ASCII_PERL = {"d": "0-9", "w": "0-9A-Za-z_", "s": "\\t\\n\\f\\r "}


def translate(pat, uni=False):
    """Go regexp syntax -> (Python pattern, flags); uni: for the `regex` module with Unicode folding (see the module text)."""
    flags = 0 if uni else re.ASCII
    multi = False
    m = re.match(r"\(\?([imsU]+)\)", pat)
    if m:
        if "U" in m.group(1):
            raise NotImplementedError("(?U)")
        for ch in m.group(1):
            flags |= {"i": re.I, "m": re.M, "s": re.S}[ch]
        multi = "m" in m.group(1)
        pat = pat[m.end():]
    out, i, n = [], 0, len(pat)

    def hex_escape(j):   # after "\x": returns (char, next index)
        if pat[j] == "{":
            k = pat.index("}", j)
            return chr(int(pat[j + 1:k], 16)), k + 1
        return chr(int(pat[j:j + 2], 16)), j + 2

    while i < n:
        c = pat[i]
        if c == "\\":
            d = pat[i + 1]
            if d == "z":
                out.append("\\Z"); i += 2
            elif d == "x":
                ch, i = hex_escape(i + 2)
                out.append(re.escape(ch))
            elif d == "b":   # spelled out: CPython's \b / \B never match in an empty text, Go's \B does
                W = "(?-i:[0-9A-Za-z_])" if uni else "\\w"   # word boundaries are ASCII and never folded (syntax.IsWordChar)
                out.append("(?:(?<=%s)(?!%s)|(?<!%s)(?=%s))" % (W, W, W, W)); i += 2
            elif d == "B":
                W = "(?-i:[0-9A-Za-z_])" if uni else "\\w"   # word boundaries are ASCII and never folded (syntax.IsWordChar)
                out.append("(?:(?<=%s)(?=%s)|(?<!%s)(?!%s))" % (W, W, W, W)); i += 2
            elif d in "pP" and uni:
                k = pat.index("}", i) + 1 if pat[i + 2] == "{" else i + 3
                out.append(pat[i:k]); i = k
            elif uni and d in "dwsDWS":
                out.append("[%s%s]" % ("^" if d.isupper() else "", ASCII_PERL[d.lower()])); i += 2
            elif d in "QEpPC":
                raise NotImplementedError("\\" + d)
            else:
                out.append(pat[i:i + 2]); i += 2
        elif c == "[":
            j = i + 1
            body = ["["]
            if pat[j] == "^":
                body.append("^"); j += 1
            first = True
            while True:
                if pat[j] == "]" and not first:
                    break
                first = False
                if pat.startswith("[:", j):
                    k = pat.index(":]", j)
                    name = pat[j + 2:k]
                    if name.startswith("^"):
                        raise NotImplementedError("[[:^class:]]")
                    body.append(POSIX[name]); j = k + 2
                elif pat[j] == "\\" and pat[j + 1] == "x":
                    ch, j = hex_escape(j + 2)
                    body.append(re.escape(ch))
                elif pat[j] == "\\" and pat[j + 1] in "pP" and uni:
                    k = pat.index("}", j) + 1 if pat[j + 2] == "{" else j + 3
                    body.append(pat[j:k]); j = k
                elif pat[j] == "\\" and uni and pat[j + 1] in "dws":
                    body.append(ASCII_PERL[pat[j + 1]]); j += 2
                elif pat[j] == "\\" and uni and pat[j + 1] in "DWS":
                    raise NotImplementedError("negated Perl class inside a class")
                elif pat[j] == "\\":
                    body.append(pat[j:j + 2]); j += 2
                elif pat[j] == "[":
                    body.append("\\["); j += 1
                else:
                    body.append(pat[j]); j += 1
            body.append("]")
            out.append("".join(body)); i = j + 1
        elif c == "$":
            out.append("$" if multi else "\\Z"); i += 1
        elif c == "{" and not re.match(r"\{\d+(,\d*)?\}", pat[i:]):
            out.append("\\{"); i += 1   # Go: a brace that is not {n} {n,} {n,m} is a literal (CPython reads {,m} as a repeat)
        elif pat.startswith("(?<", i) and pat[i + 3] not in "=!":
            out.append("(?P<"); i += 3
        elif pat.startswith("(?", i) and pat[i + 2] in "imsU-":
            raise NotImplementedError("flag group inside the pattern")
        else:
            out.append(c); i += 1
    return "".join(out), flags


def regexp_match(pattern, rows):
    """[bool] per row: Go's regexp.MustCompile(pattern).Match(row)."""
    if isinstance(pattern, bytes):
        pattern = pattern.decode("utf-8")
    py, flags = translate(pattern)
    rx = re.compile(py, flags)
    return [rx.search(go_decode(bytes(r))) is not None for r in rows]


def regexp_match_unicode(pattern, rows):
    """[bool] per row for patterns with \\p{..} classes / non-ASCII folding, on the `regex` module (Unicode 17.0)."""
    import regex
    if isinstance(pattern, bytes):
        pattern = pattern.decode("utf-8")
    py, flags = translate(pattern, uni=True)
    rflags = 0
    for f_re, f_rx in ((re.I, regex.I), (re.M, regex.M), (re.S, regex.S)):
        if flags & f_re:
            rflags |= f_rx
    rx = regex.compile(py, rflags | regex.V0)
    return [rx.search(go_decode(bytes(r))) is not None for r in rows]


def fold_orbit(cp):
    """unicode.SimpleFold's orbit of a code point (CaseFolding.txt C + S: the `regex` module's cases without the Turkic pair)."""
    import regex
    import regex._regex as R
    if cp in (0x130, 0x131):
        return {cp}
    return set(R.get_all_cases(regex.I | regex.U, cp)) - {0x130, 0x131}


def fold_class_model(name, cp, negated=False):
    """Is rune cp in (?i)\\p{name} (negated: \\P{name}) the way regexp/syntax builds it: the table together with every rune
    whose fold orbit enters the table, negated LAST."""
    import regex
    rx = regex.compile(r"\p{%s}" % name)
    inside = any(rx.match(chr(o)) is not None for o in fold_orbit(cp))
    return inside != negated
