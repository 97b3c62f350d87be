// Host-side compiler for the bytes REGEXP matcher (internal/operator/filter/match_bytes.go:543-609:
// bytesRegexpMatcher = Go's regexp.Regexp.Match on every row).
//
// Go's regexp package (RE2 syntax, https://pkg.go.dev/regexp/syntax) is standard library, not part of
// the reference tree, so this restates its published semantics for Match (is there ANY match):
//   * the input is decoded rune by rune exactly as utf8.DecodeRune does; an invalid byte is one rune
//     U+FFFD of width 1 (regexp/regexp.go inputBytes.step),
//   * ^ $ are text anchors unless (?m); \b \B are ASCII word boundaries; \d \s \w and the POSIX
//     classes are ASCII; `.` excludes \n unless (?s); (?i) closes every literal and class under the
//     orbits of Unicode simple case folding (unicode.SimpleFold: K k U+212A, S s U+017F, the Greek
//     theta's four forms ...), as regexp/syntax's appendFoldedRange does,
//   * \pN \p{Name} \P{Name} \p{^Name}, alone or inside [...]: the general categories (one- and
//     two-letter names, LC, Cn, and the long aliases such as Letter / Uppercase_Letter / punct), the
//     scripts by their long names, and Any / Assigned / ASCII; names are matched ignoring case,
//     spaces, underscores and hyphens (regexp/syntax unicodeTable since Go 1.25; earlier releases
//     accept the exact names only and reject the rest at Compile, before this code is reached);
//     under (?i) the class is the table together with its fold-equivalents, THEN negated,
//   * greediness / (?U) / captures do not change whether a match exists and are parsed and ignored.
// The Unicode data is knox_unicode_tables.h: Unicode 17.0.0 (knox_unicode::kVersion, exported as
// knox_regexp_unicode_version so that a host whose unicode.Version differs keeps \p{..} and folded
// non-ASCII patterns on its own regexp).  Anything this parser does not know (a construct Go rejects)
// is refused with `unsupported`, never approximated.
//
// Output: a DFA over RUNE CLASSES (the partition of [0, 0x10FFFF] induced by every range endpoint of
// the pattern).  DFA state = (set of NFA states, kind of the previous rune); the empty-width
// assertions are resolved when the closure is taken, with the previous and the next rune's kind in
// hand, so ^ $ \b \B \A \z and (?m) all cost nothing at match time.  State 0 is the absorbing
// ACCEPT state (unanchored search: a leading "any rune" loop is built in by re-adding the start
// state after every step).  The device walks table[state][class] once per decoded rune.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "knox_unicode_tables.h"

namespace knox_regex {

constexpr uint32_t kMaxRune = 0x10FFFF;
constexpr int kMaxNfa = 20000;      // NFA nodes (repeat expansion bound)
constexpr int kMaxStates = 4096;    // DFA states
constexpr int kMaxClasses = 255;    // rune classes
constexpr uint16_t kDeadBit = 0x8000;  // table entry: the target state is dead (no accepting state reachable)

struct Range { uint32_t lo, hi; };
using Set = std::vector<Range>;

inline void normalize(Set &s) {
    std::sort(s.begin(), s.end(), [](const Range &a, const Range &b) { return a.lo < b.lo || (a.lo == b.lo && a.hi < b.hi); });
    Set out;
    for (const Range &r : s) {
        if (!out.empty() && r.lo <= out.back().hi + 1 && out.back().hi != 0xffffffffu) out.back().hi = std::max(out.back().hi, r.hi);
        else out.push_back(r);
    }
    s.swap(out);
}
inline Set negate(Set s) {
    normalize(s);
    Set out;
    uint32_t next = 0;
    for (const Range &r : s) {
        if (r.lo > next) out.push_back({next, r.lo - 1});
        next = r.hi + 1;
    }
    if (next <= kMaxRune) out.push_back({next, kMaxRune});
    return out;
}
inline bool contains(const Set &s, uint32_t r) {
    for (const Range &x : s)
        if (r >= x.lo && r <= x.hi) return true;
    return false;
}

// Unicode simple case folding: the set closed under unicode.SimpleFold orbits (regexp/syntax appendFoldedRange)
inline void fold_set(Set &s) {
    using knox_unicode::kFolds;
    using knox_unicode::kNumFolds;
    Set add;
    for (const Range &r : s) {
        // the folding entries whose rune lies in [r.lo, r.hi]
        int lo = 0, hi = kNumFolds;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (kFolds[mid].rune < r.lo) lo = mid + 1; else hi = mid;
        }
        for (int i = lo; i < kNumFolds && kFolds[i].rune <= r.hi; i++) {
            uint32_t c = kFolds[i].next;
            for (int guard = 0; c != kFolds[i].rune && guard < 8; guard++) {
                add.push_back({c, c});
                int a = 0, b = kNumFolds;  // next of c
                while (a < b) {
                    const int mid = (a + b) >> 1;
                    if (kFolds[mid].rune < c) a = mid + 1; else b = mid;
                }
                if (a >= kNumFolds || kFolds[a].rune != c) break;
                c = kFolds[a].next;
            }
        }
    }
    s.insert(s.end(), add.begin(), add.end());
    normalize(s);
}

// \p{name}: canonical form of a property name (upper case; spaces, underscores and hyphens dropped)
inline std::string canonical_name(const uint8_t *p, size_t n) {
    std::string out;
    for (size_t i = 0; i < n; i++) {
        const uint8_t c = p[i];
        if (c == '_' || c == ' ' || c == '-') continue;
        out.push_back((char)(c >= 'a' && c <= 'z' ? c - 32 : c));
    }
    return out;
}

// the table a \p name stands for (regexp/syntax unicodeTable): false = unknown name.  negate: the name means the complement.
inline bool unicode_table(const std::string &name, Set &out, bool &neg) {
    using namespace knox_unicode;
    neg = false;
    out.clear();
    if (name == "ANY") { out = {{0, kMaxRune}}; return true; }
    if (name == "ASCII") { out = {{0, 0x7f}}; return true; }
    std::string cat = name;
    if (name == "ASSIGNED") { cat = "CN"; neg = true; }
    static const struct { const char *alias, *cat; } kAliases[] = {
        {"OTHER", "C"}, {"CONTROL", "CC"}, {"CNTRL", "CC"}, {"FORMAT", "CF"}, {"UNASSIGNED", "CN"}, {"PRIVATEUSE", "CO"}, {"SURROGATE", "CS"},
        {"LETTER", "L"}, {"CASEDLETTER", "LC"}, {"LOWERCASELETTER", "LL"}, {"MODIFIERLETTER", "LM"}, {"OTHERLETTER", "LO"},
        {"TITLECASELETTER", "LT"}, {"UPPERCASELETTER", "LU"}, {"MARK", "M"}, {"COMBININGMARK", "M"}, {"SPACINGMARK", "MC"},
        {"ENCLOSINGMARK", "ME"}, {"NONSPACINGMARK", "MN"}, {"NUMBER", "N"}, {"DECIMALNUMBER", "ND"}, {"DIGIT", "ND"},
        {"LETTERNUMBER", "NL"}, {"OTHERNUMBER", "NO"}, {"PUNCTUATION", "P"}, {"PUNCT", "P"}, {"CONNECTORPUNCTUATION", "PC"},
        {"DASHPUNCTUATION", "PD"}, {"CLOSEPUNCTUATION", "PE"}, {"FINALPUNCTUATION", "PF"}, {"INITIALPUNCTUATION", "PI"},
        {"OTHERPUNCTUATION", "PO"}, {"OPENPUNCTUATION", "PS"}, {"SYMBOL", "S"}, {"CURRENCYSYMBOL", "SC"}, {"MODIFIERSYMBOL", "SK"},
        {"MATHSYMBOL", "SM"}, {"OTHERSYMBOL", "SO"}, {"SEPARATOR", "Z"}, {"LINESEPARATOR", "ZL"}, {"PARAGRAPHSEPARATOR", "ZP"},
        {"SPACESEPARATOR", "ZS"}};
    // categories first (one letter = every category of that group, C with Cn as in unicode.C since Go 1.25; LC = Lu | Ll | Lt)
    auto cat_mask = [&](const std::string &c, uint64_t &mask) {
        mask = 0;
        for (int i = 0; i < kNumCategories; i++) {
            const char a = kCategoryNames[i][0], b = (char)(kCategoryNames[i][1] - 32);
            const bool hit = c.size() == 1 ? c[0] == a : c == "LC" ? (a == 'L' && (b == 'U' || b == 'L' || b == 'T')) : (c.size() == 2 && c[0] == a && c[1] == b);
            if (hit) mask |= (uint64_t)1 << i;
        }
        return mask != 0;
    };
    uint64_t mask = 0;
    bool is_cat = cat.size() <= 2 && cat_mask(cat, mask);
    if (!is_cat) {
        for (int si = 0; si < kNumScripts; si++)
            if (name == kScriptNames[si]) {
                for (int i = 0; i < kNumScriptRuns; i++)
                    if (kScriptRuns[i].id == si) out.push_back({kScriptRuns[i].lo, kScriptRuns[i].hi});
                normalize(out);
                return true;
            }
        for (const auto &al : kAliases)
            if (name == al.alias) { is_cat = cat_mask(al.cat, mask); break; }
    }
    if (!is_cat) return false;
    for (int i = 0; i < kNumCategoryRuns; i++)
        if ((mask >> kCategoryRuns[i].id) & 1) out.push_back({kCategoryRuns[i].lo, kCategoryRuns[i].hi});
    normalize(out);
    return true;
}

enum AssertKind { A_BOT, A_EOT, A_BOL, A_EOL, A_WORDB, A_NWORDB };
enum Op { OP_EMPTY, OP_SET, OP_CAT, OP_ALT, OP_REPEAT, OP_ASSERT };

struct Ast {
    int op = OP_EMPTY;
    Set set;
    std::vector<int> sub;
    int min = 0, max = 0;  // repeat, max -1 = unbounded
    int kind = 0;          // assert
};

struct Flags { bool fold = false, multi = false, dotnl = false; };

enum Status { OK = 0, UNSUPPORTED = 1, TOO_LARGE = 2 };

struct Parser {
    const uint8_t *p, *e;
    std::vector<Ast> ast;
    int status = OK;
    int depth = 0;

    int fail() { status = UNSUPPORTED; return -1; }
    int node(int op) { ast.push_back(Ast()); ast.back().op = op; return (int)ast.size() - 1; }
    bool more() const { return p < e; }

    // one rune of the pattern text (the pattern must be valid UTF-8, as regexp.Compile demands)
    bool rune(uint32_t &r) {
        if (p >= e) return false;
        const uint8_t b0 = *p;
        if (b0 < 0x80) { r = b0; p++; return true; }
        int n = b0 >= 0xF0 ? 4 : b0 >= 0xE0 ? 3 : b0 >= 0xC2 ? 2 : 0;
        if (n == 0 || b0 > 0xF4 || p + n > e) return false;
        uint32_t v = b0 & (0xff >> (n + 1));
        for (int i = 1; i < n; i++) {
            if ((p[i] & 0xC0) != 0x80) return false;
            v = (v << 6) | (p[i] & 0x3f);
        }
        if ((n == 2 && v < 0x80) || (n == 3 && v < 0x800) || (n == 4 && v < 0x10000) || v > kMaxRune || (v >= 0xD800 && v <= 0xDFFF)) return false;
        r = v;
        p += n;
        return true;
    }

    static void perl_class(char c, Set &s) {
        switch (c) {
        case 'd': s = {{'0', '9'}}; break;
        case 's': s = {{'\t', '\n'}, {'\f', '\r'}, {' ', ' '}}; break;
        default: s = {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}}; break;  // w
        }
    }
    static bool posix_class(const std::string &name, Set &s) {
        if (name == "alnum") s = {{'0', '9'}, {'A', 'Z'}, {'a', 'z'}};
        else if (name == "alpha") s = {{'A', 'Z'}, {'a', 'z'}};
        else if (name == "ascii") s = {{0, 0x7f}};
        else if (name == "blank") s = {{'\t', '\t'}, {' ', ' '}};
        else if (name == "cntrl") s = {{0, 0x1f}, {0x7f, 0x7f}};
        else if (name == "digit") s = {{'0', '9'}};
        else if (name == "graph") s = {{'!', '~'}};
        else if (name == "lower") s = {{'a', 'z'}};
        else if (name == "print") s = {{' ', '~'}};
        else if (name == "punct") s = {{'!', '/'}, {':', '@'}, {'[', '`'}, {'{', '~'}};
        else if (name == "space") s = {{'\t', '\r'}, {' ', ' '}};
        else if (name == "upper") s = {{'A', 'Z'}};
        else if (name == "word") s = {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}};
        else if (name == "xdigit") s = {{'0', '9'}, {'A', 'F'}, {'a', 'f'}};
        else return false;
        return true;
    }

    // after a backslash: a single-rune escape (shared by atoms and classes).  false = not one of these
    bool escape_rune(uint32_t &r) {
        if (p >= e) return false;
        const uint8_t c = *p;
        switch (c) {
        case 'a': r = 7; p++; return true;
        case 'f': r = '\f'; p++; return true;
        case 'n': r = '\n'; p++; return true;
        case 'r': r = '\r'; p++; return true;
        case 't': r = '\t'; p++; return true;
        case 'v': r = '\v'; p++; return true;
        case 'x': {
            p++;
            auto hex = [](uint8_t ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; };
            if (p < e && *p == '{') {
                p++;
                uint32_t v = 0;
                int nd = 0;
                while (p < e && *p != '}') {
                    const int h = hex(*p);
                    if (h < 0) return false;
                    v = v * 16 + (uint32_t)h;
                    if (v > kMaxRune) return false;
                    nd++;
                    p++;
                }
                if (p >= e || nd == 0) return false;
                p++;
                r = v;
                return true;
            }
            if (p + 2 > e) return false;
            const int h1 = hex(p[0]), h2 = hex(p[1]);
            if (h1 < 0 || h2 < 0) return false;
            r = (uint32_t)(h1 * 16 + h2);
            p += 2;
            return true;
        }
        case '1': case '2': case '3': case '4': case '5': case '6': case '7':
            // a single non-zero digit is a backreference (not in RE2); two or more octal digits are octal
            if (p + 1 >= e || p[1] < '0' || p[1] > '7') return false;
            [[fallthrough]];
        case '0': {
            uint32_t v = (uint32_t)(c - '0');
            p++;
            for (int i = 0; i < 2 && p < e && *p >= '0' && *p <= '7'; i++) v = v * 8 + (uint32_t)(*p++ - '0');
            r = v;
            return true;
        }
        default:
            if (c < 0x80 && !((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) {  // punctuation: itself
                r = c;
                p++;
                return true;
            }
            return false;
        }
    }

    // after "\\" at 'p' or 'P': \pN, \p{Name}, \P{..}, \p{^Name} (regexp/syntax parseUnicodeClass).  The class is the
    // table, with its fold-equivalents under (?i), negated last.
    bool unicode_class(const Flags &f, Set &out) {
        bool neg = *p == 'P';
        p++;
        if (p >= e) return false;
        const uint8_t *st, *en;
        if (*p != '{') {  // single-letter name (one rune)
            st = p;
            uint32_t r;
            if (!rune(r)) return false;
            en = p;
        } else {
            st = ++p;
            while (p < e && *p != '}') p++;
            if (p >= e) return false;
            en = p++;
        }
        if (st < en && *st == '^') { neg = !neg; st++; }
        bool tneg = false;
        if (!unicode_table(canonical_name(st, (size_t)(en - st)), out, tneg)) return false;
        if (f.fold) fold_set(out);
        if (neg != tneg) out = negate(out);
        return true;
    }

    int make_set(Set s, const Flags &f) {
        if (f.fold) fold_set(s);
        normalize(s);
        const int n = node(OP_SET);
        ast[n].set = s;
        return n;
    }

    int parse_class(const Flags &f) {  // after '['
        bool neg = false;
        if (p < e && *p == '^') { neg = true; p++; }
        Set s;
        bool first = true;
        for (;;) {
            if (p >= e) return fail();
            if (*p == ']' && !first) { p++; break; }
            first = false;
            if (*p == '[' && p + 1 < e && p[1] == ':') {
                const uint8_t *q = p + 2;
                bool cneg = false;
                if (q < e && *q == '^') { cneg = true; q++; }
                const uint8_t *st = q;
                while (q < e && *q != ':') q++;
                if (q + 1 >= e || q[1] != ']') return fail();
                Set c;
                if (!posix_class(std::string((const char *)st, (size_t)(q - st)), c)) return fail();
                if (f.fold) fold_set(c);
                if (cneg) c = negate(c);
                s.insert(s.end(), c.begin(), c.end());
                p = q + 2;
                continue;
            }
            uint32_t lo;
            if (*p == '\\') {
                p++;
                if (p >= e) return fail();
                if (*p == 'd' || *p == 's' || *p == 'w' || *p == 'D' || *p == 'S' || *p == 'W') {
                    Set c;
                    perl_class((char)(*p | 0x20), c);
                    if (f.fold) fold_set(c);
                    if (*p < 'a') c = negate(c);
                    s.insert(s.end(), c.begin(), c.end());
                    p++;
                    continue;
                }
                if (*p == 'p' || *p == 'P') {
                    Set c;
                    if (!unicode_class(f, c)) return fail();
                    s.insert(s.end(), c.begin(), c.end());
                    continue;
                }
                if (!escape_rune(lo)) return fail();
            } else if (!rune(lo)) return fail();
            uint32_t hi = lo;
            if (p + 1 < e && *p == '-' && p[1] != ']') {
                p++;
                if (*p == '\\') {
                    p++;
                    if (!escape_rune(hi)) return fail();
                } else if (!rune(hi)) return fail();
                if (hi < lo) return fail();
            }
            Set c = {{lo, hi}};
            if (f.fold) fold_set(c);
            s.insert(s.end(), c.begin(), c.end());
        }
        normalize(s);
        if (neg) s = negate(s);
        const int n = node(OP_SET);
        ast[n].set = s;
        return n;
    }

    // "(?flags)" / "(?flags:" / "(?P<name>" / "(?<name>" after "(?".  returns 0 = flags only (no group), 1 = group opened
    int parse_group_prefix(Flags &f, bool &is_group) {
        is_group = false;
        if (p < e && (*p == 'P' || *p == '<')) {
            if (*p == 'P') p++;
            if (p >= e || *p != '<') return fail();
            p++;
            const uint8_t *st = p;
            while (p < e && *p != '>') {
                const uint8_t c = *p;
                if (!(c == '_' || (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) return fail();
                p++;
            }
            if (p >= e || p == st) return fail();
            p++;
            is_group = true;
            return 0;
        }
        bool neg = false, any = false;
        Flags nf = f;
        for (;;) {
            if (p >= e) return fail();
            const uint8_t c = *p++;
            switch (c) {
            case 'i': nf.fold = !neg; any = true; break;
            case 'm': nf.multi = !neg; any = true; break;
            case 's': nf.dotnl = !neg; any = true; break;
            case 'U': any = true; break;  // swaps greediness: irrelevant to Match
            case '-':
                if (neg) return fail();
                neg = true;
                any = false;
                break;
            case ':': case ')':
                if (neg && !any) return fail();
                f = nf;
                is_group = c == ':';
                return 0;
            default: return fail();
            }
        }
    }

    int parse_atom(Flags &f) {
        if (p >= e) return fail();
        const uint8_t c = *p;
        switch (c) {
        case '(': {
            p++;
            Flags inner = f;
            if (p < e && *p == '?') {
                p++;
                bool is_group;
                if (parse_group_prefix(inner, is_group) < 0) return -1;
                if (!is_group) {  // (?flags): applies to the rest of the enclosing group
                    f = inner;
                    return node(OP_EMPTY);
                }
            }
            if (++depth > 200) return fail();
            const int n = parse_alt(inner);
            depth--;
            if (n < 0) return -1;
            if (p >= e || *p != ')') return fail();
            p++;
            return n;
        }
        case '[': p++; return parse_class(f);
        case '.': {
            p++;
            const int n = node(OP_SET);
            if (f.dotnl) ast[n].set = {{0, kMaxRune}};
            else ast[n].set = {{0, '\n' - 1}, {'\n' + 1, kMaxRune}};
            return n;
        }
        case '^': { p++; const int n = node(OP_ASSERT); ast[n].kind = f.multi ? A_BOL : A_BOT; return n; }
        case '$': { p++; const int n = node(OP_ASSERT); ast[n].kind = f.multi ? A_EOL : A_EOT; return n; }
        case '*': case '+': case '?': return fail();  // missing argument to repetition operator
        case ')': return fail();
        case '\\': {
            p++;
            if (p >= e) return fail();
            const uint8_t d = *p;
            if (d == 'A') { p++; const int n = node(OP_ASSERT); ast[n].kind = A_BOT; return n; }
            if (d == 'z') { p++; const int n = node(OP_ASSERT); ast[n].kind = A_EOT; return n; }
            if (d == 'b') { p++; const int n = node(OP_ASSERT); ast[n].kind = A_WORDB; return n; }
            if (d == 'B') { p++; const int n = node(OP_ASSERT); ast[n].kind = A_NWORDB; return n; }
            if (d == 'd' || d == 's' || d == 'w' || d == 'D' || d == 'S' || d == 'W') {
                Set s;
                perl_class((char)(d | 0x20), s);
                if (f.fold) fold_set(s);
                if (d < 'a') s = negate(s);
                p++;
                const int n = node(OP_SET);
                ast[n].set = s;
                return n;
            }
            if (d == 'Q') {  // \Q ... \E literal text
                p++;
                const int cat = node(OP_CAT);
                while (p < e && !(p + 1 < e && p[0] == '\\' && p[1] == 'E')) {
                    uint32_t r;
                    if (!rune(r)) return fail();
                    const int s = make_set({{r, r}}, f);
                    if (s < 0) return -1;
                    ast[cat].sub.push_back(s);
                }
                if (p < e) p += 2;
                // a repetition after \Q..\E binds to the last rune only: keep it simple and refuse
                if (p < e && (*p == '*' || *p == '+' || *p == '?' || *p == '{')) return fail();
                return cat;
            }
            if (d == 'p' || d == 'P') {
                Set s;
                if (!unicode_class(f, s)) return fail();
                const int n = node(OP_SET);
                ast[n].set = s;
                return n;
            }
            uint32_t r;
            if (!escape_rune(r)) return fail();  // \C, backreferences, unknown escapes
            return make_set({{r, r}}, f);
        }
        default: {
            uint32_t r;
            if (!rune(r)) return fail();
            return make_set({{r, r}}, f);
        }
        }
    }

    // {n} {n,} {n,m} -> true and advances; otherwise the brace is a literal
    bool parse_counts(int &mn, int &mx) {
        const uint8_t *q = p + 1;
        auto num = [&](int &v) {
            if (q >= e || *q < '0' || *q > '9') return false;
            long x = 0;
            while (q < e && *q >= '0' && *q <= '9') {
                x = x * 10 + (*q - '0');
                if (x > 100000) x = 100000;
                q++;
            }
            v = (int)x;
            return true;
        };
        if (!num(mn)) return false;
        mx = mn;
        if (q < e && *q == ',') {
            q++;
            if (q < e && *q == '}') mx = -1;
            else if (!num(mx)) return false;
        }
        if (q >= e || *q != '}') return false;
        p = q + 1;
        return true;
    }

    int parse_repeat(Flags &f) {
        int a = parse_atom(f);
        if (a < 0) return -1;
        bool repeated = false;
        while (p < e) {
            int mn, mx;
            const uint8_t c = *p;
            if (c == '*') { mn = 0; mx = -1; p++; }
            else if (c == '+') { mn = 1; mx = -1; p++; }
            else if (c == '?') { mn = 0; mx = 1; p++; }
            else if (c == '{') {
                if (!parse_counts(mn, mx)) break;  // literal brace
                if (mn > 1000 || mx > 1000 || (mx >= 0 && mx < mn)) return fail();
            } else break;
            if (repeated) return fail();           // `a**`, `a*?+`: invalid nested repetition operator
            repeated = true;
            if (p < e && *p == '?') p++;           // non-greedy marker: same language
            const int r = node(OP_REPEAT);
            ast[r].sub.push_back(a);
            ast[r].min = mn;
            ast[r].max = mx;
            a = r;
        }
        return a;
    }

    int parse_cat(Flags &f) {
        const int cat = node(OP_CAT);
        while (p < e && *p != '|' && *p != ')') {
            const int r = parse_repeat(f);
            if (r < 0) return -1;
            ast[cat].sub.push_back(r);
        }
        return cat;
    }

    int parse_alt(Flags f) {  // flags by value: changes made by (?i) inside stay inside this group
        const int alt = node(OP_ALT);
        for (;;) {
            const int c = parse_cat(f);
            if (c < 0) return -1;
            ast[alt].sub.push_back(c);
            if (p < e && *p == '|') { p++; continue; }
            break;
        }
        return alt;
    }
};

// ---- NFA ------------------------------------------------------------------------------------------
enum NType { N_RUNE, N_SPLIT, N_ASSERT, N_MATCH };
struct NState { int type; int a = -1, b = -1; int set = -1; int kind = 0; };

struct Nfa {
    std::vector<NState> st;
    std::vector<Set> sets;
    const std::vector<Ast> *ast;
    bool too_large = false;

    int add(int type) {
        if ((int)st.size() >= kMaxNfa) { too_large = true; return 0; }
        st.push_back(NState());
        st.back().type = type;
        return (int)st.size() - 1;
    }
    int build(int n, int next) {
        if (too_large) return next;
        const Ast &x = (*ast)[n];
        switch (x.op) {
        case OP_EMPTY: return next;
        case OP_SET: {
            const int s = add(N_RUNE);
            sets.push_back(x.set);
            st[s].set = (int)sets.size() - 1;
            st[s].a = next;
            return s;
        }
        case OP_ASSERT: {
            const int s = add(N_ASSERT);
            st[s].kind = x.kind;
            st[s].a = next;
            return s;
        }
        case OP_CAT: {
            int cur = next;
            for (size_t i = x.sub.size(); i-- > 0;) cur = build(x.sub[i], cur);
            return cur;
        }
        case OP_ALT: {
            if (x.sub.size() == 1) return build(x.sub[0], next);
            int cur = build(x.sub.back(), next);
            for (size_t i = x.sub.size() - 1; i-- > 0;) {
                const int s = add(N_SPLIT);
                const int left = build(x.sub[i], next);
                st[s].a = left;
                st[s].b = cur;
                cur = s;
            }
            return cur;
        }
        default: {  // OP_REPEAT
            const int sub = x.sub[0];
            int cur;
            if (x.max < 0) {
                const int loop = add(N_SPLIT);
                const int body = build(sub, loop);
                st[loop].a = body;
                st[loop].b = next;
                cur = loop;
            } else {
                cur = next;
                for (int i = 0; i < x.max - x.min && !too_large; i++) {
                    const int s = add(N_SPLIT);
                    const int body = build(sub, cur);
                    st[s].a = body;
                    st[s].b = next;
                    cur = s;
                }
            }
            for (int i = 0; i < x.min && !too_large; i++) cur = build(sub, cur);
            return cur;
        }
        }
    }
};

// ---- the compiled program the device walks ----------------------------------------------------------
enum Kind { K_EDGE = 0 /* begin of text (previous) / end of text (next) */, K_NEWLINE = 1, K_WORD = 2, K_OTHER = 3 };

struct Program {
    uint32_t n_states = 0, n_classes = 0;
    uint8_t ascii_class[128];
    std::vector<uint32_t> bound_lo;    // runes >= 0x80: interval i starts at bound_lo[i] (sorted), ends before the next
    std::vector<uint8_t> bound_class;
    std::vector<uint16_t> table;       // [n_states][n_classes]: target state | kDeadBit; state 0 = ACCEPT, state 1 = start
    std::vector<uint8_t> flags;        // bit0: accepting at end of text, bit1: dead (no accepting state reachable)
};

inline int compile(const uint8_t *pat, size_t len, Program &out) {
    Parser ps;
    ps.p = pat;
    ps.e = pat + len;
    Flags f;
    const int root = ps.parse_alt(f);
    if (root < 0 || ps.status != OK) return UNSUPPORTED;
    if (ps.p != ps.e) return UNSUPPORTED;  // unmatched ')'
    Nfa nfa;
    nfa.ast = &ps.ast;
    const int match = nfa.add(N_MATCH);
    const int start = nfa.build(root, match);
    if (nfa.too_large) return TOO_LARGE;

    // rune classes: cut [0, kMaxRune] at every range endpoint (+ \n and the word ranges when an assertion looks at them)
    bool line_assert = false, word_assert = false;
    for (const NState &s : nfa.st)
        if (s.type == N_ASSERT) {
            if (s.kind == A_BOL || s.kind == A_EOL) line_assert = true;
            if (s.kind == A_WORDB || s.kind == A_NWORDB) word_assert = true;
        }
    std::vector<uint32_t> cuts = {0, 0x80};  // cut at 0x80 keeps the ASCII table and the interval list separate
    auto cut_set = [&](const Set &s) {
        for (const Range &r : s) {
            cuts.push_back(r.lo);
            if (r.hi < kMaxRune) cuts.push_back(r.hi + 1);
        }
    };
    for (const Set &s : nfa.sets) cut_set(s);
    if (line_assert) cut_set({{'\n', '\n'}});
    const Set word = {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}};
    if (word_assert) cut_set(word);
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    // intervals with the same membership signature share a class
    std::map<std::vector<uint8_t>, int> sig_class;
    std::vector<int> interval_class(cuts.size());
    std::vector<uint32_t> class_rep;
    for (size_t i = 0; i < cuts.size(); i++) {
        const uint32_t r = cuts[i];
        std::vector<uint8_t> sig((nfa.sets.size() + 7) / 8 + 1, 0);
        for (size_t k = 0; k < nfa.sets.size(); k++)
            if (contains(nfa.sets[k], r)) sig[k >> 3] |= (uint8_t)(1u << (k & 7));
        uint8_t kind = K_OTHER;
        if (line_assert && r == '\n') kind = K_NEWLINE;
        else if (word_assert && contains(word, r)) kind = K_WORD;
        sig.back() = kind;
        auto it = sig_class.find(sig);
        if (it == sig_class.end()) {
            if ((int)class_rep.size() >= kMaxClasses) return TOO_LARGE;
            it = sig_class.emplace(sig, (int)class_rep.size()).first;
            class_rep.push_back(r);
        }
        interval_class[i] = it->second;
    }
    const int ncls = (int)class_rep.size();
    std::vector<uint8_t> class_kind(ncls);
    std::vector<std::vector<uint8_t>> class_in(ncls, std::vector<uint8_t>(nfa.sets.size()));
    for (int c = 0; c < ncls; c++) {
        const uint32_t r = class_rep[c];
        class_kind[c] = (line_assert && r == '\n') ? K_NEWLINE : (word_assert && contains(word, r)) ? K_WORD : K_OTHER;
        for (size_t k = 0; k < nfa.sets.size(); k++) class_in[c][k] = contains(nfa.sets[k], r);
    }

    // closure with the assertion context (previous rune's kind, next rune's kind) -> RUNE states + match flag
    std::vector<int> mark(nfa.st.size(), 0);
    int stamp = 0;
    auto closure = [&](const std::vector<int> &raw, int pk, int nk, std::vector<int> &runes) {
        runes.clear();
        stamp++;
        bool matched = false;
        std::vector<int> stack(raw.begin(), raw.end());
        while (!stack.empty()) {
            const int s = stack.back();
            stack.pop_back();
            if (mark[s] == stamp) continue;
            mark[s] = stamp;
            const NState &x = nfa.st[s];
            switch (x.type) {
            case N_MATCH: matched = true; break;
            case N_RUNE: runes.push_back(s); break;
            case N_SPLIT: stack.push_back(x.a); stack.push_back(x.b); break;
            default: {
                bool ok;
                switch (x.kind) {
                case A_BOT: ok = pk == K_EDGE; break;
                case A_EOT: ok = nk == K_EDGE; break;
                case A_BOL: ok = pk == K_EDGE || pk == K_NEWLINE; break;
                case A_EOL: ok = nk == K_EDGE || nk == K_NEWLINE; break;
                case A_WORDB: ok = (pk == K_WORD) != (nk == K_WORD); break;
                default: ok = (pk == K_WORD) == (nk == K_WORD); break;
                }
                if (ok) stack.push_back(x.a);
            }
            }
        }
        return matched;
    };

    using Key = std::pair<std::vector<int>, int>;
    std::map<Key, int> ids;
    std::vector<Key> states;
    states.push_back(Key());  // 0 = ACCEPT
    auto intern = [&](std::vector<int> raw, int pk) {
        std::sort(raw.begin(), raw.end());
        raw.erase(std::unique(raw.begin(), raw.end()), raw.end());
        Key k(raw, pk);
        auto it = ids.find(k);
        if (it != ids.end()) return it->second;
        const int id = (int)states.size();
        ids.emplace(k, id);
        states.push_back(k);
        return id;
    };
    intern({start}, K_EDGE);  // 1 = start
    std::vector<uint16_t> table;
    std::vector<uint8_t> flags;
    table.assign(ncls, 0);    // ACCEPT row: absorbing
    flags.push_back(1);
    std::vector<int> runes, moved;
    for (size_t d = 1; d < states.size(); d++) {
        if ((int)states.size() > kMaxStates) return TOO_LARGE;
        const Key key = states[d];  // copy: `states` grows below
        std::vector<uint16_t> row(ncls);
        // one closure per distinct next kind
        std::vector<int> cl[4];
        bool cm[4] = {false, false, false, false}, have[4] = {false, false, false, false};
        for (int c = 0; c < ncls; c++) {
            const int nk = class_kind[c];
            if (!have[nk]) {
                cm[nk] = closure(key.first, key.second, nk, cl[nk]);
                have[nk] = true;
            }
            if (cm[nk]) { row[c] = 0; continue; }
            moved.clear();
            moved.push_back(start);  // unanchored: a match may start at the next rune
            for (int s : cl[nk])
                if (class_in[c][nfa.st[s].set]) moved.push_back(nfa.st[s].a);
            row[c] = (uint16_t)intern(moved, nk);
        }
        table.insert(table.end(), row.begin(), row.end());
        flags.push_back(closure(key.first, key.second, K_EDGE, runes) ? 1 : 0);
    }
    if ((int)states.size() > kMaxStates) return TOO_LARGE;
    const int ns = (int)states.size();
    // dead states: no accepting state reachable (reverse reachability from the accepting ones)
    std::vector<std::vector<int>> rev(ns);
    for (int d = 0; d < ns; d++)
        for (int c = 0; c < ncls; c++) rev[table[(size_t)d * ncls + c]].push_back(d);
    std::vector<uint8_t> live(ns, 0);
    std::vector<int> stack;
    for (int d = 0; d < ns; d++)
        if (flags[d] & 1) { live[d] = 1; stack.push_back(d); }
    while (!stack.empty()) {
        const int d = stack.back();
        stack.pop_back();
        for (int s : rev[d])
            if (!live[s]) { live[s] = 1; stack.push_back(s); }
    }
    for (int d = 0; d < ns; d++)
        if (!live[d]) flags[d] |= 2;
    // a transition into a dead state carries kDeadBit: the walk stops without a second lookup
    for (size_t i = 0; i < table.size(); i++)
        if (flags[table[i]] & 2) table[i] |= kDeadBit;

    out.n_states = (uint32_t)ns;
    out.n_classes = (uint32_t)ncls;
    out.table.swap(table);
    out.flags.swap(flags);
    out.bound_lo.clear();
    out.bound_class.clear();
    memset(out.ascii_class, 0, sizeof out.ascii_class);
    for (size_t i = 0; i < cuts.size(); i++) {
        const uint32_t lo = cuts[i], hi = i + 1 < cuts.size() ? cuts[i + 1] - 1 : kMaxRune;
        if (lo < 0x80) {
            for (uint32_t r = lo; r <= hi && r < 0x80; r++) out.ascii_class[r] = (uint8_t)interval_class[i];
        } else {
            if (!out.bound_class.empty() && out.bound_class.back() == (uint8_t)interval_class[i]) continue;  // merge neighbours
            out.bound_lo.push_back(lo);
            out.bound_class.push_back((uint8_t)interval_class[i]);
        }
    }
    return OK;
}

// ---- utf8.DecodeRune (unicode/utf8): invalid or short encodings are one rune U+FFFD of width 1 -------
#if defined(__HIPCC__)
#define KNOX_RX_HD __host__ __device__
#else
#define KNOX_RX_HD
#endif
KNOX_RX_HD inline uint32_t decode_rune(const uint8_t *p, uint32_t n, uint32_t &width) {
    const uint32_t b0 = p[0];
    width = 1;
    if (b0 < 0x80) return b0;
    if (b0 < 0xC2 || b0 > 0xF4) return 0xFFFD;
    uint32_t size, lo = 0x80, hi = 0xBF;
    if (b0 < 0xE0) size = 2;
    else if (b0 < 0xF0) { size = 3; if (b0 == 0xE0) lo = 0xA0; else if (b0 == 0xED) hi = 0x9F; }
    else { size = 4; if (b0 == 0xF0) lo = 0x90; else if (b0 == 0xF4) hi = 0x8F; }
    if (n < size) return 0xFFFD;
    const uint32_t b1 = p[1];
    if (b1 < lo || b1 > hi) return 0xFFFD;
    if (size == 2) { width = 2; return ((b0 & 0x1F) << 6) | (b1 & 0x3F); }
    const uint32_t b2 = p[2];
    if (b2 < 0x80 || b2 > 0xBF) return 0xFFFD;
    if (size == 3) { width = 3; return ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F); }
    const uint32_t b3 = p[3];
    if (b3 < 0x80 || b3 > 0xBF) return 0xFFFD;
    width = 4;
    return ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F);
}

// the same decoder on the first bytes of the rune held in a register (little endian: byte 0 in bits 0-7)
KNOX_RX_HD inline uint32_t decode_rune_word(uint32_t wd, uint32_t n, uint32_t &width) {
    const uint32_t b0 = wd & 0xff;
    width = 1;
    if (b0 < 0x80) return b0;
    if (b0 < 0xC2 || b0 > 0xF4) return 0xFFFD;
    uint32_t size, lo = 0x80, hi = 0xBF;
    if (b0 < 0xE0) size = 2;
    else if (b0 < 0xF0) { size = 3; if (b0 == 0xE0) lo = 0xA0; else if (b0 == 0xED) hi = 0x9F; }
    else { size = 4; if (b0 == 0xF0) lo = 0x90; else if (b0 == 0xF4) hi = 0x8F; }
    if (n < size) return 0xFFFD;
    const uint32_t b1 = (wd >> 8) & 0xff;
    if (b1 < lo || b1 > hi) return 0xFFFD;
    if (size == 2) { width = 2; return ((b0 & 0x1F) << 6) | (b1 & 0x3F); }
    const uint32_t b2 = (wd >> 16) & 0xff;
    if (b2 < 0x80 || b2 > 0xBF) return 0xFFFD;
    if (size == 3) { width = 3; return ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F); }
    const uint32_t b3 = wd >> 24;
    if (b3 < 0x80 || b3 > 0xBF) return 0xFFFD;
    width = 4;
    return ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F);
}

// host-side walk of the compiled program (unit tests of the compiler; the product path is the kernel)
inline bool match_host(const Program &pg, const uint8_t *s, size_t n) {
    uint32_t state = 1;
    size_t pos = 0;
    while (pos < n) {
        uint32_t w;
        const uint32_t r = decode_rune(s + pos, (uint32_t)std::min<size_t>(n - pos, 4), w);
        pos += w;
        uint32_t cls;
        if (r < 0x80) cls = pg.ascii_class[r];
        else {
            size_t lo = 0, hi = pg.bound_lo.size();
            while (hi - lo > 1) {
                const size_t mid = (lo + hi) >> 1;
                if (pg.bound_lo[mid] <= r) lo = mid; else hi = mid;
            }
            cls = pg.bound_class[lo];
        }
        const uint16_t e = pg.table[(size_t)state * pg.n_classes + cls];
        if (e == 0) return true;
        if (e & kDeadBit) return false;
        state = e;
    }
    return pg.flags[state] & 1;
}

}  // namespace knox_regex
