// Sanitizer fuzz of the host-side regexp compiler (knoxdb_amd/csrc/knox_regex.h): random pattern fragments, the
// compiled program must satisfy every bound the device kernel relies on.  Built with -fsanitize=address,undefined
// by tests/test_regexp.py (CPU only: GPU sanitizers are not available on the pool).
#include "knox_regex.h"
#include <cstdio>
#include <random>
int main(int argc, char **argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 100000;
    std::mt19937_64 rng(argc > 2 ? atol(argv[2]) : 1);
    const char *frag[] = {"a","b","c","0","9"," ","-","_",".","*","+","?","|","(",")","(?:","(?i)","(?s)","(?m)","(?U)","(?P<n>","(?<n>","[","]","[^","^","$","\\","\\d","\\w","\\s","\\D","\\W","\\S",
        "\\b","\\B","\\A","\\z","\\Q","\\E","\\x","\\x{","}","{","{2}","{1,3}","{2,}","{,2}","{1001}","[:alpha:]","[:^digit:]","[:bogus:]","\\p{L}","\\1","\\07","\\0","\xc3\xa9","\xe2\x82\xac","\xf0\x9f\x98\x80","\xff","\x80","-","a-z","z-a","\\n","\\t","(?-i)","(?i:","(?z)","\\xZZ","\\x{110000}","\\x{10FFFF}","{1000}","(a|b)","[ab]*a[ab]{8}",
        "\\pL","\\PN","\\p{Greek}","\\P{Lu}","\\p{^Han}","\\p{Any}","\\p{Cn}","\\p{old italic}","\\p{Nope}","\\p{","\\p","\\p{^}","\xcf\x89","\xc3\x9f","\xd0\xb2-\xd1\x8f","\xe1\xba\x9e"};
    const int nf = sizeof frag / sizeof frag[0];
    const char *text[] = {"", "a", "abc", "aAbB09 _-", "\n", "a\nb", "\xc3\xa9\xe2\x82\xac", "\xff\xfe", "\xe2\x82", "\xf0\x9f\x98\x80x", "zzzzabababababab"};
    long ok = 0, unsup = 0, big = 0;
    for (long it = 0; it < iters; it++) {
        std::string p;
        const int k = 1 + rng() % 8;
        for (int i = 0; i < k; i++) p += frag[rng() % nf];
        if (rng() % 16 == 0 && !p.empty()) p[rng() % p.size()] = (char)(rng() % 256);
        knox_regex::Program pg;
        const int st = knox_regex::compile((const uint8_t *)p.data(), p.size(), pg);
        if (st == knox_regex::OK) {
            ok++;
            for (const char *t : text) (void)knox_regex::match_host(pg, (const uint8_t *)t, strlen(t));
            // invariants the device relies on
            if (pg.n_states < 2 || pg.n_classes < 1 || pg.table.size() != (size_t)pg.n_states * pg.n_classes || pg.flags.size() != pg.n_states) { printf("bad program for %s\n", p.c_str()); return 1; }
            for (uint16_t e : pg.table) if ((e & 0x7fff) >= pg.n_states) { printf("bad target for %s\n", p.c_str()); return 1; }
            for (int c = 0; c < 128; c++) if (pg.ascii_class[c] >= pg.n_classes) { printf("bad class for %s\n", p.c_str()); return 1; }
            for (uint8_t c : pg.bound_class) if (c >= pg.n_classes) { printf("bad bound class for %s\n", p.c_str()); return 1; }
            if (pg.bound_lo.empty() || pg.bound_lo[0] != 0x80) { printf("bad bounds for %s\n", p.c_str()); return 1; }
        } else if (st == knox_regex::TOO_LARGE) big++;
        else unsup++;
    }
    printf("compiled %ld, unsupported %ld, too large %ld\n", ok, unsup, big);
    return 0;
}
