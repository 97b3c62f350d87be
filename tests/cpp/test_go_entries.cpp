// test_go_entries.cpp -- one block per exported function of integration/go/knoxgpu/knoxgpu.go, making the SAME C calls
// with the SAME argument marshalling the Go wrapper makes (there is no Go toolchain in the image; integration/
// check_go_calls.py checks the Go call sites against the header, this file checks that those call sequences do what
// the Go side expects, on the GPU).  Expected values come from a scalar loop over the same data -- the semantics of
// internal/cmp/generic.go and internal/reducer/reducer.go.
// Build: g++ tests/cpp/test_go_entries.cpp -Iinclude -Lknoxdb_amd -lknoxgpu
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "knoxgpu.h"

static int fails = 0;
#define EXPECT(c)                                                        \
    do {                                                                 \
        if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); fails++; } \
    } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}

template <typename T> static bool hit(int op, T v, T a, T b) {
    switch (op) {
    case KNOX_EQ: return v == a;
    case KNOX_NE: return v != a;
    case KNOX_GT: return v > a;
    case KNOX_GE: return v >= a;
    case KNOX_LT: return v < a;
    case KNOX_LE: return v <= a;
    default: return v >= a && v <= b;
    }
}

// knoxgpu.Cmp[T]: knox_cmp(ctx, dtype, mode, &src[0], len, &a, &b, &bits[0]) -> (n, n >= 0)
template <typename T> static void go_Cmp(knox_ctx *ctx, int dtype, size_t n) {
    std::vector<T> src(n);
    for (auto &v : src) v = (T)(int64_t)(rnd() % 41) - (T)(std::is_signed<T>::value || std::is_floating_point<T>::value ? 20 : 0);
    const int ops[7] = {KNOX_EQ, KNOX_NE, KNOX_GT, KNOX_GE, KNOX_LT, KNOX_LE, KNOX_RANGE};
    for (int op : ops) {
        T a = (T)5, b = (T)17;
        std::vector<uint8_t> bits((n + 7) / 8 + 1, 0);
        bits.back() = 0xa5;  // the poison tail of internal/cmp/tests/tests.go:24-48
        int64_t got = knox_cmp(ctx, dtype, op, src.data(), n, &a, &b, bits.data());
        int64_t want = 0;
        bool same = true;
        for (size_t i = 0; i < n; i++) {
            bool h = hit<T>(op, src[i], a, b);
            want += h;
            same &= (((bits[i >> 3] >> (i & 7)) & 1) != 0) == h;
        }
        EXPECT(got == want);
        EXPECT(same);
        EXPECT(bits.back() == 0xa5);
    }
    // len(src) == 0: the Go wrapper returns (0, true) without calling; the C entry does the same
    T a = 0;
    EXPECT(knox_cmp(ctx, dtype, KNOX_EQ, nullptr, 0, &a, &a, nullptr) == 0);
}

// knoxgpu.MatchWide: upload limbs (uint64 columns), alloc bits (+ mask), knox_match_wide, download, free
static void go_MatchWide(knox_ctx *ctx, int n_limbs, size_t n, bool with_mask) {
    std::vector<std::vector<uint64_t>> limbs(n_limbs, std::vector<uint64_t>(n));
    for (size_t i = 0; i < n; i++) {
        for (int l = 0; l < n_limbs; l++) limbs[l][i] = 0;
        limbs[0][i] = (rnd() & 1) ? ~0ull : 0;  // sign limb: -1 or 0 in the top limb
        limbs[n_limbs - 1][i] = rnd() % 64;
        if (limbs[0][i])
            for (int l = 1; l < n_limbs - 1; l++) limbs[l][i] = ~0ull;
        if (limbs[0][i]) limbs[n_limbs - 1][i] = ~0ull - limbs[n_limbs - 1][i];
    }
    std::vector<uint64_t> a(n_limbs, 0), b(n_limbs, 0);
    a[n_limbs - 1] = 10, b[n_limbs - 1] = 40;
    std::vector<uint8_t> mask((n + 7) / 8);
    for (auto &m : mask) m = (uint8_t)rnd();
    const int ops[7] = {KNOX_EQ, KNOX_NE, KNOX_GT, KNOX_GE, KNOX_LT, KNOX_LE, KNOX_RANGE};
    for (int op : ops) {
        std::vector<knox_col *> cols(n_limbs, nullptr);
        for (int l = 0; l < n_limbs; l++) EXPECT(knox_col_upload(ctx, KNOX_UINT64, limbs[l].data(), n, 0, &cols[l]) == 0);
        knox_bits *dbits = nullptr, *dmask = nullptr;
        EXPECT(knox_bits_alloc(ctx, n, &dbits) == 0);
        if (with_mask) {
            EXPECT(knox_bits_alloc(ctx, n, &dmask) == 0);
            EXPECT(knox_bits_upload(ctx, dmask, mask.data()) == 0);
        }
        int64_t cnt = -1;
        EXPECT(knox_match_wide(ctx, cols.data(), n_limbs, op, a.data(), op == KNOX_RANGE ? b.data() : nullptr, dmask, dbits, &cnt) == 0);
        std::vector<uint8_t> bits((n + 7) / 8, 0);
        EXPECT(knox_bits_download(ctx, dbits, bits.data()) == 0);
        int64_t want = 0;
        bool same = true;
        for (size_t i = 0; i < n; i++) {
            // the test values are small or small-negative: compare as signed 64-bit of (sign, low limb)
            int64_t v = limbs[0][i] ? -(int64_t)(~limbs[n_limbs - 1][i]) - 1 : (int64_t)limbs[n_limbs - 1][i];
            bool h = hit<int64_t>(op, v, 10, 40);
            if (with_mask) h = h && ((mask[i >> 3] >> (i & 7)) & 1);
            want += h;
            same &= (((bits[i >> 3] >> (i & 7)) & 1) != 0) == h;
        }
        EXPECT(cnt == want);
        EXPECT(same);
        knox_bits_free(dbits);
        if (dmask) knox_bits_free(dmask);
        for (auto c : cols) knox_col_free(c);
    }
}

// pkg/num/varint.go:161-197 (SQLite4-style prefix varint), the ranges a container header of this test needs
static void put_uvarint(std::vector<uint8_t> &out, uint64_t v) {
    if (v <= 240) {
        out.push_back((uint8_t)v);
    } else if (v <= 2287) {
        out.push_back((uint8_t)((v - 240) / 256 + 241));
        out.push_back((uint8_t)((v - 240) % 256));
    } else if (v <= 67823) {
        out.push_back(249);
        out.push_back((uint8_t)((v - 2288) / 256));
        out.push_back((uint8_t)((v - 2288) % 256));
    } else {
        out.push_back(250);
        out.push_back((uint8_t)(v >> 16)), out.push_back((uint8_t)(v >> 8)), out.push_back((uint8_t)v);
    }
}

// run_knoxgpu.go (GpuRun): the run-level reader's C calls in its order -- NewColumn per filter column, AppendAsync pack by
// pack, NewBits, ProfEnable, ONE Scan over the run with per-pack counts (+ fused reducers), ProfRead, ONE Download, the
// per-pack slices served; Reset and the next run; then a run whose blocks arrive ENCODED (UploadRawContainers).
static void go_RunReader(knox_ctx *ctx) {
    const size_t pack = 4096, max_packs = 8;
    knox_col *ca = nullptr, *cb = nullptr;
    EXPECT(knox_col_create(ctx, KNOX_INT64, max_packs * pack, pack, &ca) == 0);
    EXPECT(knox_col_create(ctx, KNOX_INT32, max_packs * pack, pack, &cb) == 0);
    // (a BETWEEN -500 AND 500) AND (b >= 10); reducers sum(a), max(b) fused into the run's scan
    knox_pred nodes[3];
    std::memset(nodes, 0, sizeof nodes);
    nodes[0].kind = KNOX_NODE_LEAF, nodes[0].col = 0, nodes[0].op = KNOX_RANGE, nodes[0].a = (uint64_t)(int64_t)-500, nodes[0].b = 500;
    nodes[1].kind = KNOX_NODE_LEAF, nodes[1].col = 1, nodes[1].op = KNOX_GE, nodes[1].a = 10;
    nodes[2].kind = KNOX_NODE_AND, nodes[2].nchild = 2;
    knox_agg aggs[2] = {{KNOX_R_SUM, 0}, {KNOX_R_MAX, 1}};
    std::vector<knox_agg_state> run_sum, run_max;
    uint64_t row_base = 0;
    int64_t want_total = 0, got_total = 0;
    uint64_t want_sum = 0;
    int32_t want_max = INT32_MIN;
    const size_t run_packs[3] = {8, 5, 3};  // a full run, a shorter one, one whose last pack is short
    for (int run = 0; run < 3; run++) {
        const size_t np = run_packs[run];
        std::vector<std::vector<int64_t>> va(np);
        std::vector<std::vector<int32_t>> vb(np);
        EXPECT(knox_col_reset(ctx, ca) == 0 && knox_col_reset(ctx, cb) == 0);  // the copy stream waits for the previous run's scan
        size_t rows = 0;
        for (size_t p = 0; p < np; p++) {
            const size_t n = (run == 2 && p + 1 == np) ? pack - 1000 : pack;
            va[p].resize(n), vb[p].resize(n);
            for (size_t i = 0; i < n; i++) va[p][i] = (int64_t)(rnd() % 4001) - 2000, vb[p][i] = (int32_t)(rnd() % 41) - 20;
            if (p == 2)  // one pack without a match: the reference's false positive metadata match
                for (size_t i = 0; i < n; i++) va[p][i] = 100000;
            EXPECT(knox_col_append_async(ctx, ca, va[p].data(), n) == 0);  // pageable Go slices: copied inside the call
            EXPECT(knox_col_append_async(ctx, cb, vb[p].data(), n) == 0);
            std::fill(va[p].begin(), va[p].begin() + 1, va[p][0]);  // the caller's memory is its own again after return
            rows += n;
        }
        EXPECT(knox_col_rows(ca) == rows && knox_col_rows(cb) == rows);
        knox_bits *bits = nullptr;
        EXPECT(knox_bits_alloc(ctx, rows, &bits) == 0);
        EXPECT(knox_prof_enable(ctx, 1) == 0);
        knox_col *cols[2] = {ca, cb};
        knox_agg_state st[2];
        std::vector<int64_t> pc(np, -1);
        knox_scan_out out;
        EXPECT(knox_scan(ctx, nodes, 3, cols, 2, aggs, 2, row_base, nullptr, bits, pc.data(), st, &out) == 0);
        double ms = -1;
        int launches = -1;
        EXPECT(knox_prof_read(ctx, &ms, &launches) == 0 && launches >= 1 && ms > 0);
        std::vector<uint8_t> host((rows + 7) / 8);
        EXPECT(knox_bits_download(ctx, bits, host.data()) == 0);
        // serve: pack p's bitset is the slice starting at byte p * pack / 8; its count is pc[p]
        size_t first = 0;
        for (size_t p = 0; p < np; p++) {
            int64_t cnt = 0;
            bool same = true;
            for (size_t i = 0; i < va[p].size(); i++) {
                const bool h = va[p][i] >= -500 && va[p][i] <= 500 && vb[p][i] >= 10;
                cnt += h;
                same &= (((host[(first + i) >> 3] >> ((first + i) & 7)) & 1) != 0) == h;
                if (h) want_sum += (uint64_t)va[p][i], want_max = std::max(want_max, vb[p][i]);
            }
            EXPECT(same);
            EXPECT(pc[p] == cnt);
            if (p == 2) EXPECT(pc[p] == 0);
            want_total += cnt;
            first += va[p].size();
        }
        got_total += out.count;
        run_sum.push_back(st[0]), run_max.push_back(st[1]);
        row_base += rows;
        knox_bits_free(bits);
    }
    EXPECT(got_total == want_total);
    // GpuRun.Value: Combine of the runs' states in run order, then Reducer.Value()
    knox_agg_state ms_, mm_;
    EXPECT(knox_agg_combine(run_sum.data(), (int)run_sum.size(), &ms_) == 0 && knox_agg_combine(run_max.data(), (int)run_max.size(), &mm_) == 0);
    uint64_t v = 0;
    int32_t mx = 0;
    int seen = 0;
    int64_t row = -1;
    EXPECT(knox_agg_value(&ms_, KNOX_R_SUM, &v, &seen, &row) == 0 && seen && v == want_sum);
    EXPECT(knox_agg_value(&mm_, KNOX_R_MAX, &mx, &seen, &row) == 0 && seen && mx == want_max && row >= 0 && (uint64_t)row < row_base);
    // over capacity: KNOX_E_RANGE, the column stays as it was
    {
        std::vector<int64_t> big(max_packs * pack + 1, 1);
        EXPECT(knox_col_reset(ctx, ca) == 0);
        EXPECT(knox_col_append_async(ctx, ca, big.data(), big.size()) == KNOX_E_RANGE && knox_col_rows(ca) == 0);
    }
    knox_col_free(ca), knox_col_free(cb);

    // an ENCODED run: every pack's block arrives as its raw container [7][uvarint n][LE values] (int_raw.go:73-94) and the
    // run's column is made by ONE knox_col_upload_raw_containers (UploadRawContainers: C copies of the Go buffers + a C
    // array of pointers)
    {
        const size_t np = 4;
        std::vector<std::vector<uint8_t>> bufs(np);
        std::vector<std::vector<int64_t>> vals(np);
        for (size_t p = 0; p < np; p++) {
            vals[p].resize(pack);
            for (auto &x : vals[p]) x = (int64_t)(rnd() % 2001) - 1000;
            bufs[p].push_back(7);
            put_uvarint(bufs[p], pack);
            const uint8_t *raw = (const uint8_t *)vals[p].data();
            bufs[p].insert(bufs[p].end(), raw, raw + pack * 8);
        }
        uint8_t **ptrs = (uint8_t **)std::malloc(np * sizeof(uint8_t *));
        size_t *lens = (size_t *)std::malloc(np * sizeof(size_t));
        for (size_t p = 0; p < np; p++) {
            ptrs[p] = (uint8_t *)std::malloc(bufs[p].size());  // C.CBytes
            std::memcpy(ptrs[p], bufs[p].data(), bufs[p].size());
            lens[p] = bufs[p].size();
        }
        knox_col *col = nullptr;
        EXPECT(knox_col_upload_raw_containers(ctx, KNOX_INT64, ptrs, lens, np, pack, &col) == 0);
        for (size_t p = 0; p < np; p++) std::free(ptrs[p]);
        std::free(ptrs), std::free(lens);
        EXPECT(col && knox_col_rows(col) == np * pack);
        knox_pred leaf;
        std::memset(&leaf, 0, sizeof leaf);
        leaf.kind = KNOX_NODE_LEAF, leaf.op = KNOX_LT, leaf.a = (uint64_t)(int64_t)-900;
        knox_bits *bits = nullptr;
        EXPECT(knox_bits_alloc(ctx, np * pack, &bits) == 0);
        knox_col *cols[1] = {col};
        knox_agg none[1] = {{0, 0}};
        std::vector<int64_t> pc(np, -1);
        knox_scan_out out;
        EXPECT(knox_scan(ctx, &leaf, 1, cols, 1, none, 0, 0, nullptr, bits, pc.data(), nullptr, &out) == 0);
        for (size_t p = 0; p < np; p++) {
            int64_t cnt = 0;
            for (auto x : vals[p]) cnt += x < -900;
            EXPECT(pc[p] == cnt);
        }
        knox_bits_free(bits), knox_col_free(col);
    }
}

// run_knoxgpu.go (GpuRun.scanEncoded): a run whose one filter column arrived as containers and whose plan is one numeric leaf --
// UploadEncoded (C copies of the Go buffers + a C pointer array, cBufs), NewBits, ONE (*Encoded).Scan with the bitset, the per-pack
// counts and the Sum / Count state, ONE Download, Free.  The run mixes container schemes pack by pack, as the encoder would:
// dictionary [5], delta [2], raw [7], constant [1]; a short last pack.
static void go_EncodedRun(knox_ctx *ctx) {
    const size_t pack = 8192, np = 4, last_rows = 5000;
    std::vector<std::vector<uint8_t>> bufs(np);
    std::vector<std::vector<int64_t>> vals(np);
    auto put_raw = [&](std::vector<uint8_t> &out, const void *p, size_t n, size_t w) {   // [7][uvarint n][LE values]
        out.push_back(7);
        put_uvarint(out, n);
        const uint8_t *b = (const uint8_t *)p;
        out.insert(out.end(), b, b + n * w);
    };
    {   // pack 0: dictionary of 37 sorted values, uint16 codes
        std::vector<int64_t> dict(37);
        for (size_t i = 0; i < dict.size(); i++) dict[i] = -500 + (int64_t)i * 29;
        std::vector<uint16_t> codes(pack);
        vals[0].resize(pack);
        for (size_t i = 0; i < pack; i++) codes[i] = (uint16_t)(rnd() % dict.size()), vals[0][i] = dict[codes[i]];
        bufs[0].push_back(5);
        put_raw(bufs[0], dict.data(), dict.size(), 8);
        put_raw(bufs[0], codes.data(), pack, 2);
    }
    {   // pack 1: delta container, row i = -1000 + 3 i
        bufs[1].push_back(2);
        std::vector<uint8_t> &b = bufs[1];
        // For = uint64(int64(-1000)) needs the 9-byte varint form: 255 + 8 big-endian bytes (pkg/num/varint.go:161-197)
        const uint64_t f = (uint64_t)(int64_t)-1000;
        b.push_back(255);
        for (int k = 7; k >= 0; k--) b.push_back((uint8_t)(f >> (8 * k)));
        put_uvarint(b, 3);
        put_uvarint(b, pack);
        vals[1].resize(pack);
        for (size_t i = 0; i < pack; i++) vals[1][i] = -1000 + 3 * (int64_t)i;
    }
    {   // pack 2: raw
        vals[2].resize(pack);
        for (size_t i = 0; i < pack; i++) vals[2][i] = (int64_t)(rnd() % 4000) - 2000;
        put_raw(bufs[2], vals[2].data(), pack, 8);
    }
    {   // pack 3 (short): constant 123
        bufs[3].push_back(1);
        put_uvarint(bufs[3], 123);
        put_uvarint(bufs[3], last_rows);
        vals[3].assign(last_rows, 123);
    }
    const uint8_t *ptrs[np];
    size_t lens[np];
    size_t rows = 0;
    for (size_t p = 0; p < np; p++) ptrs[p] = bufs[p].data(), lens[p] = bufs[p].size(), rows += vals[p].size();
    knox_enc *enc = nullptr;
    EXPECT(knox_enc_upload(ctx, KNOX_INT64, ptrs, lens, np, pack, &enc) == 0);
    if (!enc) return;
    EXPECT(knox_enc_rows(enc) == rows);
    size_t nd = 9, nr = 9, na = 9, nw = 9;
    EXPECT(knox_enc_kinds(enc, &nd, &nr, &na, &nw) == 0 && nd == 1 && nr == 0 && na == 2 && nw == 1);
    knox_bits *bits = nullptr;
    EXPECT(knox_bits_alloc(ctx, rows, &bits) == 0);
    std::vector<int64_t> pc(np, -1);
    int64_t cnt = -1;
    knox_agg_state st;
    // leaf: v BETWEEN -300 AND 123 (ValueBits: the raw bits of the column type)
    const int64_t lo = -300, hi = 123;
    EXPECT(knox_enc_scan(ctx, enc, KNOX_RANGE, (uint64_t)lo, (uint64_t)hi, 1, bits, pc.data(), &cnt, &st) == 0);
    std::vector<uint8_t> host((rows + 7) / 8);
    EXPECT(knox_bits_download(ctx, bits, host.data()) == 0);
    int64_t want = 0;
    uint64_t want_sum = 0;
    bool same = true;
    size_t base = 0;
    for (size_t p = 0; p < np; p++) {
        int64_t c = 0;
        for (size_t i = 0; i < vals[p].size(); i++) {
            const bool h = vals[p][i] >= lo && vals[p][i] <= hi;
            c += h;
            if (h) want_sum += (uint64_t)vals[p][i];
            const size_t r = base + i;
            same &= (((host[r >> 3] >> (r & 7)) & 1) != 0) == h;
        }
        EXPECT(pc[p] == c);
        want += c;
        base += pack;   // packs start pack_rows apart on the device
    }
    EXPECT(cnt == want && same);
    uint64_t vbits = 0;
    int seen = 0;
    int64_t row = 0;
    EXPECT(knox_agg_value(&st, KNOX_R_SUM, &vbits, &seen, &row) == 0 && seen == 1 && vbits == want_sum);
    EXPECT(knox_agg_value(&st, KNOX_R_COUNT, &vbits, &seen, &row) == 0 && (int64_t)vbits == want);
    knox_bits_free(bits);
    knox_enc_free(enc);
}

// operator_knoxgpu.go (deviceCols.get / evictOthers): under a tiny arena budget an upload fails with KNOX_E_NOMEM; the
// cache frees columns of OTHER packages only and retries; the handles of the package being planned stay valid.
static void go_EvictOthers() {
    knox_ctx *ctx = nullptr;
    if (knox_ctx_open(0, &ctx) != 0) return;
    // 8 GiB: the arena's own 1 GiB heap chunk + a 32 MiB reference window + three columns (a 1 GiB column with its
    // padding takes two 1 GiB chunks of ONE class zone); the fourth upload does not fit.  (6 GiB until round 4: on a box whose
    // first four chunks fall into three or four placement classes -- each class's founder chunk carries its reference
    // window -- the FIRST column found no two chunks of one class inside the budget and the test failed before it began.)
    EXPECT(knox_ctx_set_arena(ctx, (size_t)8 << 30) == 0);
    const size_t n = (size_t)1 << 27;                       // 1 GiB of int64 per column
    std::vector<int64_t> v(n);
    for (size_t i = 0; i < n; i++) v[i] = (int64_t)(i & 1023);
    std::vector<knox_col *> resident;   // d.order: columns of earlier packages
    std::vector<knox_col *> mine;       // columns of the package being planned
    int failures = 0;
    for (int k = 0; k < 6; k++) {
        knox_col *c = nullptr;
        int rc = knox_col_upload(ctx, KNOX_INT64, v.data(), n, 0, &c);
        if (rc == KNOX_E_NOMEM) {
            failures++;
            // evictOthers: the oldest columns that do not belong to this package
            EXPECT(!resident.empty());
            if (resident.empty()) break;
            const size_t drop = (resident.size() + 1) / 2;
            for (size_t i = 0; i < drop; i++) knox_col_free(resident[i]);
            resident.erase(resident.begin(), resident.begin() + (long)drop);
            rc = knox_col_upload(ctx, KNOX_INT64, v.data(), n, 0, &c);
        }
        EXPECT(rc == 0 && c != nullptr);
        if (!c) break;
        if (k < 4) resident.push_back(c);  // earlier packages
        else mine.push_back(c);            // two leaves of the current package: both handles must survive the evictions
    }
    EXPECT(failures >= 1);                 // the budget really was too small for everything
    EXPECT(mine.size() == 2);
    if (mine.size() == 2) {
        knox_pred nodes[3];
        std::memset(nodes, 0, sizeof nodes);
        nodes[0].kind = KNOX_NODE_LEAF, nodes[0].col = 0, nodes[0].op = KNOX_LT, nodes[0].a = 8;
        nodes[1].kind = KNOX_NODE_LEAF, nodes[1].col = 1, nodes[1].op = KNOX_GE, nodes[1].a = 4;
        nodes[2].kind = KNOX_NODE_AND, nodes[2].nchild = 2;
        knox_agg none[1] = {{0, 0}};
        knox_scan_out out;
        EXPECT(knox_scan(ctx, nodes, 3, mine.data(), 2, none, 0, 0, nullptr, nullptr, nullptr, nullptr, &out) == 0);
        EXPECT(out.count == (int64_t)(n / 1024 * 4));
    }
    for (auto c : resident) knox_col_free(c);
    for (auto c : mine) knox_col_free(c);
    knox_ctx_close(ctx);
}

int main() {
    // knoxgpu.Open / (*Ctx).SetArena / errOf
    knox_ctx *ctx = nullptr;
    if (knox_ctx_open(0, &ctx) != 0) {
        std::printf("no device\n");
        return 2;
    }
    EXPECT(knox_ctx_set_arena(ctx, (size_t)8 << 30) == 0);
    EXPECT(std::string(knox_strerror(KNOX_E_INVALID)).size() > 0);

    // Cmp[T] for the 10 numeric types x 7 modes (the 70 func vars of cmp.go:6-114)
    const size_t N = 1000 + 37;
    go_Cmp<int64_t>(ctx, KNOX_INT64, N), go_Cmp<int32_t>(ctx, KNOX_INT32, N), go_Cmp<int16_t>(ctx, KNOX_INT16, N);
    go_Cmp<int8_t>(ctx, KNOX_INT8, N), go_Cmp<uint64_t>(ctx, KNOX_UINT64, N), go_Cmp<uint32_t>(ctx, KNOX_UINT32, N);
    go_Cmp<uint16_t>(ctx, KNOX_UINT16, N), go_Cmp<uint8_t>(ctx, KNOX_UINT8, N), go_Cmp<double>(ctx, KNOX_FLOAT64, N);
    go_Cmp<float>(ctx, KNOX_FLOAT32, N);

    // MatchWide: Int128 (2 limbs) and Int256 (4 limbs), with and without a mask (the 14 wide func vars)
    go_MatchWide(ctx, 2, 777, false), go_MatchWide(ctx, 2, 777, true);
    go_MatchWide(ctx, 4, 515, false), go_MatchWide(ctx, 4, 515, true);

    // Upload / NewBitsFor / (*Ctx).Scan with aggregates / (*State).Value / Combine / (*Bits).Indexes / Download
    {
        const size_t n = 3 * 4096 + 100, pack = 4096, n_packs = 4;
        std::vector<uint64_t> u(n);
        std::vector<double> f(n);
        for (size_t i = 0; i < n; i++) u[i] = rnd() % 1000, f[i] = (double)(rnd() % 4096) / 16.0;
        knox_col *cu = nullptr, *cf = nullptr;
        EXPECT(knox_col_upload(ctx, KNOX_UINT64, u.data(), n, pack, &cu) == 0);
        EXPECT(knox_col_upload(ctx, KNOX_FLOAT64, f.data(), n, pack, &cf) == 0);
        knox_bits *bits = nullptr;
        EXPECT(knox_bits_alloc_for(ctx, n, cu, &bits) == 0);
        // postfix: (u BETWEEN 100 AND 499) AND (f < 128.0)
        knox_pred nodes[3];
        std::memset(nodes, 0, sizeof nodes);
        nodes[0].kind = KNOX_NODE_LEAF, nodes[0].col = 0, nodes[0].op = KNOX_RANGE, nodes[0].a = 100, nodes[0].b = 499;
        double lim = 128.0;
        nodes[1].kind = KNOX_NODE_LEAF, nodes[1].col = 1, nodes[1].op = KNOX_LT;
        std::memcpy(&nodes[1].a, &lim, 8);
        nodes[2].kind = KNOX_NODE_AND, nodes[2].nchild = 2;
        knox_col *cols[2] = {cu, cf};
        knox_agg aggs[3] = {{KNOX_R_SUM, 0}, {KNOX_R_MAX, 1}, {KNOX_R_MEAN, 1}};
        knox_agg_state st[3];
        std::vector<int64_t> pc(n_packs, -1);
        std::vector<uint8_t> skip(n_packs, 0);
        skip[1] = 1;  // pack 1 pruned by the caller's statistics
        knox_scan_out out;
        EXPECT(knox_scan(ctx, nodes, 3, cols, 2, aggs, 3, 0, skip.data(), bits, pc.data(), st, &out) == 0);
        int64_t want = 0, wpc[4] = {0, 0, 0, 0};
        uint64_t wsum = 0;
        double wmax = -1, wfs = 0;
        int64_t wmax_row = -1;
        std::vector<uint32_t> widx;
        for (size_t i = 0; i < n; i++) {
            if (i / pack == 1) continue;
            if (u[i] >= 100 && u[i] <= 499 && f[i] < 128.0) {
                want++, wpc[i / pack]++, wsum += u[i], wfs += f[i];
                if (f[i] > wmax) wmax = f[i], wmax_row = (int64_t)i;
                widx.push_back((uint32_t)i);
            }
        }
        if (out.count != want)
            std::printf("scan: count %lld want %lld, rows_scanned %lld want %lld, pack counts %lld %lld %lld %lld\n", (long long)out.count,
                        (long long)want, (long long)out.rows_scanned, (long long)(n - pack), (long long)pc[0], (long long)pc[1], (long long)pc[2],
                        (long long)pc[3]);
        EXPECT(out.count == want);
        // rows actually read: the pruned pack is left unread when the shape allows it (whole power-of-two packs)
        EXPECT(out.rows_scanned == (int64_t)(n - pack) || out.rows_scanned == (int64_t)n);
        for (int p = 0; p < 4; p++) EXPECT(pc[p] == wpc[p]);
        uint64_t v = 0;
        int seen = 0;
        int64_t row = -1;
        EXPECT(knox_agg_value(&st[0], KNOX_R_SUM, &v, &seen, &row) == 0 && seen && v == wsum);
        double d = 0;
        EXPECT(knox_agg_value(&st[1], KNOX_R_MAX, &d, &seen, &row) == 0 && seen && d == wmax && row == wmax_row);
        EXPECT(knox_agg_value(&st[2], KNOX_R_MEAN, &d, &seen, &row) == 0 && seen && d == wfs / (double)want);  // exact: 1/16 grid
        // Combine of {state, state} doubles the sum and the count
        knox_agg_state two[2] = {st[0], st[0]}, merged;
        EXPECT(knox_agg_combine(two, 2, &merged) == 0);
        EXPECT(merged.count == 2 * st[0].count && merged.isum == 2 * st[0].isum);
        // Indexes into a Go slice with cap = count
        std::vector<uint32_t> idx((size_t)want + 1);
        size_t n_idx = 0;
        EXPECT(knox_bits_indexes(ctx, bits, idx.data(), (size_t)want, &n_idx) == 0);
        EXPECT(n_idx == (size_t)want);
        idx.resize(n_idx);
        EXPECT(idx == widx);
        std::vector<uint8_t> host((n + 7) / 8);
        EXPECT(knox_bits_download(ctx, bits, host.data()) == 0);
        int64_t pop = 0;
        for (auto b : host) pop += __builtin_popcount(b);
        EXPECT(pop == want);
        knox_bits_free(bits), knox_col_free(cu), knox_col_free(cf);
    }

    // NewBytesColumn / AppendPack / BuildBloom / HashBytes / (*Bloom).Probe / Scan with a bytes leaf (C.CBytes copy)
    {
        const size_t pack = 1024, n_packs = 3;
        knox_col *col = nullptr;
        EXPECT(knox_col_bytes_create(ctx, pack, n_packs, 1 << 20, &col) == 0);
        std::vector<std::string> all;
        for (size_t p = 0; p < n_packs; p++) {
            std::vector<uint64_t> ptr(pack);
            std::vector<uint8_t> buf;
            for (size_t i = 0; i < pack; i++) {
                std::string s = "key-" + std::to_string(p * 100000 + rnd() % 500);
                ptr[i] = ((uint64_t)buf.size() << 32) | s.size();
                buf.insert(buf.end(), s.begin(), s.end());
                all.push_back(s);
            }
            EXPECT(knox_col_bytes_append_pack(ctx, col, ptr.data(), pack, buf.data(), buf.size()) == 0);
        }
        const std::string needle = all[pack + 17];  // lives in pack 1 only
        knox_bloom *bl = nullptr;
        EXPECT(knox_bloom_build(ctx, col, 2, &bl) == 0);
        std::vector<uint8_t> maybe(n_packs, 9);
        uint64_t h = knox_hash_bytes((const uint8_t *)needle.data(), needle.size());
        EXPECT(knox_bloom_probe(ctx, bl, h, maybe.data()) == 0);
        EXPECT(maybe[1] == 1);
        std::vector<uint8_t> skip(n_packs);
        for (size_t p = 0; p < n_packs; p++) skip[p] = maybe[p] ? 0 : 1;
        void *cval = std::malloc(needle.size());  // C.CBytes
        std::memcpy(cval, needle.data(), needle.size());
        knox_pred leaf;
        std::memset(&leaf, 0, sizeof leaf);
        leaf.kind = KNOX_NODE_LEAF, leaf.op = KNOX_EQ, leaf.val = (const uint8_t *)cval, leaf.val_len = needle.size();
        knox_col *cols[1] = {col};
        knox_agg none[1] = {{0, 0}};
        knox_scan_out out;
        EXPECT(knox_scan(ctx, &leaf, 1, cols, 1, none, 0, 0, skip.data(), nullptr, nullptr, nullptr, &out) == 0);
        int64_t want = 0;
        for (auto &s : all) want += s == needle;
        EXPECT(out.count == want);
        std::free(cval);
        knox_bloom_free(bl), knox_col_free(col);
    }

    // gpuBytesMatcher.MatchVector (integration/go/bytes_knoxgpu.go): UploadPool = NewBytesColumn(n, 1, len(buf)) + AppendPack,
    // NewBits, Scan with ONE bytes leaf (compare modes: val / val2; REGEXP: the pattern), Download, ResetCount(res.Count),
    // bits.And(mask) on the host; RegexpUnicodeVersion; MatchRegexp with a device mask; ArenaClasses
    {
        const size_t n = 70000;
        const char *words[] = {"stra\xc3\x9f" "e", "STRA\xe1\xba\x9e" "E", "\xcf\x89\xce\xbc\xce\xad\xce\xb3\xce\xb1", "\xce\xa9\xce\x9c", "key-17", "KEY-17", "\xe6\xbc\xa2\xe5\xad\x97", "\xff\xfe", "", "x9"};
        std::vector<uint64_t> ptr(n);
        std::vector<uint8_t> buf;
        std::vector<std::string> rows(n);
        for (size_t i = 0; i < n; i++) {
            std::string r = words[rnd() % 10];
            if (rnd() % 3 == 0) r += words[rnd() % 10];
            rows[i] = r;
            ptr[i] = ((uint64_t)buf.size() << 32) | r.size();
            buf.insert(buf.end(), r.begin(), r.end());
        }
        buf.resize(buf.size() + 8);
        EXPECT(std::strcmp(knox_regexp_unicode_version(), "17.0.0") == 0);
        knox_col *col = nullptr;
        EXPECT(knox_col_bytes_create(ctx, n, 1, buf.size(), &col) == KNOX_E_INVALID && col == nullptr);  // pack sizes are multiples of 64 rows
        {
            knox_pred lf;
            std::memset(&lf, 0, sizeof lf);
            knox_col *none_cols[1] = {nullptr};
            knox_agg na[1] = {{0, 0}};
            knox_scan_out o;
            EXPECT(knox_scan(ctx, &lf, 1, none_cols, 1, na, 0, 0, nullptr, nullptr, nullptr, nullptr, &o) == KNOX_E_INVALID);  // a nil column is refused, not followed
        }
        EXPECT(knox_col_bytes_create(ctx, (n + 63) & ~(size_t)63, 1, buf.size(), &col) == 0);
        EXPECT(knox_col_bytes_append_pack(ctx, col, ptr.data(), n, buf.data(), buf.size()) == 0);
        knox_bits *dbits = nullptr;
        EXPECT(knox_bits_alloc(ctx, n, &dbits) == 0);
        std::vector<uint8_t> mask((n + 7) / 8), host((n + 7) / 8);
        for (auto &b : mask) b = (uint8_t)rnd();
        mask.back() &= (uint8_t)(0xff >> ((8 - n % 8) % 8));
        struct Case { int op; const char *val, *val2; } cases[] = {
            {KNOX_EQ, "key-17", nullptr}, {KNOX_NE, "key-17", nullptr}, {KNOX_LT, "key-17", nullptr}, {KNOX_GE, "\xce\xa9", nullptr},
            {KNOX_RANGE, "KEY", "stra\xc3\x9f" "e"}, {KNOX_REGEXP, "(?i)^stra\xc3\x9f" "e$", nullptr}, {KNOX_REGEXP, "^\\p{Greek}+$", nullptr},
            {KNOX_REGEXP, "\\p{Han}{2}|\\p{Nd}$", nullptr}, {KNOX_REGEXP, "(?i)\xcf\x89\xce\xbc", nullptr},
            {KNOX_IN, nullptr, nullptr}, {KNOX_NOTIN, nullptr, nullptr}};
        // setBytes([][]byte) of operator_knoxgpu.go: [uint32 little-endian length][bytes] entries, any order, duplicates allowed
        const std::vector<std::string> set_members = {"key-17", "x9", "", "key-17", "\xff\xfe"};
        std::string packed_set;
        for (const std::string &m : set_members) {
            const uint32_t l = (uint32_t)m.size();
            packed_set.append((const char *)&l, 4);
            packed_set += m;
        }
        for (const Case &c : cases) {
            const std::string v = c.val ? std::string(c.val) : packed_set, v2 = c.val2 ? c.val2 : "";
            knox_pred leaf;
            std::memset(&leaf, 0, sizeof leaf);
            leaf.kind = KNOX_NODE_LEAF, leaf.op = c.op, leaf.val = (const uint8_t *)v.data(), leaf.val_len = v.size();
            if (c.val2) leaf.val2 = (const uint8_t *)v2.data(), leaf.val2_len = v2.size();
            knox_col *cols[1] = {col};
            knox_agg none[1] = {{0, 0}};
            knox_scan_out out;
            EXPECT(knox_scan(ctx, &leaf, 1, cols, 1, none, 0, 0, nullptr, dbits, nullptr, nullptr, &out) == 0);
            EXPECT(knox_bits_download(ctx, dbits, host.data()) == 0);
            int64_t want = 0, got = 0;
            bool same = true;
            for (size_t i = 0; i < n; i++) {
                const std::string &r = rows[i];
                bool w;
                switch (c.op) {
                case KNOX_EQ: w = r == v; break;
                case KNOX_NE: w = r != v; break;
                case KNOX_LT: w = r < v; break;   // std::string compares bytes as unsigned chars, like bytes.Compare
                case KNOX_GE: w = r >= v; break;
                case KNOX_RANGE: w = r >= v && r <= v2; break;
                case KNOX_IN: w = std::find(set_members.begin(), set_members.end(), r) != set_members.end(); break;
                case KNOX_NOTIN: w = std::find(set_members.begin(), set_members.end(), r) == set_members.end(); break;
                default: w = knox_regexp_match_value((const uint8_t *)v.data(), v.size(), (const uint8_t *)r.data(), r.size()) == 1; break;
                }
                const bool g = (host[i >> 3] >> (i & 7)) & 1;
                same = same && g == w;
                want += w, got += g;
            }
            EXPECT(same && got == want && out.count == want);
            if (c.op == KNOX_REGEXP) {  // MatchRegexp with a device mask: only masked rows are set
                knox_bits *dm = nullptr, *d2 = nullptr;
                EXPECT(knox_bits_alloc(ctx, n, &dm) == 0 && knox_bits_alloc(ctx, n, &d2) == 0);
                EXPECT(knox_bits_upload(ctx, dm, mask.data()) == 0);
                std::vector<uint8_t> zero((n + 7) / 8, 0), h2((n + 7) / 8);
                EXPECT(knox_bits_upload(ctx, d2, zero.data()) == 0);
                int64_t cnt = -1;
                EXPECT(knox_match_regexp(ctx, col, (const uint8_t *)v.data(), v.size(), dm, d2, &cnt) == 0);
                EXPECT(knox_bits_download(ctx, d2, h2.data()) == 0);
                int64_t pm = 0;
                bool ok = true;
                for (size_t b = 0; b < h2.size(); b++) {
                    ok = ok && h2[b] == (uint8_t)(host[b] & mask[b]);  // = bits.And(mask) of the Go matcher
                    pm += __builtin_popcount(h2[b]);
                }
                EXPECT(ok && cnt == pm);
                knox_bits_free(dm), knox_bits_free(d2);
            }
        }
        uint64_t cb[4];
        int seen = -1;
        EXPECT(knox_ctx_arena_classes(ctx, cb, &seen) == 0 && seen >= 0 && seen <= 4);
        knox_bits_free(dbits), knox_col_free(col);
    }

    // CommInit / AllReduceSumCount / AllGatherStates with one rank (the N-rank form runs under bench.py --gpus N)
    {
        uint8_t id[KNOX_COMM_ID_BYTES];
        int rc = knox_comm_unique_id(id);
        if (rc == KNOX_E_COMM || rc == KNOX_E_UNSUPPORTED) {
            std::printf("comm: RCCL not available here, skipped\n");
        } else {
            EXPECT(rc == 0);
            EXPECT(knox_comm_init(ctx, 1, 0, id) == 0);
            uint64_t s = 0xfedcba9876543210ull;
            int64_t c = 12345;
            EXPECT(knox_comm_allreduce_sum_count(ctx, &s, &c) == 0);
            EXPECT(s == 0xfedcba9876543210ull && c == 12345);
            knox_agg_state mine, merged;
            std::memset(&mine, 0, sizeof mine);
            mine.dtype = KNOX_UINT64, mine.flags = 1, mine.count = 3, mine.isum = 42;
            mine.min_row = mine.max_row = mine.first_row = mine.last_row = -1;
            int64_t cnt = 3;
            EXPECT(knox_comm_allgather_states(ctx, &mine, 1, &merged, &cnt) == 0);
            EXPECT(cnt == 3 && merged.isum == 42 && merged.count == 3);
            // AllReduceSumCountStart / Finish (FIFO of 4 on the collective stream) and CommRanks = ncclCommCount
            for (int k = 0; k < 4; k++) EXPECT(knox_comm_allreduce_sum_count_start(ctx, 1000ull + (uint64_t)k, 10 + k) == 0);
            EXPECT(knox_comm_allreduce_sum_count_start(ctx, 1, 1) == KNOX_E_BUSY);
            for (int k = 0; k < 4; k++) {
                uint64_t s2 = 0;
                int64_t c2 = 0;
                EXPECT(knox_comm_allreduce_sum_count_finish(ctx, &s2, &c2) == 0 && s2 == 1000ull + (uint64_t)k && c2 == 10 + k);
            }
            int nr = 0, rk = -1;
            EXPECT(knox_comm_ranks(ctx, &nr, &rk) == 0 && nr == 1 && rk == 0);
            knox_comm_destroy(ctx);
        }
    }
    go_RunReader(ctx);
    go_EncodedRun(ctx);
    knox_ctx_close(ctx);
    go_EvictOthers();
    std::printf(fails ? "FAILED: %d\n" : "go entries ok\n", fails);
    return fails ? 1 : 0;
}
