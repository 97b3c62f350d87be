// test_go_entries.cpp -- one block per exported function of integration/go/knoxgpu/knoxgpu.go, making the SAME C calls
// with the SAME argument marshalling the Go wrapper makes (there is no Go toolchain in the image; integration/
// check_go_calls.py checks the Go call sites against the header, this file checks that those call sequences do what
// the Go side expects, on the GPU).  Expected values come from a scalar loop over the same data -- the semantics of
// internal/cmp/generic.go and internal/reducer/reducer.go.
// Build: g++ tests/cpp/test_go_entries.cpp -Iinclude -Lknoxdb_amd -lknoxgpu
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
            knox_comm_destroy(ctx);
        }
    }
    knox_ctx_close(ctx);
    std::printf(fails ? "FAILED: %d\n" : "go entries ok\n", fails);
    return fails ? 1 : 0;
}
