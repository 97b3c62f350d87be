// knox_regexp_host.cpp -- bytes REGEXP on the host side: Go-regexp syntax -> DFA image (knox_regex.h), staged to the
// device; knox_match_regexp and the single-value host matcher.
#include "knox_internal.h"
#include "knox_regex.h"

// bytes REGEXP: compile on the host, stage the DFA image in ctx->bytes_val, one lane per row
int run_regexp(knox_ctx *ctx, const knox_col *col, const uint8_t *pat, uint64_t pat_len, const uint8_t *mask, uint8_t *bits,
               SkipPlan *skip) {
    if (col->dtype != KNOX_BYTES) return KNOX_E_INVALID;  // only string/byte blocks support regexp (operator/filter/match.go:76)
    if (pat_len && !pat) return KNOX_E_INVALID;
    if (pat_len > (1u << 20)) return KNOX_E_RANGE;
    knox_regex::Program pg;
    const int st = knox_regex::compile(pat, (size_t)pat_len, pg);
    if (st != knox_regex::OK) return KNOX_E_UNSUPPORTED;  // syntax outside the implemented subset / DFA too large: host regexp.Match
    const size_t tb = (size_t)pg.n_states * pg.n_classes * 2;
    // ASCII runes step through a direct [state][byte] table when it fits the LDS budget with the rest of the image
    const size_t db_want = (size_t)pg.n_states * 256;
    const size_t db = (db_want + tb + pg.n_states + 128 <= 40 * 1024) ? db_want : 0;
    const size_t image = db + tb + pg.n_states + 128;            // [direct] | table | flags | ascii (what the kernel stages in LDS)
    const size_t off_lo = (image + 15) & ~(size_t)15;
    const size_t off_cls = off_lo + pg.bound_lo.size() * 4;
    const size_t total = off_cls + pg.bound_class.size() + 16;
    std::vector<uint8_t> h(total, 0);
    if (db) {
        uint16_t *d16 = (uint16_t *)h.data();
        for (uint32_t st = 0; st < pg.n_states; st++)
            for (uint32_t b = 0; b < 128; b++) d16[st * 128 + b] = pg.table[(size_t)st * pg.n_classes + pg.ascii_class[b]] & 0x7fff;  // plain target: ACCEPT and dead states are absorbing
    }
    memcpy(h.data() + db, pg.table.data(), tb);
    memcpy(h.data() + db + tb, pg.flags.data(), pg.n_states);
    memcpy(h.data() + db + tb + pg.n_states, pg.ascii_class, 128);
    memcpy(h.data() + off_lo, pg.bound_lo.data(), pg.bound_lo.size() * 4);
    memcpy(h.data() + off_cls, pg.bound_class.data(), pg.bound_class.size());
    int rc = ensure(ctx, ctx->bytes_val, total);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, h.data(), total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host image not retained
    const uint8_t *d = (const uint8_t *)ctx->bytes_val.p;
    RegexProg dp;
    dp.direct = db ? (const uint16_t *)d : nullptr;
    dp.table = (const uint16_t *)(d + db);
    dp.flags = d + db + tb;
    dp.ascii_class = d + db + tb + pg.n_states;
    dp.bound_lo = (const uint32_t *)(d + off_lo);
    dp.bound_class = d + off_cls;
    dp.n_bounds = (uint32_t)pg.bound_lo.size();
    dp.n_classes = pg.n_classes;
    dp.n_states = pg.n_states;
    dp.lds_image_bytes = (uint32_t)image;
    hipError_t e = launch_regexp_match(ctx->stream, dp, (uint32_t)image, bytes_view(col, col->n_rows, skip), bits, mask);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

// bytesRegexpMatcher.MatchVector (match_bytes.go:587-609): bits.Set(i) for every (masked) row the pattern matches
KNOX_API int knox_match_regexp(knox_ctx *ctx, const knox_col *col, const uint8_t *pattern, uint64_t pattern_len, const knox_bits *mask,
                               knox_bits *bits, int64_t *count) {
    if (!ctx || !col || !bits || bits->size != col->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != col->n_rows) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, col); if (rc_) return rc_; }
    int rc = run_regexp(ctx, col, pattern, pattern_len, mask ? (const uint8_t *)mask->dev : nullptr, (uint8_t *)bits->dev);
    if (rc) return rc;
    if (count) return knox_bits_popcount(ctx, bits, count);
    return KNOX_OK;
}

// bytesRegexpMatcher.MatchValue (match_bytes.go:578-580) for one value: pure host code, the same compiled program
KNOX_API int knox_regexp_match_value(const uint8_t *pattern, uint64_t pattern_len, const uint8_t *value, uint64_t value_len) {
    if ((pattern_len && !pattern) || (value_len && !value)) return KNOX_E_INVALID;
    knox_regex::Program pg;
    if (knox_regex::compile(pattern, (size_t)pattern_len, pg) != knox_regex::OK) return KNOX_E_UNSUPPORTED;
    return knox_regex::match_host(pg, value, (size_t)value_len) ? 1 : 0;
}


KNOX_API const char *knox_regexp_unicode_version(void) { return knox_unicode::kVersion; }
