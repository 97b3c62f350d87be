// knox_containers.cpp -- ingest of the reference's on-disk block containers (internal/encode): headers parsed on the
// host, payloads expanded on the device; bitmap containers of bool columns; bit-packed columns kept packed and scanned
// as they are.
#include "knox_internal.h"

// pkg/num/varint.go:161-197 (SQLite4-style prefix varint); returns bytes consumed, 0 on short input
static int knox_uvarint(const uint8_t *b, size_t len, uint64_t *out) {
    if (len == 0) return 0;
    const uint8_t b0 = b[0];
    const int need = b0 <= 240 ? 1 : b0 <= 248 ? 2 : (int)b0 - 246;
    if ((size_t)need > len) return 0;
    if (b0 <= 240) { *out = b0; return 1; }
    if (b0 <= 248) { *out = 240 + ((uint64_t)(b0 - 241) << 8) + b[1]; return 2; }
    if (b0 == 249) { *out = 2288 + ((uint64_t)b[1] << 8) + b[2]; return 3; }
    if (b0 == 250) { *out = ((uint64_t)b[1] << 16) + ((uint64_t)b[2] << 8) + b[3]; return 4; }
    uint64_t x = ((uint64_t)b[1] << 24) + ((uint64_t)b[2] << 16) + ((uint64_t)b[3] << 8) + b[4];
    for (int i = 5; i < need; i++) x = (x << 8) + b[i];
    *out = x;
    return need;
}

// ---- on-disk containers -> device vectors ------------------------------------------------------------
// Container headers are parsed on the host (a few varints); payloads go to the device as they are and
// are expanded there.  Nested containers (dictionary, run-end) decode their children first.
//   [7]/[15] raw            internal/encode/int_raw.go:73-94, float_raw.go:73-89
//   [4]  bit-packed         int_bitpack.go:93-133
//   [1]/[10] constant       int_const.go:67-84, float_const.go:67-82
//   [2]  delta              int_delta.go:70-95
//   [5]/[12] dictionary     int_dict.go:76-103, float_dict.go:75-102     (T dict, uint16 codes)
//   [3]/[11] run-end        int_runend.go:84-112, float_runend.go:80-110 (T values, uint32 inclusive ends)
//   [6]  simple8            int_s8b.go:87-115, s8b/generic/decode.go:15-80
//   [14] ALP-RD             float_alprd.go:80-110, alp/rd.go:200-225 (uint16 left parts, right parts, shift)
//   [13] ALP                float_alp.go:109-165, alp/decoder.go:40-130 (values of int64 / int32, patches, positions)
static int container_measure(const uint8_t *b, size_t len, int w, bool isf, size_t *n_out, size_t *consumed, int depth = 0) {
    if (!b || len < 2 || depth > 4) return KNOX_E_INVALID;
    const uint8_t type = b[0];
    size_t o = 1;
    uint64_t v = 0;
    auto uv = [&](uint64_t *x) {
        const int k = knox_uvarint(b + o, len - o, x);
        if (k == 0) return false;
        o += (size_t)k;
        return true;
    };
    if (type == (isf ? 15 : 7)) {
        if (!uv(&v)) return KNOX_E_INVALID;
        const uint64_t n = isf ? v / (uint64_t)w : v;  // floats store the BYTE count (float_raw.go:75)
        if (o + n * (uint64_t)w > len) return KNOX_E_RANGE;
        *n_out = (size_t)n;
        *consumed = o + (size_t)n * (size_t)w;
        return KNOX_OK;
    }
    if (type == 4 && !isf) {
        uint64_t minv, l2, n;
        if (!uv(&minv) || !uv(&l2) || l2 > 64 || !uv(&n)) return KNOX_E_INVALID;
        const uint64_t sz = ((l2 * n + 63) & ~63ull) / 8;  // bitpack.EstimateSize
        if (o + sz > len) return KNOX_E_RANGE;
        *n_out = (size_t)n;
        *consumed = o + (size_t)sz;
        return KNOX_OK;
    }
    if ((type == 1 || type == 2) && !isf) {
        uint64_t base, delta = 0, n;
        if (!uv(&base) || (type == 2 && !uv(&delta)) || !uv(&n)) return KNOX_E_INVALID;
        *n_out = (size_t)n;
        *consumed = o;
        return KNOX_OK;
    }
    if (type == 6 && !isf) {  // int_s8b.go:87-93 [6][For][N][len(Packed)][Packed]
        uint64_t minv, n, plen;
        if (!uv(&minv) || !uv(&n) || !uv(&plen) || (plen & 7)) return KNOX_E_INVALID;
        if (o + plen > len) return KNOX_E_RANGE;
        if (n > plen / 8 * 128) return KNOX_E_INVALID;  // more values than the words can hold
        *n_out = (size_t)n;
        *consumed = o + (size_t)plen;
        return KNOX_OK;
    }
    if (type == 10 && isf) {
        if (o + (size_t)w >= len) return KNOX_E_RANGE;
        o += (size_t)w;
        uint64_t n;
        if (!uv(&n)) return KNOX_E_INVALID;
        *n_out = (size_t)n;
        *consumed = o;
        return KNOX_OK;
    }
    if (type == (isf ? 12 : 5) || type == (isf ? 11 : 3)) {
        // [type][child of T][child of uint16 codes / uint32 ends]
        const bool dict = type == (isf ? 12 : 5);
        size_t n1 = 0, c1 = 0, n2 = 0, c2 = 0;
        int rc = container_measure(b + o, len - o, w, isf, &n1, &c1, depth + 1);
        if (rc) return rc;
        o += c1;
        if (o >= len) return KNOX_E_RANGE;
        rc = container_measure(b + o, len - o, dict ? 2 : 4, false, &n2, &c2, depth + 1);
        if (rc) return rc;
        o += c2;
        if (dict) {
            if (n1 == 0 && n2 != 0) return KNOX_E_INVALID;
            *n_out = n2;
        } else {
            if (n1 != n2) return KNOX_E_INVALID;
            *n_out = 0;  // = ends[last] + 1, known after the ends are decoded: reported by container_rows()
            if (n2 == 0) return KNOX_E_INVALID;
        }
        *consumed = o;
        return KNOX_OK;
    }
    if (type == 13 && isf) {  // float_alp.go:109-120 [13][Exponent][Factor][flags][Values of E][Patches of T][Positions of uint32]
        uint64_t ex, fa;
        if (!uv(&ex) || !uv(&fa) || o >= len) return KNOX_E_INVALID;
        if (ex > (w == 8 ? 20u : 10u) || fa > (w == 8 ? 23u : 10u)) return KNOX_E_INVALID;  // alp/constants.go:47-130 table sizes
        const uint8_t flags = b[o++];
        size_t nv = 0, cv = 0;
        int rc = container_measure(b + o, len - o, w, false, &nv, &cv, depth + 1);  // E = int64 / int32
        if (rc) return rc;
        if (b[o] == 3) return KNOX_E_UNSUPPORTED;  // run-end values child: length known only after decode (not produced for ALP)
        o += cv;
        if (flags & 1) {  // FlagPatched
            size_t np = 0, cp = 0, nq = 0, cq = 0;
            if (o >= len) return KNOX_E_RANGE;
            rc = container_measure(b + o, len - o, w, true, &np, &cp, depth + 1);
            if (rc) return rc;
            o += cp;
            if (o >= len) return KNOX_E_RANGE;
            rc = container_measure(b + o, len - o, 4, false, &nq, &cq, depth + 1);
            if (rc) return rc;
            o += cq;
            if (np != nq || b[o - cq] == 3 || b[o - cq - cp] == 11) return np != nq ? KNOX_E_INVALID : KNOX_E_UNSUPPORTED;
        }
        *n_out = nv;
        *consumed = o;
        return KNOX_OK;
    }
    if (type == 14 && isf) {  // float_alprd.go:80-86 [14][Left of uint16][Right of uint64|uint32][Shift]
        size_t nl = 0, cl = 0, nr = 0, cr = 0;
        int rc = container_measure(b + o, len - o, 2, false, &nl, &cl, depth + 1);
        if (rc) return rc;
        if (b[o] == 3 || o + cl >= len) return b[o] == 3 ? KNOX_E_UNSUPPORTED : KNOX_E_RANGE;
        rc = container_measure(b + o + cl, len - o - cl, w, false, &nr, &cr, depth + 1);
        if (rc) return rc;
        if (b[o + cl] == 3) return KNOX_E_UNSUPPORTED;
        o += cl + cr;
        uint64_t shift;
        if (nl != nr || !uv(&shift) || shift >= (uint64_t)(8 * w)) return KNOX_E_INVALID;
        *n_out = nl;
        *consumed = o;
        return KNOX_OK;
    }
    return KNOX_E_UNSUPPORTED;  // 128/256-bit, strings, bitmaps: host decode path
}

static int container_decode(knox_ctx *ctx, const uint8_t *b, size_t len, int w, bool isf, void *dst, size_t n, int depth = 0);

// run-end containers carry their length in the last end: decode the ends child to learn it
static int runend_rows(knox_ctx *ctx, const uint8_t *b, size_t len, int w, bool isf, size_t *rows) {
    size_t n1 = 0, c1 = 0, n2 = 0, c2 = 0;
    int rc = container_measure(b + 1, len - 1, w, isf, &n1, &c1, 1);
    if (!rc) rc = container_measure(b + 1 + c1, len - 1 - c1, 4, false, &n2, &c2, 1);
    if (rc) return rc;
    void *d_ends = nullptr;
    if (dev_malloc(ctx, &d_ends, n2 * 4 + 64) != hipSuccess) return KNOX_E_NOMEM;
    rc = container_decode(ctx, b + 1 + c1, len - 1 - c1, 4, false, d_ends, n2, 1);
    uint32_t last = 0;
    if (!rc && (hipMemcpyAsync(&last, (const uint32_t *)d_ends + (n2 - 1), 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = KNOX_E_HIP;
    dev_free(ctx, d_ends);
    *rows = (size_t)last + 1;
    return rc;
}

static int container_rows(knox_ctx *ctx, const uint8_t *b, size_t len, int w, bool isf, size_t *rows) {
    size_t consumed = 0;
    int rc = container_measure(b, len, w, isf, rows, &consumed);
    if (rc) return rc;
    if (b[0] == (isf ? 11 : 3)) rc = runend_rows(ctx, b, len, w, isf, rows);
    return rc;
}

static int container_decode(knox_ctx *ctx, const uint8_t *b, size_t len, int w, bool isf, void *dst, size_t n, int depth) {
    if (n == 0) return KNOX_OK;
    const uint8_t type = b[0];
    size_t o = 1;
    uint64_t v = 0;
    auto uv = [&](uint64_t *x) { o += (size_t)knox_uvarint(b + o, len - o, x); };  // lengths were validated by container_measure
    hipError_t e = hipSuccess;
    if (type == (isf ? 15 : 7)) {
        uv(&v);
        return upload(ctx, dst, b + o, n * (size_t)w);
    }
    if (type == 4) {
        uint64_t minv, l2, nn;
        uv(&minv); uv(&l2); uv(&nn);
        const size_t sz = (size_t)(((l2 * nn + 63) & ~63ull) / 8);
        void *d_packed = nullptr;  // packed bits cross PCIe (log2/64 of the decoded size), the decode runs on the device
        if (dev_malloc(ctx, &d_packed, sz + 64) != hipSuccess) return KNOX_E_NOMEM;
        int rc = upload(ctx, d_packed, b + o, sz);
        if (!rc) {
            e = launch_bitpack_decode(ctx->stream, w, (const uint64_t *)d_packed, sz / 8, (uint32_t)l2, minv, nn, dst);
            if (e != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = KNOX_E_HIP;
        }
        dev_free(ctx, d_packed);
        return rc;
    }
    if (type == 6) {
        uint64_t minv, nn, plen;
        uv(&minv); uv(&nn); uv(&plen);
        const size_t n_words = (size_t)plen / 8;
        void *d_packed = nullptr, *d_scr = nullptr;
        if (dev_malloc(ctx, &d_packed, (size_t)plen + 64) != hipSuccess) return KNOX_E_NOMEM;
        if (dev_malloc(ctx, &d_scr, (n_words + 2) * 4 + 64) != hipSuccess) { dev_free(ctx, d_packed); return KNOX_E_NOMEM; }
        int rc = ensure(ctx, ctx->misc, 256);
        if (!rc) rc = upload(ctx, d_packed, b + o, (size_t)plen);
        if (!rc) {
            unsigned long long *d_tot = (unsigned long long *)((char *)ctx->misc.p + 128);
            e = launch_s8b_decode(ctx->stream, w, (const uint64_t *)d_packed, n_words, minv, nn, (uint32_t *)d_scr, d_tot, dst);
            unsigned long long tot = 0;
            if (e != hipSuccess || hipMemcpyAsync(&tot, d_tot, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                rc = KNOX_E_HIP;
            else if (tot < nn) rc = KNOX_E_INVALID;  // the words hold fewer values than the header claims
        }
        dev_free(ctx, d_packed);
        dev_free(ctx, d_scr);
        if (rc == KNOX_E_HIP) ctx->last_hip = (int)e;
        return rc;
    }
    if (type == 14) {
        size_t nl = 0, cl = 0, nr = 0, cr = 0;
        int rc = container_measure(b + o, len - o, 2, false, &nl, &cl, depth + 1);
        if (!rc) rc = container_measure(b + o + cl, len - o - cl, w, false, &nr, &cr, depth + 1);
        if (rc) return rc;
        uint64_t shift = 0;
        knox_uvarint(b + o + cl + cr, len - o - cl - cr, &shift);
        void *d_l = nullptr, *d_r = nullptr;
        if (dev_malloc(ctx, &d_l, nl * 2 + 64) != hipSuccess) return KNOX_E_NOMEM;
        if (dev_malloc(ctx, &d_r, nr * (size_t)w + 64) != hipSuccess) { dev_free(ctx, d_l); return KNOX_E_NOMEM; }
        rc = container_decode(ctx, b + o, len - o, 2, false, d_l, nl, depth + 1);
        if (!rc) rc = container_decode(ctx, b + o + cl, len - o - cl, w, false, d_r, nr, depth + 1);
        if (!rc) {
            e = launch_alprd_merge(ctx->stream, w, (const uint16_t *)d_l, d_r, (uint32_t)shift, nl, dst);
            if (e != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = KNOX_E_HIP;
        }
        dev_free(ctx, d_l);
        dev_free(ctx, d_r);
        if (rc == KNOX_E_HIP) ctx->last_hip = (int)e;
        return rc;
    }
    if (type == 13) {
        static const double kF10[24] = {1.0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                        1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22, 1e23};
        static const double kIF10[21] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001, 0.00000001, 0.000000001, 0.0000000001,
                                         0.00000000001, 0.000000000001, 0.0000000000001, 0.00000000000001, 0.000000000000001,
                                         0.0000000000000001, 0.00000000000000001, 0.000000000000000001, 0.0000000000000000001,
                                         0.00000000000000000001};
        static const float kF10f[11] = {1.0f, 10.0f, 100.0f, 1000.0f, 10000.0f, 100000.0f, 1000000.0f, 10000000.0f, 100000000.0f,
                                        1000000000.0f, 10000000000.0f};
        static const float kIF10f[11] = {1.0f, 0.1f, 0.01f, 0.001f, 0.0001f, 0.00001f, 0.000001f, 0.0000001f, 0.00000001f, 0.000000001f,
                                         0.0000000001f};
        uint64_t ex, fa;
        uv(&ex); uv(&fa);
        const uint8_t flags = b[o++];
        uint64_t f_bits = 0, e_bits = 0;
        if (w == 8) { memcpy(&f_bits, &kF10[fa], 8); memcpy(&e_bits, &kIF10[ex], 8); }
        else { memcpy(&f_bits, &kF10f[fa], 4); memcpy(&e_bits, &kIF10f[ex], 4); }
        size_t nv = 0, cv = 0, np = 0, cp = 0, nq = 0, cq = 0;
        int rc = container_measure(b + o, len - o, w, false, &nv, &cv, depth + 1);
        if (!rc && (flags & 1)) {
            rc = container_measure(b + o + cv, len - o - cv, w, true, &np, &cp, depth + 1);
            if (!rc) rc = container_measure(b + o + cv + cp, len - o - cv - cp, 4, false, &nq, &cq, depth + 1);
        }
        if (rc) return rc;
        void *d_enc = nullptr, *d_pat = nullptr, *d_pos = nullptr;
        if (dev_malloc(ctx, &d_enc, nv * (size_t)w + 64) != hipSuccess) return KNOX_E_NOMEM;
        rc = container_decode(ctx, b + o, len - o, w, false, d_enc, nv, depth + 1);
        if (!rc) {
            e = launch_alp_decode(ctx->stream, w, d_enc, f_bits, e_bits, nv, dst);
            if (e != hipSuccess) rc = KNOX_E_HIP;
        }
        if (!rc && np) {
            if (dev_malloc(ctx, &d_pat, np * (size_t)w + 64) != hipSuccess || dev_malloc(ctx, &d_pos, nq * 4 + 64) != hipSuccess) rc = KNOX_E_NOMEM;
            if (!rc) rc = container_decode(ctx, b + o + cv, len - o - cv, w, true, d_pat, np, depth + 1);
            if (!rc) rc = container_decode(ctx, b + o + cv + cp, len - o - cv - cp, 4, false, d_pos, nq, depth + 1);
            if (!rc) {
                e = launch_patch_scatter(ctx->stream, w, d_pat, (const uint32_t *)d_pos, np, nv, dst);
                if (e != hipSuccess) rc = KNOX_E_HIP;
            }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = KNOX_E_HIP;
        if (d_enc) dev_free(ctx, d_enc);
        if (d_pat) dev_free(ctx, d_pat);
        if (d_pos) dev_free(ctx, d_pos);
        if (rc == KNOX_E_HIP) ctx->last_hip = (int)e;
        return rc;
    }
    if (type == 1 || type == 2) {
        uint64_t base, delta = 0, nn;
        uv(&base);
        if (type == 2) uv(&delta);
        uv(&nn);
        e = launch_affine_fill(ctx->stream, w, base, delta, nn, dst);
    } else if (type == 10) {
        uint64_t bitsv = 0;
        memcpy(&bitsv, b + o, (size_t)w);
        e = launch_affine_fill(ctx->stream, w, bitsv, 0, n, dst);
    } else {
        const bool dict = type == (isf ? 12 : 5);
        size_t n1 = 0, c1 = 0, n2 = 0, c2 = 0;
        int rc = container_measure(b + o, len - o, w, isf, &n1, &c1, depth + 1);
        if (!rc) rc = container_measure(b + o + c1, len - o - c1, dict ? 2 : 4, false, &n2, &c2, depth + 1);
        if (rc) return rc;
        if (b[o] == (isf ? 11 : 3)) rc = runend_rows(ctx, b + o, len - o, w, isf, &n1);  // a run-end child reports its rows late
        if (rc) return rc;
        void *d1 = nullptr, *d2 = nullptr;
        if (dev_malloc(ctx, &d1, n1 * (size_t)w + 64) != hipSuccess) return KNOX_E_NOMEM;
        if (dev_malloc(ctx, &d2, n2 * (dict ? 2 : 4) + 64) != hipSuccess) { dev_free(ctx, d1); return KNOX_E_NOMEM; }
        rc = container_decode(ctx, b + o, len - o, w, isf, d1, n1, depth + 1);
        if (!rc) rc = container_decode(ctx, b + o + c1, len - o - c1, dict ? 2 : 4, false, d2, n2, depth + 1);
        if (!rc) {
            e = dict ? launch_dict_gather(ctx->stream, w, d1, n1, (const uint16_t *)d2, n, dst)
                     : launch_runend_expand(ctx->stream, w, d1, (const uint32_t *)d2, n2, n, dst);
            if (e != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = KNOX_E_HIP;
        }
        dev_free(ctx, d1);
        dev_free(ctx, d2);
        if (rc == KNOX_E_HIP) ctx->last_hip = (int)e;
        return rc;
    }
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

KNOX_API int knox_col_upload_raw_containers(knox_ctx *ctx, int dtype, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs,
                                            size_t pack_rows, knox_col **out) {
    if (!ctx || !out || (n_bufs && (!bufs || !lens))) return KNOX_E_INVALID;
    const int w = dtype_width(dtype);
    if (w == 0 || dtype == KNOX_BYTES) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    const bool isf = dtype_is_float(dtype);
    std::vector<size_t> rows(n_bufs);
    size_t total = 0;
    for (size_t i = 0; i < n_bufs; i++) {
        if (!bufs[i] || lens[i] < 2) return KNOX_E_INVALID;
        const int rc = container_rows(ctx, bufs[i], lens[i], w, isf, &rows[i]);
        if (rc) return rc;
        total += rows[i];
    }
    int rc = col_new(ctx, dtype, total, pack_rows, true, out);
    size_t row = 0;
    for (size_t i = 0; i < n_bufs && !rc; i++) {
        rc = container_decode(ctx, bufs[i], lens[i], w, isf, (char *)(*out)->dev + row * (size_t)w, rows[i]);
        row += rows[i];
    }
    if (rc) {
        if (*out) knox_col_free(*out);
        *out = nullptr;
    }
    return rc;
}

// Int128 / Int256 containers (internal/encode/int128.go:115-137, int256.go:121-150): [8][X0 of int64][X1 of uint64],
// [9][X0][X1][X2][X3] -- the stride layout the wide matchers take: one 8-byte limb column per limb, X0 first.
KNOX_API int knox_col_upload_wide_containers(knox_ctx *ctx, int n_limbs, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs,
                                             size_t pack_rows, knox_col **out_limbs) {
    if (!ctx || !out_limbs || (n_limbs != 2 && n_limbs != 4) || (n_bufs && (!bufs || !lens))) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    for (int l = 0; l < n_limbs; l++) out_limbs[l] = nullptr;
    // split every container into its limb children
    std::vector<std::vector<const uint8_t *>> cp(n_limbs, std::vector<const uint8_t *>(n_bufs));
    std::vector<std::vector<size_t>> cl(n_limbs, std::vector<size_t>(n_bufs));
    for (size_t i = 0; i < n_bufs; i++) {
        if (!bufs[i] || lens[i] < 2 || bufs[i][0] != (n_limbs == 2 ? 8 : 9)) return bufs[i] && lens[i] >= 2 ? KNOX_E_UNSUPPORTED : KNOX_E_INVALID;
        size_t o = 1;
        for (int l = 0; l < n_limbs; l++) {
            size_t n = 0, consumed = 0;
            if (o >= lens[i]) return KNOX_E_RANGE;
            const int rc = container_measure(bufs[i] + o, lens[i] - o, 8, false, &n, &consumed);
            if (rc) return rc;
            cp[l][i] = bufs[i] + o;
            cl[l][i] = consumed;
            o += consumed;
        }
    }
    for (int l = 0; l < n_limbs; l++) {
        const int rc = knox_col_upload_raw_containers(ctx, l == 0 ? KNOX_INT64 : KNOX_UINT64, cp[l].data(), cl[l].data(), n_bufs, pack_rows, &out_limbs[l]);
        if (rc || (l && out_limbs[l]->n_rows != out_limbs[0]->n_rows)) {
            for (int k = 0; k <= l; k++) {
                if (out_limbs[k]) knox_col_free(out_limbs[k]);
                out_limbs[k] = nullptr;
            }
            return rc ? rc : KNOX_E_INVALID;
        }
    }
    return KNOX_OK;
}

// String containers -> one bytes (StringPool) column: payloads are uploaded as they are, the row
// pointers are built on the device.
//   [16] constant    string_const.go:51-70      [n][len][val]
//   [17] fixed       string_fixed.go:59-82      [n][sz][n*sz bytes]
//   [18] compact     string_compact.go:64-97    [ofs: uint32 container][len: uint32 container][buflen][buf]
//   [19] dictionary  string_dict.go:70-108      [ofs][len][code: uint32 container][dictlen][dict]
KNOX_API int knox_col_upload_string_containers(knox_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs, size_t pack_rows,
                                               knox_col **out) {
    if (!ctx || !out || (n_bufs && (!bufs || !lens))) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    struct Part {
        uint8_t type;
        size_t rows, n_entries;
        const uint8_t *payload;
        size_t payload_len;
        uint64_t sz;                      // const: value length, fixed: element size
        const uint8_t *c_ofs, *c_len, *c_code;
        size_t l_ofs, l_len, l_code;
    };
    std::vector<Part> parts(n_bufs);
    size_t total_rows = 0, total_pool = 0;
    for (size_t i = 0; i < n_bufs; i++) {
        const uint8_t *b = bufs[i];
        const size_t len = lens[i];
        if (!b || len < 2) return KNOX_E_INVALID;
        Part &p = parts[i];
        memset(&p, 0, sizeof p);
        p.type = b[0];
        size_t o = 1;
        auto uv = [&](uint64_t *x) {
            const int k = knox_uvarint(b + o, len - o, x);
            if (k == 0) return false;
            o += (size_t)k;
            return true;
        };
        uint64_t n = 0, v = 0;
        if (p.type == 16 || p.type == 17) {
            if (!uv(&n) || !uv(&v)) return KNOX_E_INVALID;
            const uint64_t bytes = p.type == 16 ? v : n * v;
            if (v > 0xffffffffull || o + bytes > len) return KNOX_E_RANGE;
            p.rows = (size_t)n;
            p.sz = v;
            p.payload = b + o;
            p.payload_len = (size_t)bytes;
        } else if (p.type == 18 || p.type == 19) {
            size_t consumed = 0, rows = 0;
            auto child = [&](const uint8_t **cp, size_t *cl, size_t *rows_out) {
                if (o >= len) return (int)KNOX_E_RANGE;
                int rc = container_measure(b + o, len - o, 4, false, rows_out, &consumed);
                if (!rc && b[o] == 3) rc = runend_rows(ctx, b + o, len - o, 4, false, rows_out);
                *cp = b + o;
                *cl = consumed;
                o += consumed;
                return rc;
            };
            size_t n_ofs = 0, n_len = 0;
            int rc = child(&p.c_ofs, &p.l_ofs, &n_ofs);
            if (!rc) rc = child(&p.c_len, &p.l_len, &n_len);
            if (rc) return rc;
            if (n_ofs != n_len) return KNOX_E_INVALID;
            p.n_entries = n_ofs;
            rows = n_ofs;
            if (p.type == 19) {
                rc = child(&p.c_code, &p.l_code, &rows);
                if (rc) return rc;
                if (rows && p.n_entries == 0) return KNOX_E_INVALID;
            }
            if (!uv(&v) || o + v > len) return KNOX_E_RANGE;
            p.rows = rows;
            p.payload = b + o;
            p.payload_len = (size_t)v;
        } else {
            return KNOX_E_UNSUPPORTED;
        }
        total_rows += p.rows;
        total_pool += p.payload_len;
    }
    // One StringPool per pack, as the reference holds it, whenever the containers line up with the packs (every
    // container but the last holds exactly pack_rows rows): offsets stay relative to the pack's own payload and the
    // column as a whole is not limited to 4 GiB.  Ragged containers share one pool (32-bit offsets: <= 4 GiB).
    bool per_pack = pack_rows != 0 && (pack_rows & 63) == 0 && n_bufs > 0;
    for (size_t i = 0; i + 1 < n_bufs && per_pack; i++) per_pack = parts[i].rows == pack_rows;
    if (per_pack && (parts[n_bufs - 1].rows == 0 || parts[n_bufs - 1].rows > pack_rows)) per_pack = false;
    if (!per_pack && total_pool > 0xffffffffull) return KNOX_E_RANGE;  // StringPool offsets are 32 bit (pkg/stringx/pool.go:45-58)
    std::vector<uint64_t> bases(n_bufs + 1, 0);
    for (size_t i = 0; i < n_bufs; i++) {
        if (parts[i].payload_len > 0xffffffffull) return KNOX_E_RANGE;
        bases[i + 1] = bases[i] + parts[i].payload_len;
        if (per_pack) bases[i + 1] = (bases[i + 1] + 63) & ~(uint64_t)63;
    }
    total_pool = (size_t)bases[n_bufs];
    int rc = col_new(ctx, KNOX_BYTES, total_rows, pack_rows, true, out);
    if (rc) return rc;
    knox_col *c = *out;
    auto fail = [&](int code) {
        knox_col_free(c);
        *out = nullptr;
        return code;
    };
    if (dev_malloc(ctx, (void **)&c->buf, total_pool + 256, ARENA_READ) != hipSuccess) return fail(KNOX_E_NOMEM);
    c->buf_len = total_pool;
    if (per_pack) {
        c->h_pool_base = (uint64_t *)malloc((n_bufs + 1) * 8);
        if (!c->h_pool_base || dev_malloc(ctx, (void **)&c->pool_base, (n_bufs + 1) * 8) != hipSuccess) return fail(KNOX_E_NOMEM);
        memcpy(c->h_pool_base, bases.data(), (n_bufs + 1) * 8);
        rc = upload(ctx, c->pool_base, bases.data(), (n_bufs + 1) * 8);
        if (rc) return fail(rc);
        c->n_packs = c->packs_cap = n_bufs;
        c->rows_cap = total_rows;
        c->buf_cap = total_pool;
    }
    rc = ensure(ctx, ctx->misc, 256);
    if (rc) return fail(rc);
    uint32_t *d_flag = (uint32_t *)((char *)ctx->misc.p + 32);
    if (hipMemsetAsync(d_flag, 0, 4, ctx->stream) != hipSuccess) return fail(KNOX_E_HIP);
    size_t row = 0;
    for (size_t i = 0; i < n_bufs; i++) {
        const Part &p = parts[i];
        uint64_t *dst = (uint64_t *)c->dev + row;
        const uint64_t pool = bases[i];               // where this container's payload sits in the column's buffer
        const uint64_t rel = per_pack ? 0 : pool;     // what the row offsets are relative to
        rc = upload(ctx, c->buf + pool, p.payload, p.payload_len);
        if (rc) return fail(rc);
        hipError_t e = hipSuccess;
        if (p.type == 16) {
            e = launch_affine_fill(ctx->stream, 8, (rel << 32) | p.sz, 0, p.rows, dst);
        } else if (p.type == 17) {
            e = launch_affine_fill(ctx->stream, 8, (rel << 32) | p.sz, p.sz << 32, p.rows, dst);
        } else {
            void *d_ofs = nullptr, *d_len = nullptr, *d_code = nullptr;
            if (dev_malloc(ctx, &d_ofs, p.n_entries * 4 + 64) != hipSuccess || dev_malloc(ctx, &d_len, p.n_entries * 4 + 64) != hipSuccess ||
                (p.type == 19 && dev_malloc(ctx, &d_code, p.rows * 4 + 64) != hipSuccess))
                rc = KNOX_E_NOMEM;
            if (!rc) rc = container_decode(ctx, p.c_ofs, p.l_ofs, 4, false, d_ofs, p.n_entries);
            if (!rc) rc = container_decode(ctx, p.c_len, p.l_len, 4, false, d_len, p.n_entries);
            if (!rc && p.type == 19) rc = container_decode(ctx, p.c_code, p.l_code, 4, false, d_code, p.rows);
            if (!rc) {
                e = launch_str_ptr(ctx->stream, (const uint32_t *)d_ofs, (const uint32_t *)d_len, p.n_entries, (const uint32_t *)d_code, p.rows,
                                   rel, dst);
                if (e != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = KNOX_E_HIP;
            }
            if (d_ofs) dev_free(ctx, d_ofs);
            if (d_len) dev_free(ctx, d_len);
            if (d_code) dev_free(ctx, d_code);
            if (rc) {
                if (rc == KNOX_E_HIP) ctx->last_hip = (int)e;
                return fail(rc);
            }
        }
        // every (off, len) must lie inside its pool: the match kernels never read out of bounds
        if (e == hipSuccess && per_pack) e = launch_ptr_check(ctx->stream, dst, p.rows, p.payload_len, d_flag);
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            return fail(KNOX_E_HIP);
        }
        row += p.rows;
    }
    uint32_t bad = 0;
    if ((!per_pack && launch_ptr_check(ctx->stream, (const uint64_t *)c->dev, total_rows, total_pool, d_flag) != hipSuccess) ||
        hipMemcpyAsync(&bad, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(KNOX_E_HIP);
    if (bad) return fail(KNOX_E_RANGE);
    return KNOX_OK;
}

// Bitmap containers of bool columns (internal/encode/bitmap.go:103-135, :240-262): [20] zero, [21] one,
// [22] dense = the Bitset bytes; [23] sparse is a serialized roaring bitmap (internal/xroar): host path.
// TBitmapSparse [23]: the payload is a serialized xroar bitmap (internal/xroar/bitmap.go:98-113, keys.go:3-26,
// container.go:33-78,356-377 -- the sroar layout): uint16 words; the first u64[0] words are the key node: u64[1] = number
// of keys, then (key, offset) pairs, key = the value's upper 48 bits, offset = the container's position in words.  A
// container is [size][type][cardinality lo][cardinality hi] + data: type 0 = `cardinality` sorted low-16-bit values,
// type 1 = 4096 words of one bit per value, MSB first (bitmapMask[i] = 1 << (15 - i)).  Decoded on the host into the
// dense LSB-first bytes of the bitset (the container is chosen for < 1 bit in 16 set, bitmap.go:253-257: the dense form
// is at most twice its size); every offset and value is bounds-checked.
static int roaring_to_dense(const uint8_t *buf, size_t len, size_t n_bits, std::vector<uint8_t> &dense) {
    dense.assign(bits_bytes(n_bits), 0);
    if (len < 8) return KNOX_OK;  // NewFromBytes: "len(src) < 8 -> New()": empty
    if (len & 1) return KNOX_E_INVALID;
    const size_t n_words = len / 2;
    auto w16 = [&](size_t i) { uint16_t v; memcpy(&v, buf + 2 * i, 2); return v; };
    auto w64 = [&](size_t i) { uint64_t v; memcpy(&v, buf + 8 * i, 8); return v; };
    const uint64_t node_words = w64(0);
    if (node_words < 8 || node_words > n_words || (node_words & 3)) return KNOX_E_RANGE;
    const uint64_t n_keys = w64(1);
    if (n_keys > (node_words / 4 - 2) / 2) return KNOX_E_RANGE;
    for (uint64_t i = 0; i < n_keys; i++) {
        const uint64_t key = w64(2 + 2 * i), off = w64(3 + 2 * i);
        if (key & 0xffffull) return KNOX_E_INVALID;
        if (off + 4 > n_words) return KNOX_E_RANGE;
        const size_t size = w16(off), type = w16(off + 1), card = (size_t)w16(off + 2) + (size_t)w16(off + 3);
        if (size < 4 || off + size > n_words) return KNOX_E_RANGE;
        if (card == 0) continue;  // the container New() makes for key 0
        if (key >= n_bits) return KNOX_E_RANGE;
        if (type == 0) {
            if (4 + card > size) return KNOX_E_RANGE;
            for (size_t j = 0; j < card; j++) {
                const uint64_t v = key | w16(off + 4 + j);
                if (v >= n_bits) return KNOX_E_RANGE;
                dense[v >> 3] |= (uint8_t)(1u << (v & 7));
            }
        } else if (type == 1) {
            if (size < 4 + 4096) return KNOX_E_RANGE;
            for (size_t wi = 0; wi < 4096; wi++) {
                uint16_t x = w16(off + 4 + wi);
                while (x) {
                    const int lead = __builtin_clz((unsigned)x) - 16;  // MSB first: bit 15 is value 16*wi + 0
                    const uint64_t v = key | (uint64_t)(wi * 16 + (size_t)lead);
                    if (v >= n_bits) return KNOX_E_RANGE;
                    dense[v >> 3] |= (uint8_t)(1u << (v & 7));
                    x = (uint16_t)(x & ~(0x8000u >> lead));
                }
            }
        } else {
            return KNOX_E_INVALID;
        }
    }
    return KNOX_OK;
}

KNOX_API int knox_bits_upload_bitmap_container(knox_ctx *ctx, knox_bits *b, const uint8_t *buf, size_t len) {
    if (!ctx || !b || !buf || len < 3) return KNOX_E_INVALID;
    const uint8_t type = buf[0];
    if (type < 20 || type > 23) return KNOX_E_INVALID;
    size_t o = 1;
    uint64_t n = 0, sz = 0;
    int k = knox_uvarint(buf + o, len - o, &n);
    if (k == 0) return KNOX_E_INVALID;
    o += (size_t)k;
    k = knox_uvarint(buf + o, len - o, &sz);
    if (k == 0) return KNOX_E_INVALID;
    o += (size_t)k;
    if (n != b->size) return KNOX_E_INVALID;
    if (o + sz > len) return KNOX_E_RANGE;
    if (type == 20) return knox_bits_zero(ctx, b);
    if (type == 21) return knox_bits_one(ctx, b);
    if (type == 23) {
        std::vector<uint8_t> dense;
        int rc = roaring_to_dense(buf + o, (size_t)sz, b->size, dense);
        if (rc) return rc;
        return b->size ? knox_bits_upload(ctx, b, dense.data()) : KNOX_OK;
    }
    if (sz < bits_bytes(b->size)) return KNOX_E_RANGE;
    return knox_bits_upload(ctx, b, buf + o);
}

// ---- resident bit-packed columns: scanned as they are --------------------------------------------------
struct knox_packed {
    knox_ctx *ctx;
    int dtype;
    size_t n_rows, pack_rows, n_packs;
    uint64_t *d_words;
    PackedPack *d_packs;
    size_t words_bytes;
    // appendable packed columns (knox_packed_create / _append_async): capacities and the last copy's event
    size_t packs_cap = 0, words_cap = 0;
    hipEvent_t ready_ev = nullptr;
    bool ready_pending = false;
    uint32_t max_log2 = 0;     // widest field of any pack
    bool aligned16 = true;     // every pack's first word is 16-byte aligned (the LDS-staged scan loads 16 bytes per lane)
};

// [4][uvarint For][uvarint log2][uvarint N][packed bits] (int_bitpack.go:93-133): header fields + where the payload lies
static int parse_bitpack_header(const uint8_t *buf, size_t len, uint64_t *minv, uint64_t *l2, uint64_t *n, size_t *payload_off, size_t *payload_len) {
    if (!buf || len < 2) return KNOX_E_INVALID;
    if (buf[0] != 4) return KNOX_E_UNSUPPORTED;  // only bit-packed containers stay packed
    size_t o = 1;
    int k = knox_uvarint(buf + o, len - o, minv);
    if (k == 0) return KNOX_E_INVALID;
    o += (size_t)k;
    k = knox_uvarint(buf + o, len - o, l2);
    if (k == 0 || *l2 > 64) return KNOX_E_INVALID;
    o += (size_t)k;
    k = knox_uvarint(buf + o, len - o, n);
    if (k == 0) return KNOX_E_INVALID;
    o += (size_t)k;
    const uint64_t sz = ((*l2 * *n + 63) & ~63ull) / 8;
    if (o + sz > len) return KNOX_E_RANGE;
    *payload_off = o;
    *payload_len = (size_t)sz;
    return KNOX_OK;
}

// ---- packed columns filled on the copy stream: a cold scan of bit-packed data crosses PCIe at log2 / 64 of the bytes ------
KNOX_API int knox_packed_create(knox_ctx *ctx, int dtype, size_t pack_rows, size_t max_packs, size_t max_payload_bytes, knox_packed **out) {
    if (!ctx || !out || dtype_width(dtype) == 0 || dtype_is_float(dtype) || dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (pack_rows == 0 || (pack_rows & 4095) || max_packs == 0 || max_packs > 0xffffffffull || max_payload_bytes == 0) return KNOX_E_INVALID;
    *out = nullptr;
    hipSetDevice(ctx->device);
    knox_packed *p = new (std::nothrow) knox_packed();
    if (!p) return KNOX_E_NOMEM;
    p->ctx = ctx;
    p->dtype = dtype;
    p->n_rows = p->n_packs = p->words_bytes = 0;
    p->pack_rows = pack_rows;
    p->d_words = nullptr;
    p->d_packs = nullptr;
    p->packs_cap = max_packs;
    p->words_cap = (max_payload_bytes + 7) / 8;
    if (dev_malloc(ctx, (void **)&p->d_words, p->words_cap * 8 + 64) != hipSuccess ||
        dev_malloc(ctx, (void **)&p->d_packs, max_packs * sizeof(PackedPack) + 64) != hipSuccess) {
        if (p->d_words) dev_free(ctx, p->d_words);
        delete p;
        return KNOX_E_NOMEM;
    }
    // the kernel reads one word past a field's word unconditionally: the tail is defined once
    if (hipMemsetAsync((char *)p->d_words + p->words_cap * 8, 0, 64, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        knox_packed_free(p);
        return KNOX_E_HIP;
    }
    *out = p;
    return KNOX_OK;
}

KNOX_API int knox_packed_append_async(knox_ctx *ctx, knox_packed *p, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs) {
    if (!ctx || !p || p->ctx != ctx || p->packs_cap == 0 || !bufs || !lens) return KNOX_E_INVALID;
    if (n_bufs == 0) return KNOX_OK;
    if (n_bufs > p->packs_cap - p->n_packs) return KNOX_E_RANGE;
    if (p->n_packs && p->n_rows != p->n_packs * p->pack_rows) return KNOX_E_INVALID;  // a short pack can only be the last one
    hipSetDevice(ctx->device);
    int rc = copy_stream_init(ctx);
    if (rc) return rc;
    std::vector<PackedPack> packs(n_bufs);
    std::vector<size_t> off(n_bufs), plen(n_bufs);
    size_t words = p->words_bytes / 8, rows = 0;
    for (size_t i = 0; i < n_bufs; i++) {
        uint64_t minv = 0, l2 = 0, n = 0;
        rc = parse_bitpack_header(bufs[i], lens[i], &minv, &l2, &n, &off[i], &plen[i]);
        if (rc) return rc;
        if (n > p->pack_rows || (i + 1 < n_bufs && n != p->pack_rows) || n == 0) return KNOX_E_INVALID;
        if (words + plen[i] / 8 > p->words_cap) return KNOX_E_RANGE;
        packs[i] = {words, minv, (uint32_t)n, (uint32_t)l2};
        p->max_log2 = std::max(p->max_log2, (uint32_t)l2);
        p->aligned16 = p->aligned16 && (words & 1) == 0;
        words += plen[i] / 8;
        rows += (size_t)n;
    }
    // payloads (page-locked sources in place, others through the copy stream's ring), then the pack descriptors
    for (size_t i = 0; i < n_bufs && !rc; i++) rc = upload_async(ctx, p->d_words + packs[i].word_off, bufs[i] + off[i], plen[i]);
    if (!rc) rc = upload_async(ctx, p->d_packs + p->n_packs, packs.data(), n_bufs * sizeof(PackedPack));
    if (rc) return rc;
    if (!p->ready_ev) HIPCHK(ctx, hipEventCreateWithFlags(&p->ready_ev, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(p->ready_ev, ctx->copy_stream));
    p->ready_pending = true;
    p->n_packs += n_bufs;
    p->n_rows += rows;
    p->words_bytes = words * 8;
    return KNOX_OK;
}

KNOX_API int knox_packed_reset(knox_ctx *ctx, knox_packed *p) {
    if (!ctx || !p || p->ctx != ctx || p->packs_cap == 0) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = copy_stream_init(ctx);
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->fence_ev, ctx->stream));   // the scans enqueued so far may still read the old words
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->fence_ev, 0));
    p->n_packs = p->n_rows = p->words_bytes = 0;
    p->max_log2 = 0;
    p->aligned16 = true;
    return KNOX_OK;
}

KNOX_API int knox_packed_wait(knox_ctx *ctx, knox_packed *p) {
    if (!ctx || !p) return KNOX_E_INVALID;
    if (p->ready_pending) {
        HIPCHK(ctx, hipEventSynchronize(p->ready_ev));
        p->ready_pending = false;
    }
    return KNOX_OK;
}

KNOX_API int knox_packed_upload(knox_ctx *ctx, int dtype, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs, size_t pack_rows,
                                knox_packed **out) {
    if (!ctx || !out || !n_bufs || !bufs || !lens || dtype_width(dtype) == 0 || dtype_is_float(dtype) || dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (pack_rows == 0 || (pack_rows & 4095) || n_bufs > 0xffffffffull) return KNOX_E_INVALID;  // whole 4096-value wave chunks
    *out = nullptr;
    hipSetDevice(ctx->device);
    std::vector<PackedPack> packs(n_bufs);
    std::vector<const uint8_t *> payload(n_bufs);
    std::vector<size_t> payload_len(n_bufs);
    size_t words = 0, rows = 0;
    uint32_t max_l2 = 0;
    bool al16 = true;
    for (size_t i = 0; i < n_bufs; i++) {
        uint64_t minv = 0, l2 = 0, n = 0;
        size_t o = 0, sz = 0;
        const int prc = parse_bitpack_header(bufs[i], lens[i], &minv, &l2, &n, &o, &sz);
        if (prc) return prc;
        if (n > pack_rows || (i + 1 < n_bufs && n != pack_rows) || n == 0) return KNOX_E_INVALID;  // only the last pack may be short
        packs[i] = {words, minv, (uint32_t)n, (uint32_t)l2};
        max_l2 = std::max(max_l2, (uint32_t)l2);
        al16 = al16 && (words & 1) == 0;
        payload[i] = bufs[i] + o;
        payload_len[i] = (size_t)sz;
        words += (size_t)sz / 8;
        rows += (size_t)n;
    }
    knox_packed *p = new (std::nothrow) knox_packed();
    if (!p) return KNOX_E_NOMEM;
    p->ctx = ctx;
    p->dtype = dtype;
    p->n_rows = rows;
    p->pack_rows = pack_rows;
    p->n_packs = n_bufs;
    p->d_words = nullptr;
    p->d_packs = nullptr;
    p->words_bytes = words * 8;
    p->max_log2 = max_l2;
    p->aligned16 = al16;
    // + 64 bytes: the kernel reads one word past a field's word unconditionally
    if (dev_malloc(ctx, (void **)&p->d_words, words * 8 + 64) != hipSuccess || dev_malloc(ctx, (void **)&p->d_packs, n_bufs * sizeof(PackedPack) + 64) != hipSuccess) {
        if (p->d_words) dev_free(p->ctx, p->d_words);
        delete p;
        return KNOX_E_NOMEM;
    }
    int rc = KNOX_OK;
    if (hipMemsetAsync((char *)p->d_words + words * 8, 0, 64, ctx->stream) != hipSuccess) rc = KNOX_E_HIP;
    for (size_t i = 0; i < n_bufs && !rc; i++) rc = upload(ctx, p->d_words + packs[i].word_off, payload[i], payload_len[i]);
    if (!rc) rc = upload(ctx, p->d_packs, packs.data(), n_bufs * sizeof(PackedPack));
    if (rc) {
        dev_free(p->ctx, p->d_words);
        dev_free(p->ctx, p->d_packs);
        delete p;
        return rc;
    }
    *out = p;
    return KNOX_OK;
}

KNOX_API int knox_packed_free(knox_packed *p) {
    if (!p) return KNOX_E_INVALID;
    hipSetDevice(p->ctx->device);
    if (p->ready_pending) hipEventSynchronize(p->ready_ev);
    if (p->ready_ev) hipEventDestroy(p->ready_ev);
    hipStreamSynchronize(p->ctx->stream);
    if (p->d_words) dev_free(p->ctx, p->d_words);
    if (p->d_packs) dev_free(p->ctx, p->d_packs);
    delete p;
    return KNOX_OK;
}

KNOX_API size_t knox_packed_rows(const knox_packed *p) { return p ? p->n_rows : 0; }
KNOX_API size_t knox_packed_bytes(const knox_packed *p) { return p ? p->words_bytes : 0; }

// one leaf (any numeric mode) + count [+ wrapping sum] [+ bitset] [+ per-pack counts] straight on the packed words
KNOX_API int knox_packed_scan(knox_ctx *ctx, const knox_packed *p, int op, uint64_t a, uint64_t b, int want_sum, knox_bits *bits_out,
                              int64_t *pack_counts, int64_t *count, knox_agg_state *sum_state) {
    if (!ctx || !p || p->ctx != ctx || (want_sum && !sum_state)) return KNOX_E_INVALID;
    if (bits_out && bits_out->size != p->n_rows) return KNOX_E_INVALID;
    if (pack_counts && !bits_out) return KNOX_E_INVALID;  // per-pack counts come from the bitset
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    if (p->ready_pending) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, p->ready_ev, 0));  // behind the column's asynchronous appends
    LeafHost lh;
    int rc = make_leaf(p->dtype, op, a, b, lh);
    if (rc) return rc;
    const int grid_max = std::max(1, ctx->n_cus * 8);
    rc = ensure(ctx, ctx->partials, sizeof(AggPartial) * (size_t)(grid_max + 1) * 16);
    if (!rc) rc = ensure(ctx, ctx->states, sizeof(AggStateDev) * 16 + 8 * 16);
    if (rc) return rc;
    PackedScanArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.words = p->d_words;
    pa.packs = p->d_packs;
    pa.n_packs = (uint32_t)p->n_packs;
    pa.groups_per_pack = (uint32_t)(p->pack_rows / 4096);
    pa.pack_rows = p->pack_rows;
    pa.n_rows = p->n_rows;
    pa.lo = lh.lo;
    pa.diff = lh.diff;
    pa.key_xor = 0;  // integer leaves compare raw bits (the reference's unsigned-difference form)
    pa.wmask = wmask(dtype_width(p->dtype));
    pa.negate = lh.neg ? 1u : 0u;
    pa.want_sum = want_sum ? 1u : 0u;
    pa.bits = bits_out ? bits_out->dev : nullptr;
    pa.partials = (AggPartial *)ctx->partials.p;
    const uint64_t n_chunks = (uint64_t)p->n_packs * pa.groups_per_pack;
    const int blocks_per_cu = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 8;  // latency bound: 32 waves/CU (tools/bench_packed.py sweep)
    const int grid = (int)std::min<uint64_t>((n_chunks + kWavesPerBlock - 1) / kWavesPerBlock, (uint64_t)std::min(grid_max, ctx->n_cus * blocks_per_cu));
    prof_mark(ctx, 0);
    // fields of at most 32 bits: chunks staged in LDS (knox_decode.hip); KNOX_PACKED_FORM=global pins the general form (measurements, tests)
    const char *pf = getenv("KNOX_PACKED_FORM");
    const bool staged = p->max_log2 <= 32 && p->aligned16 && !(pf && !strcmp(pf, "global"));
    hipError_t e = staged ? launch_packed_scan_lds(ctx->stream, grid, pa, p->max_log2) : launch_packed_scan(ctx->stream, grid, pa);
    prof_mark(ctx, 1);
    if (e == hipSuccess)
        e = launch_finalize(ctx->stream, dtype_width(p->dtype), pa.partials, grid, nullptr, 0, p->dtype, (AggStateDev *)ctx->states.p, (int64_t *)((AggStateDev *)ctx->states.p + 16));
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    if (pack_counts) {
        rc = ensure(ctx, ctx->pack_counts, p->n_packs * 8 + 64);
        if (rc) return rc;
        e = launch_popcount_segments(ctx->stream, bits_out->dev, p->n_rows, p->pack_rows, (int64_t *)ctx->pack_counts.p, p->n_packs);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        rc = download(ctx, pack_counts, ctx->pack_counts.p, p->n_packs * 8);
        if (rc) return rc;
    }
    AggStateDev st;
    HIPCHK(ctx, hipMemcpyAsync(&st, ctx->states.p, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (count) *count = st.count;
    if (sum_state) memcpy(sum_state, &st, sizeof st);
    return KNOX_OK;
}

// ------------------------------------------------------------------------------------------------
// Integer columns kept ENCODED in HBM and scanned as they are (round 4).  The reference matches on containers without
// materializing them (internal/encode/int_dict.go:181-380: the compare runs on the uint16 codes; int_runend.go:224-330: on the run
// values, hit runs become row ranges; int_const.go / int_delta.go: from the header alone); round 3 expanded everything but
// bit-packed containers to full-width vectors at upload.  Here every pack keeps the form it is cheapest to scan in, pack by pack
// (the reference's encoder chooses the scheme per block, so a run of packs is usually MIXED):
//   dictionary [5]   2 B of codes per row + the dictionary        run-end [3]   (W + 4) B per run
//   constant [1], delta [2]   nothing but the header             everything else (raw, bit-packed, simple8, nested containers): decoded on
//   the device at upload into W B per row and compared there (kind 7).
// The children of dictionary / run-end containers are decoded on the device too (any integer container the reference nests there but
// another dictionary / run-end); what crosses PCIe is always the container's own bytes.
// ------------------------------------------------------------------------------------------------
struct knox_enc {
    knox_ctx *ctx;
    int dtype;
    size_t n_rows, pack_rows, n_packs;
    void *d_vals = nullptr;       // dictionary entries / run values / raw values of all packs
    uint16_t *d_codes = nullptr;  // pack_rows codes per dictionary pack
    uint32_t *d_ends = nullptr;   // indexed like d_vals (only run-end packs use their slots)
    EncPack *d_packs = nullptr;
    uint32_t *d_lists = nullptr;  // pack numbers by scanned kind: dictionary | run-end | constant + delta | raw
    size_t n_dict = 0, n_run = 0, n_aff = 0, n_raw = 0;
    size_t resident_bytes = 0, container_bytes = 0;
};

KNOX_API int knox_enc_free(knox_enc *e) {
    if (!e) return KNOX_E_INVALID;
    hipSetDevice(e->ctx->device);
    hipStreamSynchronize(e->ctx->stream);
    if (e->d_vals) dev_free(e->ctx, e->d_vals);
    if (e->d_codes) dev_free(e->ctx, e->d_codes);
    if (e->d_ends) dev_free(e->ctx, e->d_ends);
    if (e->d_packs) dev_free(e->ctx, e->d_packs);
    if (e->d_lists) dev_free(e->ctx, e->d_lists);
    delete e;
    return KNOX_OK;
}
KNOX_API size_t knox_enc_rows(const knox_enc *e) { return e ? e->n_rows : 0; }
KNOX_API size_t knox_enc_bytes(const knox_enc *e) { return e ? e->resident_bytes : 0; }
KNOX_API size_t knox_enc_container_bytes(const knox_enc *e) { return e ? e->container_bytes : 0; }
KNOX_API int knox_enc_kinds(const knox_enc *e, size_t *n_dictionary, size_t *n_runend, size_t *n_const_delta, size_t *n_raw) {
    if (!e) return KNOX_E_INVALID;
    if (n_dictionary) *n_dictionary = e->n_dict;
    if (n_runend) *n_runend = e->n_run;
    if (n_const_delta) *n_const_delta = e->n_aff;
    if (n_raw) *n_raw = e->n_raw;
    return KNOX_OK;
}

KNOX_API int knox_enc_upload(knox_ctx *ctx, int dtype, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs, size_t pack_rows, knox_enc **out) {
    if (!ctx || !out || !n_bufs || !bufs || !lens || dtype_width(dtype) == 0 || dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (dtype_is_float(dtype)) return KNOX_E_UNSUPPORTED;  // float containers take the decoding path (knox_col_upload_raw_containers)
    if (pack_rows == 0 || (pack_rows % kEncChunkRows) || n_bufs > 0xffffffffull) return KNOX_E_INVALID;  // whole 8192-row chunks
    *out = nullptr;
    hipSetDevice(ctx->device);
    const int w = dtype_width(dtype);
    // pass 1: headers -- how every pack will be scanned, element counts
    struct Plan { uint32_t kind; size_t off1, len1, n1, off2, len2, n2; uint64_t base, delta; };
    std::vector<Plan> pl(n_bufs);
    size_t total_vals = 0, total_runs = 0, total_bytes = 0, n_dict = 0;
    for (size_t i = 0; i < n_bufs; i++) {
        const uint8_t *b = bufs[i];
        if (!b || lens[i] < 2) return KNOX_E_INVALID;
        Plan &p = pl[i];
        p = Plan{7, 0, lens[i], 0, 0, 0, 0, 0, 0};
        total_bytes += lens[i];
        const bool nested = (b[0] == 5 || b[0] == 3) && lens[i] >= 3 && (b[1] == 3 || b[1] == 5);  // a dictionary / run-end child reports its rows late
        if ((b[0] == 5 || b[0] == 3) && !nested) {
            size_t n1 = 0, c1 = 0, n2 = 0, c2 = 0;
            int rc = container_measure(b + 1, lens[i] - 1, w, false, &n1, &c1, 1);
            if (rc) return rc;
            if (1 + c1 >= lens[i]) return KNOX_E_RANGE;
            const bool nested2 = b[1 + c1] == 3 || b[1 + c1] == 5;
            if (!nested2) {
                rc = container_measure(b + 1 + c1, lens[i] - 1 - c1, b[0] == 5 ? 2 : 4, false, &n2, &c2, 1);
                if (rc) return rc;
                if (b[0] == 5 && n1 >= 1 && n1 <= 65536 && n2 >= 1) {
                    p = Plan{5, 1, c1, n1, 1 + c1, c2, n2, 0, 0};
                    total_vals += n1;
                    n_dict++;
                    continue;
                }
                if (b[0] == 3 && n1 >= 1 && n1 == n2) {
                    p = Plan{3, 1, c1, n1, 1 + c1, c2, n2, 0, 0};
                    total_vals += n1;
                    total_runs += n1;
                    continue;
                }
            }
        }
        if (b[0] == 1 || b[0] == 2) {  // [1][base][N] / [2][For][Delta][N] (int_const.go:67-84, int_delta.go:70-95)
            size_t o = 1;
            uint64_t base = 0, delta = 0, n = 0;
            int k = knox_uvarint(b + o, lens[i] - o, &base);
            if (k == 0) return KNOX_E_INVALID;
            o += (size_t)k;
            if (b[0] == 2) {
                k = knox_uvarint(b + o, lens[i] - o, &delta);
                if (k == 0) return KNOX_E_INVALID;
                o += (size_t)k;
            }
            k = knox_uvarint(b + o, lens[i] - o, &n);
            if (k == 0 || n == 0) return KNOX_E_INVALID;
            p = Plan{(uint32_t)b[0], 0, 0, 0, 0, 0, (size_t)n, base, delta};
            continue;
        }
        // everything else: the container decoded on the device (W B per row), scanned as raw values
        size_t n = 0, c = 0;
        int rc = container_measure(b, lens[i], w, false, &n, &c, 0);
        if (rc) return rc;
        if (b[0] == 3) {  // a (nested) run-end container knows its row count only from its ends
            rc = container_rows(ctx, b, lens[i], w, false, &n);
            if (rc) return rc;
        }
        if (n == 0) return KNOX_E_INVALID;
        p.n2 = n;
        total_vals += n;
    }
    knox_enc *e = new (std::nothrow) knox_enc();
    if (!e) return KNOX_E_NOMEM;
    e->ctx = ctx;
    e->dtype = dtype;
    e->pack_rows = pack_rows;
    e->n_packs = n_bufs;
    e->container_bytes = total_bytes;
    hipError_t he = dev_malloc(ctx, &e->d_vals, total_vals * (size_t)w + 64);
    if (he == hipSuccess) he = dev_malloc(ctx, (void **)&e->d_packs, n_bufs * sizeof(EncPack) + 64);
    if (he == hipSuccess) he = dev_malloc(ctx, (void **)&e->d_lists, n_bufs * 4 + 64);
    if (he == hipSuccess && n_dict) he = dev_malloc(ctx, (void **)&e->d_codes, n_dict * pack_rows * 2 + 256);
    if (he == hipSuccess && total_runs) he = dev_malloc(ctx, (void **)&e->d_ends, total_runs * 4 + 64);
    if (he != hipSuccess) {
        ctx->last_hip = (int)he;
        knox_enc_free(e);
        return KNOX_E_NOMEM;
    }
    // pass 2: payloads onto the device (children / whole containers decoded there), pack table
    std::vector<EncPack> packs(n_bufs);
    std::vector<uint32_t> l_dict, l_run, l_aff, l_raw;
    size_t voff = 0, coff = 0, eoff = 0, rows = 0, resident = 0;
    int rc = KNOX_OK;
    for (size_t i = 0; i < n_bufs && !rc; i++) {
        const Plan &c = pl[i];
        EncPack hp;
        memset(&hp, 0, sizeof hp);
        hp.kind = c.kind;
        uint32_t prows = 0;
        if (c.kind == 5) {
            rc = container_decode(ctx, bufs[i] + c.off1, c.len1, w, false, (char *)e->d_vals + voff * (size_t)w, c.n1, 1);
            if (!rc) rc = container_decode(ctx, bufs[i] + c.off2, c.len2, 2, false, e->d_codes + coff, c.n2, 1);
            hp.off = voff, hp.code_off = coff, hp.len = (uint32_t)c.n1;
            prows = (uint32_t)c.n2;
            voff += c.n1;
            coff += pack_rows;
            resident += c.n1 * (size_t)w + c.n2 * 2;
            l_dict.push_back((uint32_t)i);
        } else if (c.kind == 3) {
            rc = container_decode(ctx, bufs[i] + c.off1, c.len1, w, false, (char *)e->d_vals + voff * (size_t)w, c.n1, 1);
            if (!rc) rc = container_decode(ctx, bufs[i] + c.off2, c.len2, 4, false, e->d_ends + eoff, c.n2, 1);
            uint32_t last = 0;  // the pack's row count = its last inclusive end + 1 (int_runend.go:109)
            if (!rc && (hipMemcpyAsync(&last, e->d_ends + eoff + (c.n2 - 1), 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess))
                rc = KNOX_E_HIP;
            hp.off = voff, hp.code_off = eoff, hp.len = (uint32_t)c.n1;   // code_off of a run-end pack: its first end
            prows = last + 1;
            voff += c.n1;
            eoff += c.n1;
            resident += c.n1 * ((size_t)w + 4);
            l_run.push_back((uint32_t)i);
        } else if (c.kind == 1 || c.kind == 2) {
            hp.base = c.base, hp.delta = c.kind == 2 ? c.delta : 0;
            prows = (uint32_t)c.n2;
            l_aff.push_back((uint32_t)i);
        } else {
            rc = container_decode(ctx, bufs[i], lens[i], w, false, (char *)e->d_vals + voff * (size_t)w, c.n2, 0);
            hp.off = voff, hp.len = (uint32_t)c.n2;
            prows = (uint32_t)c.n2;
            voff += c.n2;
            resident += c.n2 * (size_t)w;
            l_raw.push_back((uint32_t)i);
        }
        if (!rc && ((uint64_t)prows > pack_rows || prows == 0 || (i + 1 < n_bufs && (uint64_t)prows != pack_rows))) rc = KNOX_E_INVALID;  // only the last pack may be short
        hp.rows = prows;
        packs[i] = hp;
        rows += prows;
    }
    if (!rc) rc = upload(ctx, e->d_packs, packs.data(), n_bufs * sizeof(EncPack));
    if (!rc) {
        std::vector<uint32_t> all;
        all.insert(all.end(), l_dict.begin(), l_dict.end());
        all.insert(all.end(), l_run.begin(), l_run.end());
        all.insert(all.end(), l_aff.begin(), l_aff.end());
        all.insert(all.end(), l_raw.begin(), l_raw.end());
        rc = upload(ctx, e->d_lists, all.data(), all.size() * 4);
    }
    if (rc) {
        knox_enc_free(e);
        return rc;
    }
    e->n_dict = l_dict.size(), e->n_run = l_run.size(), e->n_aff = l_aff.size(), e->n_raw = l_raw.size();
    e->n_rows = rows;
    e->resident_bytes = resident + n_bufs * (sizeof(EncPack) + 4);
    *out = e;
    return KNOX_OK;
}

KNOX_API int knox_enc_scan(knox_ctx *ctx, const knox_enc *e, int op, uint64_t a, uint64_t b, int want_sum, knox_bits *bits_out, int64_t *pack_counts,
                           int64_t *count, knox_agg_state *sum_state) {
    if (!ctx || !e || e->ctx != ctx || (want_sum && !sum_state)) return KNOX_E_INVALID;
    if (bits_out && bits_out->size != e->n_rows) return KNOX_E_INVALID;
    if (pack_counts && !bits_out) return KNOX_E_INVALID;  // per-pack counts come from the bitset
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    LeafHost lh;
    int rc = make_leaf(e->dtype, op, a, b, lh);
    if (rc) return rc;
    const int grid_max = std::max(1, ctx->n_cus * 8);
    rc = ensure(ctx, ctx->partials, sizeof(AggPartial) * (size_t)(4 * grid_max + 1) * 16);
    if (!rc) rc = ensure(ctx, ctx->states, sizeof(AggStateDev) * 16 + 8 * 16);
    if (rc) return rc;
    EncScanArgs ea;
    memset(&ea, 0, sizeof ea);
    ea.vals = e->d_vals;
    ea.codes = e->d_codes;
    ea.ends = e->d_ends;
    ea.packs = e->d_packs;
    ea.width = (uint32_t)dtype_width(e->dtype);
    ea.pack_rows = e->pack_rows;
    ea.n_rows = e->n_rows;
    ea.lo = lh.lo;
    ea.diff = lh.diff;
    ea.wmask = wmask(dtype_width(e->dtype));
    ea.negate = lh.neg ? 1u : 0u;
    ea.want_sum = want_sum ? 1u : 0u;
    ea.bits = bits_out ? bits_out->dev : nullptr;
    AggPartial *part = (AggPartial *)ctx->partials.p;
    int n_part = 0;                       // partials written so far: one launch after the other appends its workgroups
    hipError_t he = hipSuccess;
    prof_mark(ctx, 0);
    const uint32_t *lists = e->d_lists;
    if (e->n_dict && he == hipSuccess) {  // the dominant kernel when there are dictionary packs
        ea.list = lists, ea.n_list = (uint32_t)e->n_dict, ea.partials = part + n_part;
        const uint64_t n_chunks = (uint64_t)e->n_dict * (e->pack_rows / kEncChunkRows);
        // latency bound (one 4 KiB wave iteration in flight per wave): the count / bitset form holds 8 KiB of LDS and runs 8 workgroups
        // per CU; with the sum's per-code counters (40 KiB) 4 fit
        const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)std::min(grid_max, ctx->n_cus * (ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : (want_sum ? 4 : 8))));
        he = launch_dict_scan(ctx->stream, grid, ea);
        n_part += grid;
    }
    if (e->n_run && he == hipSuccess) {
        ea.list = lists + e->n_dict, ea.n_list = (uint32_t)e->n_run, ea.partials = part + n_part;
        const uint64_t n_words = (uint64_t)e->n_run * (e->pack_rows >> 6);
        const int grid_bits = (int)std::min<uint64_t>((n_words + kBlock - 1) / kBlock, (uint64_t)ctx->n_cus * 16);
        const int grid = (int)std::min<uint64_t>(e->n_run, (uint64_t)grid_max);
        he = launch_runend_scan(ctx->stream, grid_bits, grid, ea);
        n_part += grid;
    }
    for (int raw = 0; raw < 2 && he == hipSuccess; raw++) {
        const size_t n = raw ? e->n_raw : e->n_aff;
        if (!n) continue;
        ea.list = lists + e->n_dict + e->n_run + (raw ? e->n_aff : 0), ea.n_list = (uint32_t)n, ea.partials = part + n_part;
        const uint64_t n_words = (uint64_t)n * (e->pack_rows >> 6);
        const int grid = (int)std::min<uint64_t>((n_words + kWavesPerBlock - 1) / kWavesPerBlock, (uint64_t)grid_max);
        he = launch_flat_scan(ctx->stream, grid, raw != 0, ea);
        n_part += grid;
    }
    prof_mark(ctx, 1);
    if (he == hipSuccess)
        he = launch_finalize(ctx->stream, dtype_width(e->dtype), part, n_part, nullptr, 0, e->dtype, (AggStateDev *)ctx->states.p,
                             (int64_t *)((AggStateDev *)ctx->states.p + 16));
    if (he != hipSuccess) { ctx->last_hip = (int)he; return KNOX_E_HIP; }
    if (pack_counts) {
        rc = ensure(ctx, ctx->pack_counts, e->n_packs * 8 + 64);
        if (rc) return rc;
        he = launch_popcount_segments(ctx->stream, bits_out->dev, e->n_rows, e->pack_rows, (int64_t *)ctx->pack_counts.p, e->n_packs);
        if (he != hipSuccess) { ctx->last_hip = (int)he; return KNOX_E_HIP; }
        rc = download(ctx, pack_counts, ctx->pack_counts.p, e->n_packs * 8);
        if (rc) return rc;
    }
    AggStateDev st;
    HIPCHK(ctx, hipMemcpyAsync(&st, ctx->states.p, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (count) *count = st.count;
    if (sum_state) memcpy(sum_state, &st, sizeof st);
    return KNOX_OK;
}
