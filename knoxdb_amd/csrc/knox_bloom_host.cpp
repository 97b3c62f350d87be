// knox_bloom_host.cpp -- bloom filters per pack on the host side: the probe value's hash (fixed-width and []byte
// XXH3-64), filter upload / build / probe / download.
#include "knox_internal.h"
#include "knox_xxh3.h"  // host flavour: KNOX_HD / KNOX_XXH3_CONST empty

// ------------------------------------------------------------------------------------------------
// bloom
// ------------------------------------------------------------------------------------------------
static inline uint64_t h_rol64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t h_rrmxmx(uint64_t h, uint64_t len) {
    h ^= h_rol64(h, 49) ^ h_rol64(h, 24);
    h *= 0x9fb21c651e98df25ull;
    h ^= (h >> 35) + len;
    h *= 0x9fb21c651e98df25ull;
    return h ^ (h >> 28);
}
static uint64_t h_avalanche_small(uint64_t x) {
    x ^= x >> 33; x *= 14029467366897019727ull; x ^= x >> 29; x *= 1609587929392839161ull; x ^= x >> 32;
    return x;
}

// internal/hash/hash.go:67-92 + xxh3.go:22-58.  float32/float64 hash their IEEE bytes with the
// []byte hash, which for 4 and 8 byte inputs is the same 4..8 byte XXH3 branch as u32 / u64.
// hash.Hash (internal/hash/hash.go:26): XXH3-64 of a byte string -- the probe value of bytes filters
KNOX_API uint64_t knox_hash_bytes(const uint8_t *val, size_t len) {
    if (len && !val) return 0;
    return knox_xxh3::hash(val, (uint64_t)len);
}

KNOX_API uint64_t knox_hash_value(int dtype, const void *val) {
    if (!val) return 0;
    const uint64_t K = 0x1cad21f72c81017cull ^ 0xdb979083e96dd4deull;
    switch (dtype_width(dtype)) {
    case 8: { uint64_t v; memcpy(&v, val, 8); return h_rrmxmx(((v >> 32) + (v << 32)) ^ K, 8); }
    case 4: { uint32_t v; memcpy(&v, val, 4); return h_rrmxmx(((uint64_t)v + ((uint64_t)v << 32)) ^ K, 4); }
    case 2: { uint16_t v; memcpy(&v, val, 2); uint64_t h = (((uint64_t)v * ((1ull << 24) + 1)) >> 8) + (2ull << 8); return h_avalanche_small(h ^ (0xbe4ba423ull ^ 0x396cfeb8ull)); }
    case 1: { uint8_t v; memcpy(&v, val, 1); uint64_t h = ((uint64_t)v * ((1ull << 24) + (1ull << 16) + 1)) + (1ull << 8); return h_avalanche_small(h ^ (0xbe4ba423ull ^ 0x396cfeb8ull)); }
    default: return 0;
    }
}

struct knox_bloom {
    knox_ctx *ctx;
    uint8_t *d_filters = nullptr;
    uint64_t *d_offsets = nullptr;
    uint8_t *d_maybe = nullptr;
    size_t n_packs = 0, bytes = 0;
    int batches = 0, exact_batches = 0;  // how the filters were built (knox_bloom_build_info)
    std::vector<uint64_t> h_offsets;
};

static int bloom_alloc(knox_ctx *ctx, const uint64_t *offsets, size_t n_packs, knox_bloom **out) {
    *out = nullptr;
    for (size_t p = 0; p < n_packs; p++) {
        if (offsets[p + 1] < offsets[p]) return KNOX_E_INVALID;
        const uint64_t len = offsets[p + 1] - offsets[p];
        if (len > 1 && ((len - 1) & (len - 2)) != 0) return KNOX_E_INVALID;  // m must be a power of two (bloom.go:83-101)
    }
    knox_bloom *b = new (std::nothrow) knox_bloom();
    if (!b) return KNOX_E_NOMEM;
    b->ctx = ctx;
    b->n_packs = n_packs;
    b->bytes = (size_t)offsets[n_packs];
    b->h_offsets.assign(offsets, offsets + n_packs + 1);
    if (dev_malloc(ctx, (void **)&b->d_filters, b->bytes + 64) != hipSuccess || dev_malloc(ctx, (void **)&b->d_offsets, (n_packs + 1) * 8 + 64) != hipSuccess ||
        dev_malloc(ctx, (void **)&b->d_maybe, n_packs + 64) != hipSuccess) {
        if (b->d_filters) dev_free(b->ctx, b->d_filters);
        if (b->d_offsets) dev_free(b->ctx, b->d_offsets);
        delete b;
        return KNOX_E_NOMEM;
    }
    int rc = upload(ctx, b->d_offsets, offsets, (n_packs + 1) * 8);
    if (rc) {
        knox_bloom_free(b);
        return rc;
    }
    *out = b;
    return KNOX_OK;
}

KNOX_API int knox_bloom_upload(knox_ctx *ctx, const uint8_t *filters, const uint64_t *offsets, size_t n_packs, knox_bloom **out) {
    if (!ctx || !offsets || !out || (n_packs && offsets[n_packs] && !filters)) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = bloom_alloc(ctx, offsets, n_packs, out);
    if (rc) return rc;
    rc = upload(ctx, (*out)->d_filters, filters, (*out)->bytes);
    if (rc) {
        knox_bloom_free(*out);
        *out = nullptr;
    }
    return rc;
}

KNOX_API int knox_bloom_build(knox_ctx *ctx, const knox_col *col, int factor, knox_bloom **out) {
    if (!ctx || !col || !out || factor <= 0 || col->pack_rows == 0) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, col); if (rc_) return rc_; }
    const size_t n_packs = (col->n_rows + col->pack_rows - 1) / col->pack_rows;
    std::vector<uint64_t> offsets(n_packs + 1);
    uint64_t off = 0;
    for (size_t p = 0; p < n_packs; p++) {
        const size_t rows = std::min(col->pack_rows, col->n_rows - p * col->pack_rows);
        uint64_t m = 8;  // bloom.go:242-249 pow2: next power of two >= cardinality*factor*8 bits
        while (m < (uint64_t)rows * (uint64_t)factor * 8) m <<= 1;
        offsets[p] = off;
        off += 1 + (m >> 3);
    }
    offsets[n_packs] = off;
    int rc = bloom_alloc(ctx, offsets.data(), n_packs, out);
    if (rc) return rc;
    knox_bloom *b = *out;
    hipError_t e = hipSuccess;
    const uint64_t max_fbytes = n_packs ? offsets[1] - offsets[0] - 1 : 0;   // pack 0 is a full pack: the largest filter
    static const bool part_on = !(getenv("KNOX_BLOOM_ATOMIC") && atoi(getenv("KNOX_BLOOM_ATOMIC")) != 0);  // 1: the atomic kernels (A/B runs)
    if (part_on && col->pack_rows >= 4096 && max_fbytes <= ((size_t)2 << 20)) {
        // partitioned build (knox_bloom.hip): probe locations binned by filter segment, every segment OR-ed together in LDS
        // and stored once.  Batches of packs bound the scratch (8 B per row + 2 KiB per pack) to ~1 GiB
        const BytesView bv = col->dtype == KNOX_BYTES ? bytes_view(col, col->n_rows) : BytesView();
        const int width = col->dtype == KNOX_BYTES ? 0 : col->width;
        // Round 4: the one-read build (fixed entry regions, no count pass) first; a batch whose regions overflow -- heavy hitters
        // in a column whose filters are sized for many distinct values -- is rebuilt with the exact count / scan / bin / fill
        // passes, and if the FIRST batch overflows the rest of the column goes the exact way at once.  KNOX_BLOOM_EXACT=1: exact only.
        static const bool one_on = !(getenv("KNOX_BLOOM_EXACT") && atoi(getenv("KNOX_BLOOM_EXACT")) != 0);
        const size_t per_pack = std::max(bloom_part_scratch_bytes(1, col->pack_rows), bloom_one_scratch_bytes(1, col->pack_rows, max_fbytes));
        size_t batch = std::max<size_t>(1, ((size_t)1 << 30) / per_pack);
        batch = std::min(batch, n_packs);
        const size_t n_batches = (n_packs + batch - 1) / batch;
        rc = ensure(ctx, ctx->sort_buf, std::max(bloom_part_scratch_bytes(batch, col->pack_rows), bloom_one_scratch_bytes(batch, col->pack_rows, max_fbytes)) + 256,
                    ARENA_WRITE);
        if (!rc) rc = ensure_host(ctx, ctx->bloom_flags, ctx->bloom_flags_cap, n_batches * 4 + 64);
        if (rc) {
            knox_bloom_free(b);
            *out = nullptr;
            return rc;
        }
        uint32_t *h_flags = (uint32_t *)ctx->bloom_flags;   // page-locked: one word per batch, copied behind the batch, read after one sync
        bool exact = !one_on;
        for (size_t bi = 0, p0 = 0; p0 < n_packs && e == hipSuccess; p0 += batch, bi++) {
            const size_t np = std::min(batch, n_packs - p0);
            h_flags[bi] = exact ? 2u : 0u;             // 2: built exactly, nothing to check
            b->batches++;
            if (exact) {
                b->exact_batches++;
                e = launch_bloom_build_part(ctx->stream, width, col->dev, &bv, col->n_rows, col->pack_rows, p0, np, max_fbytes, b->d_filters, b->d_offsets,
                                            ctx->sort_buf.p);
                continue;
            }
            uint32_t *d_flag = nullptr;
            e = launch_bloom_build_one(ctx->stream, width, col->dev, &bv, col->n_rows, col->pack_rows, p0, np, max_fbytes, b->d_filters, b->d_offsets,
                                       ctx->sort_buf.p, &d_flag);
            if (e == hipSuccess) e = hipMemcpyAsync(&h_flags[bi], d_flag, 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && bi == 0 && n_batches > 1) {
                e = hipStreamSynchronize(ctx->stream);  // the first batch decides how the rest is built
                exact = e == hipSuccess && h_flags[0] != 0;
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        for (size_t bi = 0, p0 = 0; p0 < n_packs && e == hipSuccess; p0 += batch, bi++) {
            if (h_flags[bi] != 1u) continue;           // 1: a region of this batch overflowed
            e = launch_bloom_build_part(ctx->stream, width, col->dev, &bv, col->n_rows, col->pack_rows, p0, std::min(batch, n_packs - p0), max_fbytes,
                                        b->d_filters, b->d_offsets, ctx->sort_buf.p);
            ctx->bloom_exact_batches++;
            b->exact_batches++;
        }
    } else {
        e = launch_zero_bytes(ctx->stream, b->d_filters, b->bytes);   // 2 GiB of filters per 2^30 rows: 1.1 ms through hipMemsetAsync's fill kernel, 0.4 ms here
        if (e == hipSuccess)
            e = col->dtype == KNOX_BYTES
                    ? launch_bloom_build_bytes(ctx->stream, bytes_view(col, col->n_rows), b->d_filters, b->d_offsets)
                    : launch_bloom_build(ctx->stream, col->width, col->dev, col->n_rows, col->pack_rows, b->d_filters, b->d_offsets);
    }
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        knox_bloom_free(b);
        *out = nullptr;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

KNOX_API int knox_bloom_build_u64(knox_ctx *ctx, const knox_col *col, int factor, knox_bloom **out) {
    if (!col || col->width != 8) return KNOX_E_INVALID;
    return knox_bloom_build(ctx, col, factor, out);
}

KNOX_API size_t knox_bloom_packs(const knox_bloom *b) { return b ? b->n_packs : 0; }
KNOX_API int knox_bloom_build_info(const knox_bloom *b, int *batches, int *exact_batches) {
    if (!b || !batches || !exact_batches) return KNOX_E_INVALID;
    *batches = b->batches;
    *exact_batches = b->exact_batches;
    return KNOX_OK;
}
KNOX_API size_t knox_bloom_bytes(const knox_bloom *b) { return b ? b->bytes : 0; }

KNOX_API int knox_bloom_download(knox_ctx *ctx, const knox_bloom *b, uint8_t *filters, size_t filters_cap, uint64_t *offsets) {
    if (!ctx || !b || !offsets || (b->bytes && !filters)) return KNOX_E_INVALID;
    if (filters_cap < b->bytes) return KNOX_E_RANGE;
    hipSetDevice(ctx->device);
    memcpy(offsets, b->h_offsets.data(), (b->n_packs + 1) * 8);
    return download(ctx, filters, b->d_filters, b->bytes);
}

KNOX_API int knox_bloom_probe(knox_ctx *ctx, const knox_bloom *b, uint64_t h, uint8_t *maybe_out) {
    if (!ctx || !b || (b->n_packs && !maybe_out)) return KNOX_E_INVALID;
    if (b->n_packs == 0) return KNOX_OK;
    hipSetDevice(ctx->device);
    hipError_t e = launch_bloom_probe(ctx->stream, b->d_filters, b->d_offsets, b->n_packs, h, b->d_maybe);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    HIPCHK(ctx, hipMemcpyAsync(maybe_out, b->d_maybe, b->n_packs, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KNOX_OK;
}

KNOX_API int knox_bloom_free(knox_bloom *b) {
    if (!b) return KNOX_E_INVALID;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    if (b->d_filters) dev_free(b->ctx, b->d_filters);
    if (b->d_offsets) dev_free(b->ctx, b->d_offsets);
    if (b->d_maybe) dev_free(b->ctx, b->d_maybe);
    delete b;
    return KNOX_OK;
}

