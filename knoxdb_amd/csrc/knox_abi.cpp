// knox_abi.cpp -- C ABI (include/knoxgpu.h) over the gfx950 kernels: contexts, columns, bitsets, the planner and the scan.
//
// Host side of the drop-in boundary: contexts, HBM-resident columns with a pinned staging
// ring (hipMemcpyAsync), the predicate-tree planner (postfix tree -> key-space leaves + truth
// table, or a multi-pass plan over scratch bitsets), result collection, exact finalisation of
// reducer values, and the RCCL exchange of partial aggregate states.
//
// There is NO CPU fallback: if the HIP runtime or the device is missing every entry point
// fails with KNOX_E_NODEVICE / KNOX_E_HIP.
#include "knox_internal.h"


// ------------------------------------------------------------------------------------------------
// library / context
// ------------------------------------------------------------------------------------------------
KNOX_API int knox_abi_version(void) { return KNOX_ABI_VERSION; }
KNOX_API int knox_scan_occupancy_mismatches(void) { return g_scan_occupancy_mismatches.load(); }
KNOX_API int knox_scan_launch_shape(int width, int n_leaves, int heavy, int is_float, int float_sum, int *per_cu, uint32_t *park_words) {
    if ((width != 1 && width != 2 && width != 4 && width != 8) || n_leaves < 1 || n_leaves > kMaxLeaves || !per_cu || !park_words) return KNOX_E_INVALID;
    const ScanShape sh = scan_shape(width, n_leaves, heavy != 0, is_float != 0, float_sum != 0);
    *per_cu = sh.per_cu;
    *park_words = sh.park_words;
    return KNOX_OK;
}

KNOX_API const char *knox_strerror(int err) {
    switch (err) {
    case KNOX_OK: return "ok";
    case KNOX_E_INVALID: return "invalid argument";
    case KNOX_E_NODEVICE: return "no HIP device";
    case KNOX_E_HIP: return "HIP runtime error";
    case KNOX_E_NOMEM: return "out of memory";
    case KNOX_E_UNSUPPORTED: return "unsupported request";
    case KNOX_E_COMM: return "RCCL error";
    case KNOX_E_RANGE: return "out of range";
    case KNOX_E_BUSY: return "too many scans in flight: collect one first";
    default: return "unknown error";
    }
}

KNOX_API int knox_device_count(int *count) {
    if (!count) return KNOX_E_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return KNOX_E_NODEVICE;
    }
    *count = n;
    return KNOX_OK;
}

KNOX_API int knox_last_hip_error(const knox_ctx *ctx) { return ctx ? ctx->last_hip : 0; }

static int ctx_open(int device, void *ext_stream, bool use_ext, knox_ctx **out) {
    if (!out) return KNOX_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return KNOX_E_NODEVICE;
    if (device < 0 || device >= n) return KNOX_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return KNOX_E_NODEVICE;
    knox_ctx *c = new (std::nothrow) knox_ctx();
    if (!c) return KNOX_E_NOMEM;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    if (use_ext) {
        c->stream = (hipStream_t)ext_stream;
        c->own_stream = false;
    } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return KNOX_E_HIP;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return KNOX_E_HIP;
    }
    const char *env = getenv("KNOX_BLOCKS_PER_CU");
    if (env && atoi(env) > 0) c->blocks_per_cu = atoi(env);
#ifdef KNOX_DIAG
    env = getenv("KNOX_VARIANT");
    if (env) c->variant = atoi(env);
#endif
    env = getenv("KNOX_PLACEMENT");
    if (env) c->placement = atoi(env);
    env = getenv("KNOX_DEFER");
    c->defer_flags = env && atoi(env) == 0 ? 1u : env && atoi(env) == 2 ? 2u : 0u;
    // arena budget: KNOX_ARENA_GIB, knox_ctx_set_arena, or three quarters of the device; physical memory is only taken
    // as allocations need it, 2 GiB at a time
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
    size_t budget = total_b / 4 * 3;
    env = getenv("KNOX_ARENA_GIB");
    if (env && atof(env) > 0) budget = (size_t)(atof(env) * (double)((size_t)1 << 30));
    c->arena.configure(device, budget, arena_probe, c, arena_probe_multi, &c->probe_launches, &c->probe_ms);
    *out = c;
    return KNOX_OK;
}

KNOX_API int knox_ctx_open(int device, knox_ctx **out) { return ctx_open(device, nullptr, false, out); }
KNOX_API int knox_ctx_open_stream(int device, void *hip_stream, knox_ctx **out) { return ctx_open(device, hip_stream, true, out); }

KNOX_API int knox_ctx_close(knox_ctx *ctx) {
    if (!ctx) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm) knox_comm_destroy(ctx);
    DevBuf *bufs[] = {&ctx->partials, &ctx->states, &ctx->tmp_vals, &ctx->tmp_bits, &ctx->pack_skip, &ctx->pack_counts,
                      &ctx->misc, &ctx->mom_partials, &ctx->tile_counts, &ctx->idx_scratch, &ctx->idx_rows, &ctx->sort_buf, &ctx->bytes_val, &ctx->inset_keys, &ctx->ring_dev,
                      &ctx->tmp_bits_parked[0], &ctx->tmp_bits_parked[1]};
    for (DevBuf *b : bufs) dev_free(ctx, b->p);
    if (ctx->comm_buf.p) hipFree(ctx->comm_buf.p);
    if (ctx->probe_partials) hipFree(ctx->probe_partials);
    ctx->arena.destroy();  // columns / bitsets still alive after this point are dangling: free them before closing the context
    for (hipEvent_t e : ctx->probe_ev)
        if (e) hipEventDestroy(e);
    for (int i = 0; i < 2; i++) {
        if (ctx->stage[i]) hipHostFree(ctx->stage[i]);
        if (ctx->stage_ev[i]) hipEventDestroy(ctx->stage_ev[i]);
    }
    for (hipEvent_t e : ctx->prof_ev)
        if (e) hipEventDestroy(e);
    for (auto &r : ctx->res) {
        if (r.h_res) hipHostFree(r.h_res);
        if (ctx->h_view) { hipHostFree(ctx->h_view); ctx->h_view = nullptr; }
        if (ctx->cmp_res) { hipHostFree(ctx->cmp_res); ctx->cmp_res = nullptr; }
        if (ctx->bloom_flags) { hipHostFree(ctx->bloom_flags); ctx->bloom_flags = nullptr; }
        if (r.h_packs) hipHostFree(r.h_packs);
        if (r.h_skip) hipHostFree(r.h_skip);
        if (r.done) hipEventDestroy(r.done);
    }
    for (int i = 0; i < 2; i++) {
        if (ctx->cstage[i]) hipHostFree(ctx->cstage[i]);
        if (ctx->cstage_ev[i]) hipEventDestroy(ctx->cstage_ev[i]);
    }
    if (ctx->fence_ev) hipEventDestroy(ctx->fence_ev);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return KNOX_OK;
}

KNOX_API int knox_ctx_sync(knox_ctx *ctx) {
    if (!ctx) return KNOX_E_INVALID;
    if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KNOX_OK;
}

KNOX_API void *knox_ctx_stream(knox_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

KNOX_API int knox_timer_start(knox_ctx *ctx) {
    if (!ctx) return KNOX_E_INVALID;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return KNOX_OK;
}

KNOX_API int knox_timer_stop(knox_ctx *ctx, float *elapsed_ms) {
    if (!ctx || !elapsed_ms) return KNOX_E_INVALID;
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return KNOX_OK;
}

KNOX_API int knox_ctx_set_tuning(knox_ctx *ctx, int blocks_per_cu) {
    if (!ctx) return KNOX_E_INVALID;
    if (blocks_per_cu >= 0) ctx->blocks_per_cu = blocks_per_cu;
    return KNOX_OK;
}
#ifdef KNOX_DIAG
// diagnostics builds only (knoxdb_amd/diag/, never the product, not in include/knoxgpu.h)
KNOX_API int knox_diag_set_variant(knox_ctx *ctx, int variant) {
    if (!ctx) return KNOX_E_INVALID;
    ctx->variant = variant;
    return KNOX_OK;
}
#endif

KNOX_API int knox_host_alloc(knox_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return KNOX_E_INVALID;
    *out = nullptr;
    if (bytes == 0) return KNOX_OK;
    hipSetDevice(ctx->device);
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        *out = nullptr;
        return KNOX_E_NOMEM;
    }
    return KNOX_OK;
}

KNOX_API int knox_host_free(knox_ctx *ctx, void *p) {
    if (!ctx) return KNOX_E_INVALID;
    if (!p) return KNOX_OK;
    hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipHostFree(p));
    return KNOX_OK;
}

KNOX_API int knox_ctx_set_placement(knox_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 1) return KNOX_E_INVALID;
    ctx->placement = mode;  // allocations made so far stay where they are and are freed where they came from
    return KNOX_OK;
}

KNOX_API int knox_ctx_set_arena(knox_ctx *ctx, size_t budget_bytes) {
    if (!ctx) return KNOX_E_INVALID;
    if (budget_bytes < ctx->arena.mapped_total) return KNOX_E_RANGE;  // already above it: free, knox_ctx_arena_trim, then shrink
    ctx->arena.budget = budget_bytes;
    return KNOX_OK;
}

KNOX_API int knox_ctx_reserve(knox_ctx *ctx, size_t bytes) {
    if (!ctx) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    if (!ctx->placement) return KNOX_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // growing the arena probes new chunks on the context's stream
    const hipError_t e = ctx->arena.reserve(bytes);
    if (e == hipErrorNotSupported) return KNOX_OK;
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_NOMEM;
    }
    return KNOX_OK;
}

KNOX_API int knox_ctx_set_probe_budget(knox_ctx *ctx, int max_launches, double max_ms) {
    if (!ctx || max_launches < 0 || !(max_ms >= 0.0)) return KNOX_E_INVALID;
    ctx->arena.grant(max_launches, max_ms);
    return KNOX_OK;
}

KNOX_API int knox_ctx_arena_info(knox_ctx *ctx, knox_arena_info *out) {
    if (!ctx || !out) return KNOX_E_INVALID;
    memset(out, 0, sizeof *out);
    out->budget_bytes = ctx->arena.budget;
    out->mapped_bytes = ctx->arena.mapped_total;
    out->used_bytes = ctx->arena.used_bytes();
    out->peak_used_bytes = ctx->arena.peak_used;
    out->write_buffers = ctx->arena.write_buffers;
    out->spare_write_buffer_bytes = ctx->arena.spare_bytes();
    out->chunks = ctx->arena.chunks_created;
    out->probes = ctx->arena.probes_run;
    out->mixed_zone_bytes = ctx->arena.mixed_bytes;
    out->enabled = ctx->placement && ctx->arena.usable;
    out->probe_launches = ctx->probe_launches;
    out->probe_ms = (float)ctx->probe_ms;
    out->unprobed_chunks = ctx->arena.unprobed_chunks;
    out->unverified_write_buffers = ctx->arena.unverified_write_buffers;
    return KNOX_OK;
}

KNOX_API int knox_ctx_arena_classes(knox_ctx *ctx, uint64_t class_bytes[4], int *classes_seen) {
    if (!ctx || !class_bytes) return KNOX_E_INVALID;
    for (int c = 0; c < 4; c++) class_bytes[c] = ctx->arena.zone_bytes(c + 1);
    if (classes_seen) *classes_seen = ctx->arena.classes_seen();
    return KNOX_OK;
}

KNOX_API int knox_ctx_arena_trim(knox_ctx *ctx, size_t *freed_bytes) {
    if (!ctx) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t f = ctx->arena.trim();
    if (freed_bytes) *freed_bytes = f;
    return KNOX_OK;
}

// bytes this context keeps from the driver WITHOUT having them in the arena's accounting: none, by construction
KNOX_API int knox_ctx_placement_info(knox_ctx *ctx, int64_t *held_bytes, int *held_regions) {
    if (!ctx) return KNOX_E_INVALID;
    if (held_bytes) *held_bytes = 0;
    if (held_regions) *held_regions = 0;
    return KNOX_OK;
}

KNOX_API int knox_prof_enable(knox_ctx *ctx, int on) {
    if (!ctx) return KNOX_E_INVALID;
    ctx->prof = on != 0;
    ctx->prof_n = 0;
    return KNOX_OK;
}

KNOX_API int knox_prof_read(knox_ctx *ctx, double *total_ms, int *launches) {
    if (!ctx || !total_ms || !launches) return KNOX_E_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double t = 0;
    for (int i = 0; i < ctx->prof_n; i++) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        t += ms;
    }
    *total_ms = t;
    *launches = ctx->prof_n;
    ctx->prof_n = 0;
    return KNOX_OK;
}

KNOX_API int knox_prof_read_each(knox_ctx *ctx, float *ms, int cap, int *launches) {
    if (!ctx || !ms || !launches || cap < 0) return KNOX_E_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int n = ctx->prof_n < cap ? ctx->prof_n : cap;
    for (int i = 0; i < n; i++) HIPCHK(ctx, hipEventElapsedTime(&ms[i], ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
    *launches = n;
    ctx->prof_n = 0;
    return KNOX_OK;
}

// bracket the dominant kernel launch with events when the stopwatch is on
void prof_mark(knox_ctx *ctx, int which) {
    if (!ctx->prof || ctx->prof_n >= knox_ctx::kProfRing) return;
    hipEvent_t &e = ctx->prof_ev[2 * ctx->prof_n + which];
    if (!e && hipEventCreate(&e) != hipSuccess) { e = nullptr; return; }
    hipEventRecord(e, ctx->stream);
    if (which == 1) ctx->prof_n++;
}

// ------------------------------------------------------------------------------------------------
// staging: host -> pinned ring -> HBM (two slots, copy of chunk i+1 overlaps the DMA of chunk i)
// ------------------------------------------------------------------------------------------------
static int stage_init(knox_ctx *ctx) {
    for (int i = 0; i < 2; i++) {
        if (!ctx->stage[i]) {
            hipError_t e = hipHostMalloc(&ctx->stage[i], knox_ctx::kStageBytes, hipHostMallocDefault);
            if (e != hipSuccess) {
                ctx->last_hip = (int)e;
                ctx->stage[i] = nullptr;
                return KNOX_E_NOMEM;
            }
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
        }
    }
    return KNOX_OK;
}

// is `p` page-locked host memory HIP can DMA from directly (knox_host_alloc, hipHostMalloc, hipHostRegister)?
static bool is_pinned_host(const void *p) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // plain pageable memory is "invalid value" to this query
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

int upload(knox_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return KNOX_OK;
    if (is_pinned_host(src)) {
        // page-locked source: the DMA engine reads it in place, no staging copy (PCIe rate)
        HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the caller's memory is not referenced after return
        return KNOX_OK;
    }
    int rc = stage_init(ctx);
    if (rc) return rc;
    size_t off = 0;
    int slot = 0;
    bool used[2] = {false, false};
    while (off < bytes) {
        const size_t n = std::min(bytes - off, knox_ctx::kStageBytes);
        if (used[slot]) HIPCHK(ctx, hipEventSynchronize(ctx->stage_ev[slot]));
        CopyPool::get().copy(ctx->stage[slot], (const char *)src + off, n);
        HIPCHK(ctx, hipMemcpyAsync((char *)dst + off, ctx->stage[slot], n, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->stage_ev[slot], ctx->stream));
        used[slot] = true;
        slot ^= 1;
        off += n;
    }
    // the caller's memory is not referenced after return; the ring is: wait for the last DMAs
    for (int i = 0; i < 2; i++)
        if (used[i]) HIPCHK(ctx, hipEventSynchronize(ctx->stage_ev[i]));
    return KNOX_OK;
}

int download(knox_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return KNOX_OK;
    if (is_pinned_host(dst)) {
        HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return KNOX_OK;
    }
    int rc = stage_init(ctx);
    if (rc) return rc;
    // two ring slots: the DMA of chunk i+1 runs while chunk i is copied out to the caller's pageable memory
    // (4 MiB pieces for small results: the first copy-out starts early; 16 MiB ones once the copy pool shares them)
    const size_t chunk = std::min<size_t>(knox_ctx::kStageBytes, bytes >= ((size_t)64 << 20) ? (size_t)16 << 20 : (size_t)4 << 20);
    size_t issued = 0, copied = 0;
    int slot = 0;
    size_t len[2] = {0, 0};
    while (copied < bytes) {
        while (issued < bytes && issued - copied < 2 * chunk && len[slot] == 0) {
            const size_t n = std::min(bytes - issued, chunk);
            HIPCHK(ctx, hipMemcpyAsync(ctx->stage[slot], (const char *)src + issued, n, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipEventRecord(ctx->stage_ev[slot], ctx->stream));
            len[slot] = n;
            issued += n;
            slot ^= 1;
        }
        const int done = len[slot] ? slot : slot ^ 1;  // the older of the slots in flight
        HIPCHK(ctx, hipEventSynchronize(ctx->stage_ev[done]));
        CopyPool::get().copy((char *)dst + copied, ctx->stage[done], len[done]);
        copied += len[done];
        len[done] = 0;
        slot = done;
    }
    return KNOX_OK;
}

// ------------------------------------------------------------------------------------------------
// columns
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// The arena's probe (knox_arena.h): scan / read-only for ONE (column region, write buffer) pair -- `rows` int64 rows at `p`
// scanned into `bits` (rows/8 bytes, the read+write mix) against the read-only form of the same scan, ALTERNATING
// launches timed one by one (a device that is busy with something else, e.g. the driver clearing memory, slows both
// alike).  Runs under its own kernel symbol (launch_probe_scan) so profiles keep probes and scans apart.
// ------------------------------------------------------------------------------------------------
static double probe_pair(knox_ctx *ctx, const void *p, uint64_t rows, void *bits) {
    const uint32_t iters = 2;
    const uint64_t tr = (uint64_t)tile_rows(8) * iters;
    if (rows < tr || rows % tr || !bits) return 0.0;
    const int grid = ctx->n_cus * 2;
    if (!ctx->probe_partials && hipMalloc(&ctx->probe_partials, sizeof(AggPartial) * (size_t)(grid + 1)) != hipSuccess) {
        (void)hipGetLastError();
        ctx->probe_partials = nullptr;
        return 0.0;
    }
    constexpr int kLaunches = 20;  // alternating read-only / scan; the first 4 of each are the clock ramp
    // every event on demand: probe_multi may have run first on this context and made only the first few
    for (int i = 0; i <= kLaunches; i++)
        if (!ctx->probe_ev[i] && hipEventCreate(&ctx->probe_ev[i]) != hipSuccess) {
            ctx->probe_ev[i] = nullptr;
            return 0.0;
        }
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.leaf[0].col = p;
    a.leaf[0].lo = 0;
    a.leaf[0].diff = ~0ull >> 2;  // contents are whatever the region held: any predicate will do
    a.tt = 2;
    a.agg_col = p;
    a.agg_flavor = FL_SIGNED;
    a.iters = iters;
    a.n_tiles = rows / tr;
    a.n_rows = rows;
    a.tail_first_row = rows;
    a.pack_shift = 255;
    a.flags = 1;  // bursts tile by tile: the classes are told apart by the UNdeferred effect (knox_scan.hip kDefer)
    a.partials = (AggPartial *)ctx->probe_partials;
    const int g = (int)std::min<uint64_t>(a.n_tiles, (uint64_t)grid);
    if (hipEventRecord(ctx->probe_ev[0], ctx->stream) != hipSuccess) return 0.0;
    for (int i = 0; i < kLaunches; i++) {
        a.bits = (i & 1) ? (uint32_t *)bits : nullptr;
        if (launch_probe_scan(ctx->stream, g, a) != hipSuccess || hipEventRecord(ctx->probe_ev[i + 1], ctx->stream) != hipSuccess) return 0.0;
    }
    if (hipEventSynchronize(ctx->probe_ev[kLaunches]) != hipSuccess) return 0.0;
    {
        float all_ms = 0.f;
        if (hipEventElapsedTime(&all_ms, ctx->probe_ev[0], ctx->probe_ev[kLaunches]) == hipSuccess) ctx->probe_ms += all_ms;
        ctx->probe_launches += kLaunches;
    }
    float ro[kLaunches / 2], mix[kLaunches / 2];
    int n_ro = 0, n_mix = 0;
    for (int i = 8; i < kLaunches; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->probe_ev[i], ctx->probe_ev[i + 1]) != hipSuccess || ms <= 0.f) return 0.0;
        if (i & 1) mix[n_mix++] = ms; else ro[n_ro++] = ms;
    }
    std::sort(ro, ro + n_ro);
    std::sort(mix, mix + n_mix);
    // rates: the scan moves 8.125 B per row, the read-only form 8 B
    return ((double)ro[n_ro / 2] / 8.0) / ((double)mix[n_mix / 2] / 8.125);
}
double arena_probe(void *user, const void *col, uint64_t rows, void *bits) { return probe_pair((knox_ctx *)user, col, rows, bits); }

// The short form for labelling a new chunk once two or more classes are known: ONE launch sequence measures the region
// against every class's reference window -- 2 ramp launches, then read-only, scan(ref 0) .. scan(ref n-1), scan(ref n-1) ..
// scan(ref 0), read-only (the mirrored order cancels a linear clock drift) = 2 + 2 (n + 1) launches: 8 for two classes
// instead of 20-40 per class.  A reading near the threshold is taken again by the caller with the long form.
static int probe_multi(knox_ctx *ctx, const void *p, uint64_t rows, void *const *bits, int n, double *ratios) {
    const uint32_t iters = 2;
    const uint64_t tr = (uint64_t)tile_rows(8) * iters;
    if (rows < tr || rows % tr || n < 1 || n > 4) return -1;
    const int grid = ctx->n_cus * 2;
    if (!ctx->probe_partials && hipMalloc(&ctx->probe_partials, sizeof(AggPartial) * (size_t)(grid + 1)) != hipSuccess) {
        (void)hipGetLastError();
        ctx->probe_partials = nullptr;
        return -1;
    }
    const int n_l = 2 + 2 * (n + 1);
    for (int i = 0; i <= n_l; i++)
        if (!ctx->probe_ev[i] && hipEventCreate(&ctx->probe_ev[i]) != hipSuccess) return -1;
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.leaf[0].col = p;
    a.leaf[0].diff = ~0ull >> 2;
    a.tt = 2;
    a.agg_col = p;
    a.agg_flavor = FL_SIGNED;
    a.iters = iters;
    a.n_tiles = rows / tr;
    a.n_rows = rows;
    a.tail_first_row = rows;
    a.pack_shift = 255;
    a.flags = 1;  // bursts tile by tile: the classes are told apart by the UNdeferred effect (knox_scan.hip kDefer)
    a.partials = (AggPartial *)ctx->probe_partials;
    const int g = (int)std::min<uint64_t>(a.n_tiles, (uint64_t)grid);
    // which buffer launch i writes: -1 = read-only
    int which[2 + 2 * (4 + 1)];
    int k = 0;
    which[k++] = -1, which[k++] = 0;                       // ramp
    which[k++] = -1;
    for (int c = 0; c < n; c++) which[k++] = c;
    for (int c = n - 1; c >= 0; c--) which[k++] = c;
    which[k++] = -1;
    if (hipEventRecord(ctx->probe_ev[0], ctx->stream) != hipSuccess) return -1;
    for (int i = 0; i < n_l; i++) {
        a.bits = which[i] < 0 ? nullptr : (uint32_t *)bits[which[i]];
        if (launch_probe_scan(ctx->stream, g, a) != hipSuccess || hipEventRecord(ctx->probe_ev[i + 1], ctx->stream) != hipSuccess) return -1;
    }
    if (hipEventSynchronize(ctx->probe_ev[n_l]) != hipSuccess) return -1;
    {
        float all_ms = 0.f;
        if (hipEventElapsedTime(&all_ms, ctx->probe_ev[0], ctx->probe_ev[n_l]) == hipSuccess) ctx->probe_ms += all_ms;
        ctx->probe_launches += n_l;
    }
    double ro = 0.0, mix[4] = {0, 0, 0, 0};
    for (int i = 2; i < n_l; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->probe_ev[i], ctx->probe_ev[i + 1]) != hipSuccess || ms <= 0.f) return -1;
        if (which[i] < 0) ro += ms; else mix[which[i]] += ms;
    }
    for (int c = 0; c < n; c++) ratios[c] = (ro / 8.0) / (mix[c] / 8.125);
    return 0;
}
int arena_probe_multi(void *user, const void *col, uint64_t rows, void *const *bits, int n, double *ratios) {
    return probe_multi((knox_ctx *)user, col, rows, bits, n, ratios);
}

int col_new(knox_ctx *ctx, int dtype, size_t n_rows, size_t pack_rows, bool alloc, knox_col **out) {
    if (!ctx || !out) return KNOX_E_INVALID;
    *out = nullptr;
    const int w = dtype == KNOX_BYTES ? 8 : dtype_width(dtype);
    if (w == 0) return KNOX_E_INVALID;
    if (pack_rows == 0) pack_rows = n_rows ? n_rows : 64;
    knox_col *c = new (std::nothrow) knox_col();
    if (!c) return KNOX_E_NOMEM;
    c->ctx = ctx;
    c->dtype = dtype;
    c->width = w;
    c->n_rows = n_rows;
    c->pack_rows = pack_rows;
    c->dev = nullptr;
    c->buf = nullptr;
    c->buf_len = 0;
    c->owned = alloc;
    if (alloc) {
        // pad to a whole 16-byte vector so the streaming kernels never read past the allocation
        size_t bytes = ((n_rows * (size_t)w + 255) & ~(size_t)255) + 256;
        hipError_t e = dev_malloc(ctx, &c->dev, bytes, ARENA_READ);
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            delete c;
            return KNOX_E_NOMEM;
        }
    }
    *out = c;
    return KNOX_OK;
}

KNOX_API int knox_col_upload(knox_ctx *ctx, int dtype, const void *host, size_t n_rows, size_t pack_rows, knox_col **out) {
    if (!ctx || !out || (n_rows && !host) || dtype == KNOX_BYTES) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = col_new(ctx, dtype, n_rows, pack_rows, true, out);
    if (rc) return rc;
    rc = upload(ctx, (*out)->dev, host, n_rows * (size_t)(*out)->width);
    if (rc) {
        knox_col_free(*out);
        *out = nullptr;
    }
    return rc;
}

// ---- ingest that overlaps the scans ----------------------------------------------------------------------
// The reference interleaves loadPack and filter.Match pack by pack (internal/pack/table/reader.go:329-334, loadPack :484).
// Here a RUN of packs is uploaded on the context's copy stream while the scan of the previous run is in flight on the
// scan stream; the only coupling is the column's `ready` event, which every reader waits for on the device (col_acquire).
int copy_stream_init(knox_ctx *ctx) {
    if (ctx->copy_stream) return KNOX_OK;
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->fence_ev, hipEventDisableTiming));
    return KNOX_OK;
}

// host -> HBM on the copy stream, enqueue only.  Page-locked sources are read in place by the DMA engine and must stay
// valid until the column's upload has completed (knox_col_wait / knox_col_ready); pageable sources are copied into the
// copy stream's own pinned ring here and are not referenced after return.
int upload_async(knox_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return KNOX_OK;
    if (is_pinned_host(src)) {
        // one DMA per 256 MiB: a single multi-GiB hipMemcpyAsync is not faster and delays the first byte
        const size_t piece = (size_t)256 << 20;
        for (size_t off = 0; off < bytes; off += piece)
            HIPCHK(ctx, hipMemcpyAsync((char *)dst + off, (const char *)src + off, std::min(piece, bytes - off), hipMemcpyHostToDevice, ctx->copy_stream));
        return KNOX_OK;
    }
    for (int i = 0; i < 2; i++) {
        if (!ctx->cstage[i]) {
            const hipError_t e = hipHostMalloc(&ctx->cstage[i], knox_ctx::kStageBytes, hipHostMallocDefault);
            if (e != hipSuccess) {
                ctx->last_hip = (int)e;
                ctx->cstage[i] = nullptr;
                return KNOX_E_NOMEM;
            }
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->cstage_ev[i], hipEventDisableTiming));
        }
    }
    for (size_t off = 0; off < bytes;) {
        const int slot = ctx->cstage_slot;
        const size_t n = std::min(bytes - off, knox_ctx::kStageBytes);
        if (ctx->cstage_used[slot]) HIPCHK(ctx, hipEventSynchronize(ctx->cstage_ev[slot]));  // the DMA that last read this slot
        CopyPool::get().copy(ctx->cstage[slot], (const char *)src + off, n);
        HIPCHK(ctx, hipMemcpyAsync((char *)dst + off, ctx->cstage[slot], n, hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->cstage_ev[slot], ctx->copy_stream));
        ctx->cstage_used[slot] = true;
        ctx->cstage_slot ^= 1;
        off += n;
    }
    return KNOX_OK;
}

KNOX_API int knox_col_create(knox_ctx *ctx, int dtype, size_t capacity_rows, size_t pack_rows, knox_col **out) {
    if (!ctx || !out || dtype == KNOX_BYTES || capacity_rows == 0) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = col_new(ctx, dtype, capacity_rows, pack_rows, true, out);
    if (rc) return rc;
    (*out)->rows_cap = capacity_rows;
    (*out)->n_rows = 0;
    return KNOX_OK;
}

KNOX_API int knox_col_append_async(knox_ctx *ctx, knox_col *col, const void *host, size_t n_rows) {
    if (!ctx || !col || col->ctx != ctx || col->dtype == KNOX_BYTES || !col->owned || col->rows_cap == 0 || (n_rows && !host)) return KNOX_E_INVALID;
    if (n_rows > col->rows_cap - col->n_rows) return KNOX_E_RANGE;
    if (n_rows == 0) return KNOX_OK;
    col->sorted_state = -1;  // new rows: the order check runs again
    col->fexp_max = -1;
    hipSetDevice(ctx->device);
    int rc = copy_stream_init(ctx);
    if (rc) return rc;
    rc = upload_async(ctx, (char *)col->dev + col->n_rows * (size_t)col->width, host, n_rows * (size_t)col->width);
    if (rc) return rc;
    if (!col->ready_ev) HIPCHK(ctx, hipEventCreateWithFlags(&col->ready_ev, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(col->ready_ev, ctx->copy_stream));
    col->ready_pending = true;
    col->n_rows += n_rows;
    return KNOX_OK;
}

KNOX_API int knox_col_reset(knox_ctx *ctx, knox_col *col) {
    if (!ctx || !col || col->ctx != ctx || col->rows_cap == 0 || col->dtype == KNOX_BYTES) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = copy_stream_init(ctx);
    if (rc) return rc;
    // the next upload overwrites rows that scans enqueued so far may still be reading: the copy stream waits for them (and
    // for nothing enqueued later -- so refill a column BEFORE enqueueing the scan of the other one)
    HIPCHK(ctx, hipEventRecord(ctx->fence_ev, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->fence_ev, 0));
    col->n_rows = 0;
    col->sorted_state = -1;  // new content: what the order check found no longer holds
    col->fexp_max = -1;
    return KNOX_OK;
}

KNOX_API int knox_col_upload_async(knox_ctx *ctx, int dtype, const void *host, size_t n_rows, size_t pack_rows, knox_col **out) {
    if (!ctx || !out || !host || n_rows == 0 || dtype == KNOX_BYTES) return KNOX_E_INVALID;
    int rc = knox_col_create(ctx, dtype, n_rows, pack_rows, out);
    if (rc) return rc;
    rc = knox_col_append_async(ctx, *out, host, n_rows);
    if (rc) {
        knox_col_free(*out);
        *out = nullptr;
    }
    return rc;
}

KNOX_API int knox_col_wait(knox_ctx *ctx, knox_col *col) {
    if (!ctx || !col) return KNOX_E_INVALID;
    if (col->ready_pending) {
        HIPCHK(ctx, hipEventSynchronize(col->ready_ev));
        col->ready_pending = false;
    }
    return KNOX_OK;
}

KNOX_API int knox_col_ready(const knox_col *col) {
    if (!col) return KNOX_E_INVALID;
    if (!col->ready_pending) return 1;
    if (hipEventQuery(col->ready_ev) == hipSuccess) {
        col->ready_pending = false;
        return 1;
    }
    (void)hipGetLastError();
    return 0;
}

KNOX_API int knox_col_upload_bytes(knox_ctx *ctx, const uint64_t *ptr, size_t n_rows, const uint8_t *buf, size_t buf_len,
                                   size_t pack_rows, knox_col **out) {
    if (!ctx || !out || (n_rows && !ptr) || (buf_len && !buf)) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    // every (off, len) must lie inside buf: the device never reads out of bounds
    for (size_t i = 0; i < n_rows; i++) {
        const uint64_t off = ptr[i] >> 32, len = ptr[i] & 0xffffffffull;
        if (off + len > buf_len) return KNOX_E_RANGE;
    }
    int rc = col_new(ctx, KNOX_BYTES, n_rows, pack_rows, true, out);
    if (rc) return rc;
    knox_col *c = *out;
    hipError_t e = dev_malloc(ctx, (void **)&c->buf, buf_len + 256, ARENA_READ);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        knox_col_free(c);
        *out = nullptr;
        return KNOX_E_NOMEM;
    }
    c->buf_len = buf_len;
    rc = upload(ctx, c->dev, ptr, n_rows * 8);
    if (!rc) rc = upload(ctx, c->buf, buf, buf_len);
    if (rc) {
        knox_col_free(c);
        *out = nullptr;
    }
    return rc;
}

// ---- bytes columns with one StringPool per pack --------------------------------------------------------
static int bytes_col_alloc(knox_ctx *ctx, size_t pack_rows, size_t max_packs, size_t max_pool_bytes, knox_col **out) {
    if (!ctx || !out || pack_rows == 0 || (pack_rows & 63) || max_packs == 0 || max_packs > 0xffffffffull) return KNOX_E_INVALID;
    if (max_packs > ((size_t)1 << 40) / pack_rows) return KNOX_E_RANGE;
    hipSetDevice(ctx->device);
    const size_t rows_cap = pack_rows * max_packs;
    // the pointer array and the pools are read together by every bytes kernel: one placement class for both when the arena
    // can find the room (best effort: a failure only means they may end up in two classes)
    if (ctx->placement) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)ctx->arena.reserve(rows_cap * 8 + max_pool_bytes + 64 * max_packs + ((size_t)4 << 20));
    }
    int rc = col_new(ctx, KNOX_BYTES, rows_cap, pack_rows, true, out);
    if (rc) return rc;
    knox_col *c = *out;
    c->n_rows = 0;
    c->rows_cap = rows_cap;
    c->packs_cap = max_packs;
    c->buf_cap = max_pool_bytes + 64 * max_packs;  // every pool starts on a 64-byte boundary
    c->h_pool_base = (uint64_t *)calloc(max_packs + 1, 8);
    if (!c->h_pool_base || dev_malloc(ctx, (void **)&c->buf, c->buf_cap + 256, ARENA_READ) != hipSuccess ||
        dev_malloc(ctx, (void **)&c->pool_base, (max_packs + 1) * 8) != hipSuccess ||
        hipMemsetAsync(c->pool_base, 0, (max_packs + 1) * 8, ctx->stream) != hipSuccess) {
        knox_col_free(c);
        *out = nullptr;
        return KNOX_E_NOMEM;
    }
    return KNOX_OK;
}

KNOX_API int knox_col_bytes_create(knox_ctx *ctx, size_t pack_rows, size_t max_packs, size_t max_pool_bytes, knox_col **out) {
    return bytes_col_alloc(ctx, pack_rows, max_packs, max_pool_bytes, out);
}

KNOX_API int knox_col_bytes_append_pack(knox_ctx *ctx, knox_col *col, const uint64_t *ptr, size_t n_rows, const uint8_t *buf, size_t buf_len) {
    if (!ctx || !col || col->ctx != ctx || !col->h_pool_base || (n_rows && !ptr) || (buf_len && !buf)) return KNOX_E_INVALID;
    if (n_rows == 0 || n_rows > col->pack_rows || buf_len > 0xffffffffull) return KNOX_E_RANGE;  // StringPool offsets are 32 bit
    if (col->n_packs >= col->packs_cap || col->n_rows != col->n_packs * col->pack_rows) return KNOX_E_RANGE;  // only the last pack may be short
    const uint64_t base = col->h_pool_base[col->n_packs];
    if (base + buf_len > col->buf_cap) return KNOX_E_RANGE;
    for (size_t i = 0; i < n_rows; i++) {  // every (off, len) must lie inside the pack's pool: the device never reads out of bounds
        const uint64_t off = ptr[i] >> 32, len = ptr[i] & 0xffffffffull;
        if (off + len > buf_len) return KNOX_E_RANGE;
    }
    hipSetDevice(ctx->device);
    int rc = upload(ctx, (uint64_t *)col->dev + col->n_rows, ptr, n_rows * 8);
    if (!rc) rc = upload(ctx, col->buf + base, buf, buf_len);
    if (rc) return rc;
    const uint64_t next = (base + buf_len + 63) & ~(uint64_t)63;
    col->h_pool_base[col->n_packs + 1] = next;
    HIPCHK(ctx, hipMemcpyAsync(col->pool_base + col->n_packs + 1, &col->h_pool_base[col->n_packs + 1], 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    col->n_packs++;
    col->n_rows += n_rows;
    col->buf_len = next;
    return KNOX_OK;
}

KNOX_API int knox_col_generate_bytes(knox_ctx *ctx, uint64_t seed, size_t n_rows, size_t pack_rows, uint32_t key_len, uint32_t window_bits,
                                     uint32_t packs_per_window, knox_col **out) {
    if (!ctx || !out || n_rows == 0 || key_len == 0 || key_len > 256 || packs_per_window == 0 || window_bits > 40) return KNOX_E_INVALID;
    if (pack_rows == 0 || (pack_rows & 63) || pack_rows * (uint64_t)key_len > 0xffffffffull) return KNOX_E_RANGE;
    const size_t n_packs = (n_rows + pack_rows - 1) / pack_rows;
    const uint64_t stride = (pack_rows * (uint64_t)key_len + 63) & ~(uint64_t)63;
    int rc = bytes_col_alloc(ctx, pack_rows, n_packs, (size_t)(stride * n_packs), out);
    if (rc) return rc;
    knox_col *c = *out;
    for (size_t p = 0; p <= n_packs; p++) c->h_pool_base[p] = stride * p;
    hipError_t e = hipMemcpyAsync(c->pool_base, c->h_pool_base, (n_packs + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = launch_generate_bytes(ctx->stream, seed, 0, n_rows, pack_rows, key_len, window_bits, packs_per_window, stride, (uint64_t *)c->dev, c->buf);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        knox_col_free(c);
        *out = nullptr;
        return KNOX_E_HIP;
    }
    c->n_packs = n_packs;
    c->n_rows = n_rows;
    c->buf_len = stride * n_packs;
    return KNOX_OK;
}

KNOX_API size_t knox_col_pool_bytes(const knox_col *col) { return col && col->dtype == KNOX_BYTES ? col->buf_len : 0; }

// pack `pack` of a bytes column as the reference holds it: its row pointers (off << 32 | len, relative to the pack's
// pool) and the pool itself; a single-pool column is "pack 0"
KNOX_API int knox_col_bytes_download_pack(knox_ctx *ctx, const knox_col *col, size_t pack, uint64_t *ptr_out, size_t ptr_cap, size_t *n_rows_out,
                                          uint8_t *buf_out, size_t buf_cap, size_t *buf_len_out) {
    if (!ctx || !col || col->dtype != KNOX_BYTES) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    size_t r0 = 0, rows = col->n_rows;
    uint64_t b0 = 0, blen = col->buf_len;
    if (col->pool_base) {
        if (pack >= col->n_packs) return KNOX_E_RANGE;
        r0 = pack * col->pack_rows;
        rows = std::min(col->pack_rows, col->n_rows - r0);
        b0 = col->h_pool_base[pack];
        blen = col->h_pool_base[pack + 1] - b0;
    } else if (pack != 0) {
        return KNOX_E_RANGE;
    }
    if (n_rows_out) *n_rows_out = rows;
    if (buf_len_out) *buf_len_out = (size_t)blen;
    int rc = KNOX_OK;
    if (ptr_out) {
        if (ptr_cap < rows) return KNOX_E_RANGE;
        rc = download(ctx, ptr_out, (const uint64_t *)col->dev + r0, rows * 8);
    }
    if (!rc && buf_out) {
        if (buf_cap < blen) return KNOX_E_RANGE;
        rc = download(ctx, buf_out, col->buf + b0, (size_t)blen);
    }
    return rc;
}

KNOX_API int knox_col_wrap(knox_ctx *ctx, int dtype, void *device_ptr, size_t n_rows, size_t pack_rows, knox_col **out) {
    if (!ctx || !out || (n_rows && !device_ptr) || dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (((uintptr_t)device_ptr & 15) != 0) return KNOX_E_INVALID;
    int rc = col_new(ctx, dtype, n_rows, pack_rows, false, out);
    if (rc) return rc;
    (*out)->dev = device_ptr;
    return KNOX_OK;
}

KNOX_API int knox_col_generate(knox_ctx *ctx, int dtype, int kind, uint64_t seed, uint64_t first_row, size_t n_rows, int64_t lo,
                               uint64_t span, size_t pack_rows, knox_col **out) {
    if (!ctx || !out || dtype == KNOX_BYTES || kind < 0 || kind > 4) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = col_new(ctx, dtype, n_rows, pack_rows, true, out);
    if (rc) return rc;
    hipError_t e = launch_generate(ctx->stream, (*out)->width, dtype == KNOX_FLOAT32, kind, seed, first_row, n_rows, lo, span, (*out)->dev);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        knox_col_free(*out);
        *out = nullptr;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

KNOX_API int knox_col_download(knox_ctx *ctx, const knox_col *col, size_t first_row, size_t n_rows, void *host) {
    if (!ctx || !col || (n_rows && !host)) return KNOX_E_INVALID;
    if (first_row + n_rows > col->n_rows) return KNOX_E_RANGE;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, col); if (rc_) return rc_; }
    return download(ctx, host, (const char *)col->dev + first_row * (size_t)col->width, n_rows * (size_t)col->width);
}

KNOX_API int knox_col_free(knox_col *col) {
    if (!col) return KNOX_E_INVALID;
    hipSetDevice(col->ctx->device);
    if (col->ready_pending) hipEventSynchronize(col->ready_ev);
    if (col->ready_ev) hipEventDestroy(col->ready_ev);
    hipStreamSynchronize(col->ctx->stream);
    if (col->owned) dev_free(col->ctx, col->dev);
    dev_free(col->ctx, col->buf);
    dev_free(col->ctx, col->pool_base);
    free(col->h_pool_base);
    delete col;
    return KNOX_OK;
}

KNOX_API int knox_col_placement_class(const knox_col *col, int which) {
    if (!col || !col->ctx) return -1;
    const void *p = which == 1 ? (const void *)col->buf : (const void *)col->dev;
    return p ? col->ctx->arena.class_of(p) : -1;
}

KNOX_API size_t knox_col_rows(const knox_col *col) { return col ? col->n_rows : 0; }
KNOX_API int knox_col_dtype(const knox_col *col) { return col ? col->dtype : 0; }
KNOX_API void *knox_col_device_ptr(const knox_col *col) { return col ? col->dev : nullptr; }

// ------------------------------------------------------------------------------------------------
// device bitsets
// ------------------------------------------------------------------------------------------------
// the stream a write buffer is verified against: the column's values (bytes columns: the pool, the larger stream)
static void col_stream(const knox_col *col, const void **p, size_t *bytes) {
    const bool pool = col->dtype == KNOX_BYTES && col->buf_len >= col->n_rows * 8;
    *p = pool ? (const void *)col->buf : (const void *)col->dev;
    *bytes = pool ? col->buf_len : col->n_rows * (size_t)col->width;
}

int ensure_for(knox_ctx *ctx, DevBuf &b, size_t bytes, const knox_col *col) {
    const void *cp = nullptr;
    size_t cb = 0;
    if (col && col->n_rows) col_stream(col, &cp, &cb);
    if (!ctx->placement || !cp || cb < ((size_t)1 << 30) || bytes < ((size_t)16 << 20)) return ensure(ctx, b, bytes, ARENA_WRITE);
    if (b.cap >= bytes && b.for_col == cp) return KNOX_OK;
    // A workload that alternates between columns keeps its verified buffers: the one in use is parked (keyed by the
    // column region it was verified against -- a verdict is about physical memory, so it survives the column object) and a
    // parked one that fits comes back without a single probe launch.  Two are kept; the oldest goes when a third arrives.
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the stream may still be using b
    if (&b == &ctx->tmp_bits) {
        for (DevBuf &pk : ctx->tmp_bits_parked) {
            if (pk.p && pk.for_col == cp && pk.cap >= bytes) {
                std::swap(b, pk);
                return KNOX_OK;
            }
        }
        if (b.p && b.for_col) {
            DevBuf *slot = nullptr;
            for (DevBuf &pk : ctx->tmp_bits_parked)
                if (!pk.p) { slot = &pk; break; }
            if (!slot) {
                slot = &ctx->tmp_bits_parked[ctx->tmp_bits_park_next];
                ctx->tmp_bits_park_next = (ctx->tmp_bits_park_next + 1) % (int)(sizeof ctx->tmp_bits_parked / sizeof ctx->tmp_bits_parked[0]);
                dev_free(ctx, slot->p);
            }
            *slot = b;
            b = DevBuf();
        }
    }
    if (b.p) {
        dev_free(ctx, b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t cap = (bytes + 255) & ~(size_t)255;
    hipError_t e = ctx->arena.alloc_write_for(&b.p, cap, cp, cb, (double)cb / (double)col->n_rows, nullptr, nullptr);
    if (e == hipErrorNotSupported) return ensure(ctx, b, bytes, ARENA_WRITE);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        b.p = nullptr;
        return KNOX_E_NOMEM;
    }
    b.cap = cap;
    b.for_col = cp;
    return KNOX_OK;
}

static int bits_alloc_cols(knox_ctx *ctx, size_t size_bits, const knox_col *const *for_cols, int n_cols, knox_bits **out);
int bits_alloc(knox_ctx *ctx, size_t size_bits, const knox_col *for_col, knox_bits **out) { return bits_alloc_cols(ctx, size_bits, &for_col, for_col ? 1 : 0, out); }

static int bits_alloc_cols(knox_ctx *ctx, size_t size_bits, const knox_col *const *for_cols, int n_cols, knox_bits **out) {
    if (!ctx || !out) return KNOX_E_INVALID;
    *out = nullptr;
    hipSetDevice(ctx->device);
    knox_bits *b = new (std::nothrow) knox_bits();
    if (!b) return KNOX_E_NOMEM;
    b->ctx = ctx;
    b->size = size_bits;
    b->cap_bytes = bits_padded(size_bits) + 64;
    hipError_t e = hipErrorNotSupported;
    if (n_cols > 0 && ctx->placement) {
        // a write buffer of its own, verified against the column(s) themselves (knox_arena.h)
        (void)hipStreamSynchronize(ctx->stream);
        Arena::ColRef refs[kMaxLeaves];
        int n = 0;
        for (int i = 0; i < n_cols && n < kMaxLeaves; i++) {
            if (!for_cols[i] || !for_cols[i]->n_rows) continue;
            const void *cp;
            size_t cb;
            col_stream(for_cols[i], &cp, &cb);
            refs[n++] = Arena::ColRef{cp, cb, (double)cb / (double)for_cols[i]->n_rows};
        }
        if (n) e = ctx->arena.alloc_write_for((void **)&b->dev, b->cap_bytes, refs, n, &b->pair_ratio, &b->pair_tried);
    }
    if (e == hipErrorNotSupported) e = dev_malloc(ctx, (void **)&b->dev, b->cap_bytes, ARENA_WRITE);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        delete b;
        return KNOX_E_NOMEM;
    }
    e = launch_zero_bytes(ctx->stream, b->dev, b->cap_bytes);   // (the runtime's fill kernel behind hipMemsetAsync writes ~2 TB/s and takes ~11 us for a few bytes)
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        dev_free(ctx, b->dev);
        delete b;
        return KNOX_E_HIP;
    }
    *out = b;
    return KNOX_OK;
}

KNOX_API int knox_bits_alloc(knox_ctx *ctx, size_t size_bits, knox_bits **out) { return bits_alloc(ctx, size_bits, nullptr, out); }

KNOX_API int knox_bits_alloc_for(knox_ctx *ctx, size_t size_bits, const knox_col *col, knox_bits **out) {
    if (!col || col->ctx != ctx) return KNOX_E_INVALID;
    { const int rc_ = col_acquire(ctx, col); if (rc_) return rc_; }  // the verification probes read the column
    return bits_alloc(ctx, size_bits, col, out);
}

KNOX_API int knox_bits_alloc_for_cols(knox_ctx *ctx, size_t size_bits, const knox_col *const *cols, int n_cols, knox_bits **out) {
    if (!cols || n_cols <= 0 || n_cols > kMaxLeaves) return KNOX_E_INVALID;
    for (int i = 0; i < n_cols; i++) {
        if (!cols[i] || cols[i]->ctx != ctx) return KNOX_E_INVALID;
        const int rc_ = col_acquire(ctx, cols[i]);
        if (rc_) return rc_;
    }
    return bits_alloc_cols(ctx, size_bits, cols, n_cols, out);
}

KNOX_API int knox_bits_pair_check(const knox_bits *b, double *scan_over_read_only, int *candidates) {
    if (!b) return KNOX_E_INVALID;
    if (scan_over_read_only) *scan_over_read_only = b->pair_ratio;
    if (candidates) *candidates = b->pair_tried;
    return KNOX_OK;
}

// Borrowed bitset memory (the counterpart of knox_col_wrap): bits_padded(size) + 64 bytes at device_ptr
KNOX_API int knox_bits_wrap(knox_ctx *ctx, void *device_ptr, size_t size_bits, knox_bits **out) {
    if (!ctx || !out || !device_ptr || ((uintptr_t)device_ptr & 127) != 0) return KNOX_E_INVALID;
    knox_bits *b = new (std::nothrow) knox_bits();
    if (!b) return KNOX_E_NOMEM;
    b->ctx = ctx;
    b->size = size_bits;
    b->dev = (uint64_t *)device_ptr;
    b->cap_bytes = bits_padded(size_bits) + 64;
    b->owned = false;
    *out = b;
    return KNOX_OK;
}

KNOX_API int knox_bits_free(knox_bits *b) {
    if (!b) return KNOX_E_INVALID;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    if (b->dev && b->owned) dev_free(b->ctx, b->dev);
    delete b;
    return KNOX_OK;
}

KNOX_API size_t knox_bits_size(const knox_bits *b) { return b ? b->size : 0; }
KNOX_API void *knox_bits_device_ptr(const knox_bits *b) { return b ? (void *)b->dev : nullptr; }

KNOX_API int knox_bits_upload(knox_ctx *ctx, knox_bits *b, const uint8_t *host) {
    if (!ctx || !b || (b->size && !host)) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    HIPCHK(ctx, launch_zero_bytes(ctx->stream, b->dev, b->cap_bytes));
    return upload(ctx, b->dev, host, bits_bytes(b->size));
}

KNOX_API int knox_bits_download(knox_ctx *ctx, const knox_bits *b, uint8_t *host) {
    if (!ctx || !b || (b->size && !host)) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    return download(ctx, host, b->dev, bits_bytes(b->size));
}

// flag / count producing bitset calls write into a result slot: 16 device bytes -> the slot's pinned mirror + its event.
// The blocking entry points use the extra slot [kResRing] and wait; the *_async ones take the next ring slot (the FIFO
// knox_scan_async uses) and return.
static int slot_prepare(knox_ctx *ctx, int slot_idx, uint32_t **dev16) {
    int rc = ensure(ctx, ctx->ring_dev, 16 * (knox_ctx::kResRing + 1));
    knox_ctx::ResSlot &slot = ctx->res[slot_idx];
    if (!rc) rc = ensure_host(ctx, slot.h_res, slot.h_res_cap, 16);
    if (rc) return rc;
    if (!slot.done) HIPCHK(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    *dev16 = (uint32_t *)((char *)ctx->ring_dev.p + 16 * slot_idx);
    HIPCHK(ctx, launch_zero_bytes(ctx->stream, *dev16, 16));
    return KNOX_OK;
}

static int slot_finish(knox_ctx *ctx, int slot_idx, const uint32_t *dev16, int kind, size_t bits_size) {
    knox_ctx::ResSlot &slot = ctx->res[slot_idx];
    HIPCHK(ctx, hipMemcpyAsync(slot.h_res, dev16, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipEventRecord(slot.done, ctx->stream));
    slot.kind = kind;
    slot.bits_size = bits_size;
    return KNOX_OK;
}

static int bitop_into_slot(knox_ctx *ctx, int op, knox_bits *dst, const knox_bits *src, int slot_idx) {
    uint32_t *flags = nullptr;
    int rc = slot_prepare(ctx, slot_idx, &flags);
    if (rc) return rc;
    hipError_t e = launch_bitop(ctx->stream, op, dst->dev, src ? src->dev : nullptr, dst->size, flags);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return slot_finish(ctx, slot_idx, flags, 1, dst->size);
}

static int popcount_into_slot(knox_ctx *ctx, const knox_bits *b, int slot_idx) {
    uint32_t *d = nullptr;
    int rc = slot_prepare(ctx, slot_idx, &d);
    if (rc) return rc;
    hipError_t e = launch_popcount(ctx->stream, b->dev, b->size, (unsigned long long *)d);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return slot_finish(ctx, slot_idx, d, 2, b->size);
}

static int slot_read(knox_ctx *ctx, int slot_idx, int *any, int *all, int64_t *count) {
    knox_ctx::ResSlot &slot = ctx->res[slot_idx];
    HIPCHK(ctx, hipEventSynchronize(slot.done));
    if (slot.kind == 1) {
        const uint32_t *h = (const uint32_t *)slot.h_res;
        if (any) *any = h[0] != 0;
        if (all) *all = (slot.bits_size > 0) && (h[1] == 0);
        if (count) *count = -1;
    } else {
        if (count) *count = (int64_t) * (const unsigned long long *)slot.h_res;
        if (any) *any = -1;
        if (all) *all = -1;
    }
    return KNOX_OK;
}

static int bitop(knox_ctx *ctx, int op, knox_bits *dst, const knox_bits *src, int *any, int *all) {
    if (!ctx || !dst) return KNOX_E_INVALID;
    const bool binary = (op == BIT_AND || op == BIT_ANDNOT || op == BIT_OR || op == BIT_XOR || op == BIT_COPY);
    if (binary && (!src || src->size != dst->size)) return KNOX_E_INVALID;  // size mismatch is a no-op in Go (bitset.go:400-518); an error here
    hipSetDevice(ctx->device);
    if (any || all) {
        int rc = bitop_into_slot(ctx, op, dst, src, knox_ctx::kResRing);
        if (rc) return rc;
        return slot_read(ctx, knox_ctx::kResRing, any, all, nullptr);
    }
    hipError_t e = launch_bitop(ctx->stream, op, dst->dev, src ? src->dev : nullptr, dst->size, nullptr);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

static int bitop_async(knox_ctx *ctx, int op, knox_bits *dst, const knox_bits *src) {
    if (!ctx || !dst || !src || src->size != dst->size) return KNOX_E_INVALID;
    if (ctx->res_head - ctx->res_tail >= (uint64_t)knox_ctx::kResRing) return KNOX_E_BUSY;  // collect first
    hipSetDevice(ctx->device);
    int rc = bitop_into_slot(ctx, op, dst, src, (int)(ctx->res_head % knox_ctx::kResRing));
    if (rc) return rc;
    ctx->res_head++;
    return KNOX_OK;
}

KNOX_API int knox_bits_and_flag_async(knox_ctx *ctx, knox_bits *dst, const knox_bits *src) { return bitop_async(ctx, BIT_AND, dst, src); }
KNOX_API int knox_bits_or_flag_async(knox_ctx *ctx, knox_bits *dst, const knox_bits *src) { return bitop_async(ctx, BIT_OR, dst, src); }

KNOX_API int knox_bits_popcount_async(knox_ctx *ctx, const knox_bits *b) {
    if (!ctx || !b) return KNOX_E_INVALID;
    if (ctx->res_head - ctx->res_tail >= (uint64_t)knox_ctx::kResRing) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    int rc = popcount_into_slot(ctx, b, (int)(ctx->res_head % knox_ctx::kResRing));
    if (rc) return rc;
    ctx->res_head++;
    return KNOX_OK;
}

// the OLDEST uncollected result of the ring, which must be a bitset result (a scan is collected by knox_scan_collect)
KNOX_API int knox_bits_collect(knox_ctx *ctx, int *any, int *all, int64_t *count) {
    if (!ctx) return KNOX_E_INVALID;
    if (ctx->res_tail == ctx->res_head) return KNOX_E_INVALID;  // nothing in flight
    const int idx = (int)(ctx->res_tail % knox_ctx::kResRing);
    if (ctx->res[idx].kind == 0) return KNOX_E_INVALID;
    int rc = slot_read(ctx, idx, any, all, count);
    if (rc) return rc;
    ctx->res_tail++;
    return KNOX_OK;
}

KNOX_API int knox_bits_zero(knox_ctx *ctx, knox_bits *b) {
    if (!ctx || !b) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    HIPCHK(ctx, launch_zero_bytes(ctx->stream, b->dev, b->cap_bytes));
    return KNOX_OK;
}
KNOX_API int knox_bits_one(knox_ctx *ctx, knox_bits *b) { return bitop(ctx, BIT_ONE, b, nullptr, nullptr, nullptr); }
KNOX_API int knox_bits_copy(knox_ctx *ctx, knox_bits *dst, const knox_bits *src) { return bitop(ctx, BIT_COPY, dst, src, nullptr, nullptr); }
KNOX_API int knox_bits_and(knox_ctx *ctx, knox_bits *dst, const knox_bits *src) { return bitop(ctx, BIT_AND, dst, src, nullptr, nullptr); }
KNOX_API int knox_bits_andnot(knox_ctx *ctx, knox_bits *dst, const knox_bits *src) { return bitop(ctx, BIT_ANDNOT, dst, src, nullptr, nullptr); }
KNOX_API int knox_bits_or(knox_ctx *ctx, knox_bits *dst, const knox_bits *src) { return bitop(ctx, BIT_OR, dst, src, nullptr, nullptr); }
KNOX_API int knox_bits_xor(knox_ctx *ctx, knox_bits *dst, const knox_bits *src) { return bitop(ctx, BIT_XOR, dst, src, nullptr, nullptr); }
KNOX_API int knox_bits_neg(knox_ctx *ctx, knox_bits *b) { return bitop(ctx, BIT_NEG, b, nullptr, nullptr, nullptr); }
KNOX_API int knox_bits_and_flag(knox_ctx *ctx, knox_bits *dst, const knox_bits *src, int *any, int *all) {
    int a = 0, l = 0;
    int rc = bitop(ctx, BIT_AND, dst, src, &a, &l);
    if (any) *any = a;
    if (all) *all = l;
    return rc;
}
KNOX_API int knox_bits_or_flag(knox_ctx *ctx, knox_bits *dst, const knox_bits *src, int *any, int *all) {
    int a = 0, l = 0;
    int rc = bitop(ctx, BIT_OR, dst, src, &a, &l);
    if (any) *any = a;
    if (all) *all = l;
    return rc;
}

KNOX_API int knox_bits_match_bool(knox_ctx *ctx, const knox_bits *column, int op, int a, int b, knox_bits *bits) {
    if (!ctx || !column || !bits || column->size != bits->size) return KNOX_E_INVALID;
    auto eq = [&](bool val) {  // Bitset.MatchEqual: bits.Copy(s); if !val { bits.Neg() }
        int rc = bitop(ctx, BIT_COPY, bits, column, nullptr, nullptr);
        if (!rc && !val) rc = bitop(ctx, BIT_NEG, bits, nullptr, nullptr, nullptr);
        return rc;
    };
    auto one = [&]() { return bitop(ctx, BIT_ONE, bits, nullptr, nullptr, nullptr); };
    const bool va = a != 0, vb = b != 0;
    switch (op) {
    case KNOX_EQ: return eq(va);
    case KNOX_NE: return eq(!va);
    case KNOX_LT: return va ? eq(false) : KNOX_OK;
    case KNOX_LE: return va ? one() : eq(false);
    case KNOX_GT: return !va ? eq(true) : KNOX_OK;
    case KNOX_GE: return !va ? one() : eq(true);
    case KNOX_RANGE:
        if (va && vb) return eq(true);
        if (!va && vb) return one();
        if (!va && !vb) return eq(false);
        return KNOX_OK;
    default: return KNOX_E_UNSUPPORTED;
    }
}

// Int128 / Int256 matchers over limb columns (cmp_i128_* / cmp_i256_*)
KNOX_API int knox_match_wide(knox_ctx *ctx, knox_col *const *limbs, int n_limbs, int op, const uint64_t *a, const uint64_t *b,
                             const knox_bits *mask, knox_bits *bits, int64_t *count) {
    if (!ctx || !limbs || !a || !bits || (n_limbs != 2 && n_limbs != 4)) return KNOX_E_INVALID;
    if (op == KNOX_RANGE && !b) return KNOX_E_INVALID;
    if (!(op == KNOX_EQ || op == KNOX_NE || op == KNOX_LT || op == KNOX_LE || op == KNOX_GT || op == KNOX_GE || op == KNOX_RANGE)) return KNOX_E_UNSUPPORTED;
    const size_t n_rows = limbs[0] ? limbs[0]->n_rows : 0;
    for (int l = 0; l < n_limbs; l++) {
        // X0 is []int64, X1.. are []uint64 (num.Int128Stride); any 8-byte integer column is accepted
        if (!limbs[l] || limbs[l]->ctx != ctx || limbs[l]->n_rows != n_rows || limbs[l]->width != 8 || dtype_is_float(limbs[l]->dtype)) return KNOX_E_INVALID;
    }
    if (bits->size != n_rows || (mask && mask->size != n_rows)) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    for (int i_ = 0; i_ < n_limbs; i_++) { const int rc_ = col_acquire(ctx, limbs[i_]); if (rc_) return rc_; }
    int rc = ensure(ctx, ctx->misc, 256);
    if (rc) return rc;
    unsigned long long *d_cnt = (unsigned long long *)((char *)ctx->misc.p + 64);
    HIPCHK(ctx, launch_zero_bytes(ctx->stream, d_cnt, 8));
    WideArgs w;
    memset(&w, 0, sizeof w);
    for (int l = 0; l < n_limbs; l++) {
        w.limb[l] = (const uint64_t *)limbs[l]->dev;
        w.a[l] = a[l];
        w.b[l] = b ? b[l] : a[l];
    }
    w.a[0] ^= 1ull << 63;  // signed top limb -> unsigned order
    w.b[0] ^= 1ull << 63;
    w.op = (uint32_t)op;
    w.mask = mask ? (const uint32_t *)mask->dev : nullptr;
    w.bits = (uint32_t *)bits->dev;
    w.n_tiles = n_rows / (uint64_t)tile_rows(8);
    w.n_rows = n_rows;
    w.count = d_cnt;
    const int grid = (int)std::min<uint64_t>(std::max<uint64_t>(w.n_tiles, 1), (uint64_t)ctx->n_cus * 4);
    hipError_t e = launch_wide_cmp(ctx->stream, n_limbs, grid, w);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    if (count) {
        unsigned long long h = 0;
        HIPCHK(ctx, hipMemcpyAsync(&h, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *count = (int64_t)h;
    }
    return KNOX_OK;
}

// Int128 / Int256 IN / NOT IN
KNOX_API int knox_match_wide_set(knox_ctx *ctx, knox_col *const *limbs, int n_limbs, int negate, const uint64_t *set, size_t n_set,
                                 const knox_bits *mask, knox_bits *bits) {
    if (!ctx || !limbs || !bits || (n_limbs != 2 && n_limbs != 4) || (n_set && !set) || n_set > 0xffffffffull) return KNOX_E_INVALID;
    const size_t n_rows = limbs[0] ? limbs[0]->n_rows : 0;
    for (int l = 0; l < n_limbs; l++)
        if (!limbs[l] || limbs[l]->ctx != ctx || limbs[l]->n_rows != n_rows || limbs[l]->width != 8 || dtype_is_float(limbs[l]->dtype)) return KNOX_E_INVALID;
    if (bits->size != n_rows || (mask && mask->size != n_rows)) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    for (int i_ = 0; i_ < n_limbs; i_++) { const int rc_ = col_acquire(ctx, limbs[i_]); if (rc_) return rc_; }
    // sorted unique keys in the kernels' order: unsigned lexicographic over the limbs LEAST significant first (membership needs no value
    // order, and the low limb is the one that tells small values apart: knox_bitset.hip); the limb columns are passed in that order too
    const size_t L = (size_t)n_limbs;
    std::vector<size_t> idx(n_set);
    for (size_t i = 0; i < n_set; i++) idx[i] = i;
    auto key = [&](size_t i, size_t l) { return set[i * L + (L - 1 - l)]; };
    auto less = [&](size_t x, size_t y) {
        for (size_t l = 0; l < L; l++)
            if (key(x, l) != key(y, l)) return key(x, l) < key(y, l);
        return false;
    };
    std::sort(idx.begin(), idx.end(), less);
    idx.erase(std::unique(idx.begin(), idx.end(), [&](size_t x, size_t y) { return !less(x, y) && !less(y, x); }), idx.end());
    std::vector<uint64_t> keys(idx.size() * L);
    for (size_t k = 0; k < idx.size(); k++)
        for (size_t l = 0; l < L; l++) keys[k * L + l] = key(idx[k], l);
    int rc = ensure(ctx, ctx->bytes_val, keys.size() * 8 + 64);
    if (rc) return rc;
    if (!keys.empty()) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vector not retained
    }
    const uint64_t *lp[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int l = 0; l < n_limbs; l++) lp[l] = (const uint64_t *)limbs[n_limbs - 1 - l]->dev;
    hipError_t e = launch_wide_inset(ctx->stream, n_limbs, negate != 0, lp, n_rows, (const uint64_t *)ctx->bytes_val.p, (uint32_t)idx.size(),
                                     (uint8_t *)bits->dev, mask ? (const uint8_t *)mask->dev : nullptr);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    return KNOX_OK;
}

KNOX_API int knox_bits_popcount(knox_ctx *ctx, const knox_bits *b, int64_t *count) {
    if (!ctx || !b || !count) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = popcount_into_slot(ctx, b, knox_ctx::kResRing);
    if (rc) return rc;
    return slot_read(ctx, knox_ctx::kResRing, nullptr, nullptr, count);
}

KNOX_API int knox_bits_popcount_segments(knox_ctx *ctx, const knox_bits *b, size_t seg_bits, int64_t *counts, size_t n_counts) {
    if (!ctx || !b || !counts || seg_bits == 0 || (seg_bits & 63)) return KNOX_E_INVALID;
    const size_t n_seg = (b->size + seg_bits - 1) / seg_bits;
    if (n_counts < n_seg) return KNOX_E_RANGE;
    hipSetDevice(ctx->device);
    int rc = ensure(ctx, ctx->pack_counts, n_seg * 8 + 8);
    if (rc) return rc;
    hipError_t e = launch_popcount_segments(ctx->stream, b->dev, b->size, seg_bits, (int64_t *)ctx->pack_counts.p, n_seg);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return download(ctx, counts, ctx->pack_counts.p, n_seg * 8);
}

// row ids of the set bits into ctx->idx_rows (device); *n = how many there are
static int indexes_dev(knox_ctx *ctx, const knox_bits *b, size_t capacity, size_t *n) {
    if (b->size > 0xffffffffull) return KNOX_E_RANGE;  // uint32 row ids (pack-local in the reference)
    hipSetDevice(ctx->device);
    const size_t nwords = (b->size + 63) >> 6;
    const size_t nblocks = (nwords + 127) / 128;
    int rc = ensure(ctx, ctx->idx_scratch, (nblocks + 2) * 4);
    if (!rc) rc = ensure(ctx, ctx->idx_rows, std::max<size_t>(capacity, 1) * 4);
    if (!rc) rc = ensure(ctx, ctx->misc, 256);
    if (rc) return rc;
    unsigned long long *d_n = (unsigned long long *)((char *)ctx->misc.p + 128);
    hipError_t e = launch_indexes(ctx->stream, b->dev, b->size, (uint32_t *)ctx->idx_scratch.p, (uint32_t *)ctx->idx_rows.p, capacity, d_n);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    unsigned long long h = 0;
    HIPCHK(ctx, hipMemcpyAsync(&h, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n = (size_t)h;
    return KNOX_OK;
}

KNOX_API int knox_bits_indexes(knox_ctx *ctx, const knox_bits *b, uint32_t *rows, size_t capacity, size_t *n_out) {
    if (!ctx || !b || !n_out || (capacity && !rows)) return KNOX_E_INVALID;
    int rc = indexes_dev(ctx, b, capacity, n_out);
    if (rc) return rc;
    if (*n_out > capacity) return KNOX_E_RANGE;
    return download(ctx, rows, ctx->idx_rows.p, *n_out * 4);
}

// The same into the context's page-locked result buffer: no capacity to guess, no copy into pageable memory (whose
// first touch costs more than the transfer).  *rows_view stays valid until the next *_view / *_values call on the ctx.
KNOX_API int knox_bits_indexes_view(knox_ctx *ctx, const knox_bits *b, const uint32_t **rows_view, size_t *n_out) {
    if (!ctx || !b || !n_out || !rows_view) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    // the count first (one streaming pass over the bitset), so the row buffers are sized exactly
    int64_t cnt = 0;
    int rc = knox_bits_popcount(ctx, b, &cnt);
    if (rc) return rc;
    rc = indexes_dev(ctx, b, (size_t)cnt, n_out);
    if (!rc) rc = ensure_host(ctx, ctx->h_view, ctx->h_view_cap, std::max<size_t>(*n_out * 4, 64));
    if (rc) return rc;
    if (*n_out) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_view, ctx->idx_rows.p, *n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *rows_view = (const uint32_t *)ctx->h_view;
    return KNOX_OK;
}

// ------------------------------------------------------------------------------------------------
// predicate planning: (dtype, mode, a, b) -> key-space leaf + negation
// ------------------------------------------------------------------------------------------------

int make_leaf(int dtype, int op, uint64_t a, uint64_t b, LeafHost &out) {
    const int w = dtype_width(dtype);
    if (w == 0) return KNOX_E_INVALID;
    const uint64_t m = wmask(w);
    a &= m;
    b &= m;
    out = LeafHost();
    auto full = [&](bool negate) { out.lo = 0; out.diff = m; out.neg = negate; };
    if (op == KNOX_TRUE) { full(false); out.is_float = false; return KNOX_OK; }
    if (op == KNOX_FALSE) { full(true); out.is_float = false; return KNOX_OK; }
    if (!dtype_is_float(dtype)) {
        const uint64_t MIN = dtype_is_signed(dtype) ? (1ull << (8 * w - 1)) : 0;
        const uint64_t MAX = dtype_is_signed(dtype) ? (MIN - 1) & m : m;
        switch (op) {
        case KNOX_EQ: out.lo = a; out.diff = 0; break;
        case KNOX_NE: out.lo = a; out.diff = 0; out.neg = true; break;
        case KNOX_LE: out.lo = MIN; out.diff = (a - MIN) & m; break;
        case KNOX_GT: out.lo = MIN; out.diff = (a - MIN) & m; out.neg = true; break;
        case KNOX_GE: out.lo = a; out.diff = (MAX - a) & m; break;
        case KNOX_LT: out.lo = a; out.diff = (MAX - a) & m; out.neg = true; break;
        case KNOX_RANGE: out.lo = a; out.diff = (b - a) & m; break;  // wrap-around set when a > b, number.go:211-243
        default: return KNOX_E_UNSUPPORTED;
        }
        return KNOX_OK;
    }
    out.is_float = true;
    const uint64_t sign = 1ull << (8 * w - 1);
    const uint64_t expm = w == 8 ? 0x7ff0000000000000ull : 0x7f800000ull;
    auto isnan_ = [&](uint64_t u) { return (u & (m ^ sign)) > expm; };
    auto iszero = [&](uint64_t u) { return (u & (m ^ sign)) == 0; };
    auto klo = [&](uint64_t u) { return iszero(u) ? host_fkey(sign, w) : host_fkey(u, w); };  // -0 is the lower key
    auto khi = [&](uint64_t u) { return iszero(u) ? host_fkey(0, w) : host_fkey(u, w); };
    const uint64_t KMIN = host_fkey(sign | expm, w), KMAX = host_fkey(expm, w);  // -Inf, +Inf
    switch (op) {
    case KNOX_EQ:
    case KNOX_NE:
        if (isnan_(a)) full(true);
        else { out.lo = klo(a); out.diff = khi(a) - klo(a); }
        if (op == KNOX_NE) out.neg = !out.neg;  // the only op true on NaN (float.go:46-77)
        break;
    case KNOX_LT:
        if (isnan_(a) || klo(a) == KMIN) full(true);
        else { out.lo = KMIN; out.diff = klo(a) - 1 - KMIN; }
        break;
    case KNOX_LE:
        if (isnan_(a)) full(true);
        else { out.lo = KMIN; out.diff = khi(a) - KMIN; }
        break;
    case KNOX_GT:
        if (isnan_(a) || khi(a) == KMAX) full(true);
        else { out.lo = khi(a) + 1; out.diff = KMAX - out.lo; }
        break;
    case KNOX_GE:
        if (isnan_(a)) full(true);
        else { out.lo = klo(a); out.diff = KMAX - out.lo; }
        break;
    case KNOX_RANGE:
        if (isnan_(a) || isnan_(b) || klo(a) > khi(b)) full(true);  // a <= v && v <= b, float.go:211-242
        else { out.lo = klo(a); out.diff = khi(b) - klo(a); }
        break;
    default: return KNOX_E_UNSUPPORTED;
    }
    return KNOX_OK;
}

// boolean function of the leaf hits -> truth table (bit idx = value for hits == idx)
static int make_truth_table(const knox_pred *nodes, int n_nodes, const std::vector<int> &leaf_of_node, const std::vector<bool> &leaf_neg,
                            int n_leaves, uint64_t &tt) {
    tt = 0;
    for (uint32_t idx = 0; idx < (1u << n_leaves); idx++) {
        bool st[64];
        int sp = 0;
        for (int i = 0; i < n_nodes; i++) {
            const knox_pred &p = nodes[i];
            if (p.kind == KNOX_NODE_LEAF) {
                if (sp >= 64) return KNOX_E_INVALID;
                const int l = leaf_of_node[i];
                st[sp++] = (((idx >> l) & 1) != 0) != leaf_neg[l];
            } else {
                const int k = p.nchild;
                if (k < 0 || k > sp) return KNOX_E_INVALID;
                bool v = (p.kind == KNOX_NODE_AND);  // empty AND = true, empty OR = false (match_core.go:45-56,131-138)
                for (int j = 0; j < k; j++) {
                    const bool c = st[sp - 1 - j];
                    v = (p.kind == KNOX_NODE_AND) ? (v && c) : (v || c);
                }
                sp -= k;
                st[sp++] = v;
            }
        }
        if (sp != 1) return KNOX_E_INVALID;
        if (st[0]) tt |= 1ull << idx;
    }
    return KNOX_OK;
}

// which aggregate work a reducer needs from the kernels
constexpr unsigned A_NEED_MOMENTS = 0x100;  // host-only planning bit: var/std second pass wanted
static int agg_need(int func, int dtype, unsigned &need) {
    switch (func) {
    case KNOX_R_COUNT: break;
    case KNOX_R_SUM: need |= dtype_is_float(dtype) ? A_FSUM : A_ISUM; break;
    case KNOX_R_MEAN: need |= A_FSUM; break;
    case KNOX_R_VAR: case KNOX_R_STD: need |= A_FSUM | A_NEED_MOMENTS; break;  // + the second pass (run_moments)
    case KNOX_R_MIN: case KNOX_R_MAX: need |= A_MINMAX | (dtype_is_float(dtype) ? A_FIRSTLAST : 0u); break;
    case KNOX_R_FIRST: case KNOX_R_LAST: need |= A_FIRSTLAST; break;
    default: return KNOX_E_UNSUPPORTED;
    }
    return KNOX_OK;
}
static unsigned agg_class(unsigned need) {
    need &= ~A_NEED_MOMENTS;
    if (need == 0) return 0;
    if (need == A_ISUM) return A_ISUM;
    if ((need & A_ISUM) == 0) return A_FSUM | A_MINMAX | A_FIRSTLAST;
    return A_ALL;
}

uint32_t pack_shift_for(size_t pack_rows, int width) {
    if (pack_rows == 0 || (pack_rows & (pack_rows - 1))) return 255;
    const size_t wave_rows = (size_t)tile_rows(width) / kWavesPerBlock;
    if (pack_rows < wave_rows) return 255;
    uint32_t s = 0;
    while (((size_t)1 << s) < pack_rows) s++;
    return s;
}

// Wave iterations per tile: a wave stages the bitset words of `iters` consecutive iterations and
// writes them as one burst (whole 128-B lines).  Measured over 14 column placements x iters
// {1,2,4,8,16} x {2,3} workgroups/CU (tools/placement_matrix.py, profiles/r01_placement_matrix.txt):
// iters = 2 is the fastest everywhere (7.0 TB/s typical, 5.9 TB/s on the slow placements, vs
// 6.7 / 5.5 TB/s at iters = 1).  Short inputs keep tiles small so every CU gets work.
uint32_t pick_iters(knox_ctx *ctx, size_t n_rows, int width, size_t pack_rows_or_0, uint32_t dflt) {
    const char *env = getenv("KNOX_ITERS");
    uint32_t it = env && atoi(env) > 0 ? (uint32_t)atoi(env) : dflt;
    if (it > (uint32_t)kMaxIters) it = kMaxIters;
    const uint64_t base = (uint64_t)tile_rows(width);
    const uint64_t want_tiles = (uint64_t)ctx->n_cus * 8;
    while (it > 1 && n_rows / (base * it) < want_tiles) it >>= 1;
    if (pack_rows_or_0) while (it > 1 && (pack_rows_or_0 % (base * it)) != 0) it >>= 1;
    return it;
}


// device view of a bytes column (knox_kernels.h BytesView); with a skip plan only the surviving packs are walked
BytesView bytes_view(const knox_col *c, size_t n_rows, SkipPlan *skip) {
    BytesView v;
    memset(&v, 0, sizeof v);
    v.ptr = (const uint64_t *)c->dev;
    v.buf = c->buf;
    v.pool_base = c->pool_base;
    v.n_rows = n_rows;
    v.pack_rows = c->pack_rows;
    v.pack_shift = 255;
    if (c->pack_rows >= 64 && !(c->pack_rows & (c->pack_rows - 1))) {
        uint32_t sh = 6;
        while (((size_t)1 << sh) < c->pack_rows) sh++;
        v.pack_shift = sh;
    }
    if (skip && skip->d_live && c->pack_rows && (c->pack_rows & 63) == 0) {
        v.live = skip->d_live;
        v.n_live = skip->n_live;
        skip->used = true;
    }
    return v;
}

// single fused pass: leaves (same width) + optional aggregate column, into partial slot `slot`
static int run_fused(knox_ctx *ctx, int width, const LeafHost *lh, const knox_col *const *lcols, int n_leaves, uint64_t tt,
                     const knox_col *agg_col, unsigned agg, size_t n_rows, size_t pack_rows, uint64_t row_base,
                     SkipPlan *skip, uint32_t *bits, int slot, int grid, int dtype_for_state, uint32_t *per_pack_tiles = nullptr) {
    ScanArgs a;
    memset(&a, 0, sizeof a);
    if (per_pack_tiles) *per_pack_tiles = 0;
    for (int l = 0; l < n_leaves; l++) {
        a.leaf[l].col = lcols[l]->dev;
        a.leaf[l].lo = lh[l].lo;
        a.leaf[l].diff = lh[l].diff;
        a.leaf[l].is_float = lh[l].is_float ? 1 : 0;
    }
    a.tt = tt;
    bool sep = false;
    if (agg_col) {
        a.agg_col = agg_col->dev;
        a.agg_flavor = dtype_flavor(agg_col->dtype);
        sep = agg_col->dev != lcols[0]->dev;
    }
    const bool want_skip = skip && skip->d_skip && skip->d_live && pack_rows && !(pack_rows & (pack_rows - 1)) && n_rows % pack_rows == 0;
    // aggregate-heavy single-leaf scans (the software-pipelined instantiations) run 1.5-3.5 % faster with 4 wave iterations
    // per tile than with 2 (tools/sweep_narrow.py all, round 3: int64 ALL 6.48 -> 6.64, int32 ALL 6.33 -> 6.51, int8 ALL 5.30 ->
    // 5.65 TB/s together with the occupancy below); the plain filter (+ sum) scans keep 2
    const bool plain_scan = !agg_col || (agg & ~A_ISUM) == 0;
    // (round 3, with the bursts deferred: scans that also record per-tile counts for the per-pack counts run 1.5 % faster with 4
    //  iterations -- half as many tiles, half as many count reductions: DPP chain + 4 readlanes per tile -- 8-, 4- and 2-byte columns;
    //  without the counts 2 and 4 are equal, tools/sweep_iters.sh, tools/ab_kernels.py envrun:KNOX_ITERS=4; the count-only
    //  form -- BASELINE configs[3]'s uint64 EQ -- loses 1-2 % with 4 and keeps 2)
    const bool counted = per_pack_tiles != nullptr && bits != nullptr && width >= 2 && agg_col != nullptr && agg != 0;  // agg == 0: count only; int32 85.4 -> 87.0, int16 78.8 -> 80.9 % of peak, uint8 no difference
    a.iters = pick_iters(ctx, n_rows, width, want_skip ? pack_rows : 0, ((!plain_scan && n_leaves == 1) || counted) ? 4u : 2u);
    // packs must stay whole numbers of tiles for the per-tile counts (small packs: 4096 rows = one 2-iteration tile of an 8-byte column)
    while (per_pack_tiles != nullptr && pack_rows && a.iters > 1 && pack_rows % ((size_t)tile_rows(width) * a.iters) != 0) a.iters >>= 1;
    const uint64_t tr = (uint64_t)tile_rows(width) * a.iters;
    a.n_tiles = n_rows / tr;
    a.n_rows = n_rows;
    a.tail_first_row = a.n_tiles * tr;
    a.row_base = row_base;
    a.bits = bits;
    if (skip && skip->d_skip) {
        a.pack_skip = skip->d_skip;
        a.pack_shift = pack_shift_for(pack_rows, width);
        const bool pow2 = pack_rows && !(pack_rows & (pack_rows - 1));
        if (skip->d_live && pow2 && pack_rows >= tr && n_rows % pack_rows == 0) {
            // compact mode: walk only the surviving packs; zero the pruned ranges of the bitset
            uint32_t sh = 0;
            while (((uint64_t)tr << sh) < pack_rows) sh++;
            a.live_packs = skip->d_live;
            a.tpp_shift = sh;
            a.n_tiles = (uint64_t)skip->n_live << sh;
            skip->used = true;
            if (bits) {
                const hipError_t ze = launch_zero_pruned(ctx->stream, (uint8_t *)bits, skip->d_skip, (uint32_t)skip->n_packs, pack_rows, n_rows);
                if (ze != hipSuccess) { ctx->last_hip = (int)ze; return KNOX_E_HIP; }
            }
        }
    } else {
        a.pack_shift = 255;
    }
    // auto occupancy: the pure streaming passes run best at 2 workgroups per CU, VALU-heavier ones at 3
    // (profiles/r01_sweep_blocks_per_cu.txt): 8-byte / 4-byte integer streaming 2 WG/CU; float or min-max
    // aggregates, several leaves, 2-byte types 3; 1-byte types and float32 sums (VALU bound) 8
    // (round 3, tools/sweep_narrow.py with the compare-unit predicate of narrow_load: 2-byte filter / filter + sum scans run
    //  best at 2 like the wider ones -- 6.69 vs 6.32 TB/s at 3 --, 1-byte ones at 3 -- 6.0 vs 5.5-5.8 TB/s at 8)
    //  aggregate-heavy single-leaf scans, same sweep: 2 workgroups / CU for 8-, 4- and 1-byte columns (3 is the worst choice
    //  for int32: 6.00 against 6.33-6.51 TB/s at 2 and 4), 3 for 2-byte ones, 8 for float32 sums)
    //  The table itself is scan_shape (knox_kernels.h), shared with the kernel, whose parked LDS words assume exactly this many
    //  workgroups per CU.
    const bool heavy = !plain_scan;  // plain = filter or filter + wrapping sum
    const int per_cu = scan_shape(width, n_leaves, heavy, a.agg_flavor == FL_FLOAT && agg_col != nullptr,
                                  agg_col != nullptr && (agg & A_FSUM) != 0).per_cu;
    const int want = ctx->blocks_per_cu > 0 ? grid : ctx->n_cus * per_cu;
    const int g = (int)std::min<uint64_t>(a.n_tiles, (uint64_t)std::min(grid, want));
    // per-pack counts for free: when packs are whole numbers of tiles, nothing is pruned and nothing is
    // left for the tail kernel, every wave records the popcount of its bursts (one 128-byte line per 32 tiles)
    static const bool tile_counts_on = !(getenv("KNOX_TILE_COUNTS") && atoi(getenv("KNOX_TILE_COUNTS")) == 0);  // 0: diagnostics
#ifdef KNOX_DIAG
    const bool diag_no_counts = ctx->variant == 2 || ctx->variant == 5;
#else
    const bool diag_no_counts = false;
#endif
    if (per_pack_tiles && bits && !diag_no_counts && pack_rows && tile_counts_on && a.live_packs == nullptr && g > 0) {
        const uint64_t rounds = ((a.n_tiles + (uint64_t)g - 1) / (uint64_t)g + 31) & ~(uint64_t)31;
        if (a.tail_first_row == n_rows && n_rows % pack_rows == 0 && pack_rows % tr == 0 && pack_rows / tr <= 0xffffffffull &&
            rounds <= 0xffffffffull && ensure(ctx, ctx->tile_counts, (size_t)g * kWavesPerBlock * rounds * 4 + 256, ARENA_WRITE) == KNOX_OK) {
            a.tile_counts = (uint32_t *)ctx->tile_counts.p;
            a.tc_rounds = (uint32_t)rounds;
            *per_pack_tiles = (uint32_t)(pack_rows / tr);
            ctx->tc_grid = (uint32_t)g;
            ctx->tc_rounds = (uint32_t)rounds;
        }
    }
#ifdef KNOX_DIAG
    a.debug = ctx->variant == 2 ? 1u : ctx->variant == 3 ? 2u : ctx->variant == 4 ? 4u : ctx->variant == 5 ? 8u : 0u;  // 2 no store, 3 plain store, 4 write-through store
#endif
    a.partials = (AggPartial *)ctx->partials.p + (size_t)slot * (size_t)(grid + 1);
    a.flags |= ctx->defer_flags;  // KNOX_DEFER=0 when the context was opened (A/B): bitset bursts tile by tile; =2: deferred but not aligned in time
    prof_mark(ctx, 0);  // also the leaf passes of a multi-pass plan (slot 15): knox_prof_read then sums a scan's leaf kernels
    hipError_t e = launch_scan(ctx->stream, width, n_leaves, agg_col ? agg : 0, sep, g, a);
    prof_mark(ctx, 1);
    const bool has_tail = a.tail_first_row < a.n_rows;
    if (e == hipSuccess && has_tail) e = launch_scan_tail(ctx->stream, width, n_leaves, agg, g, a);
    char *res = ctx->res_dst ? ctx->res_dst : (char *)ctx->states.p;
    ctx->post_packs_done = false;
    if (e == hipSuccess && ctx->res_dst && ctx->post_packs && per_pack_tiles && *per_pack_tiles != 0) {
        e = launch_finalize_packs(ctx->stream, width, a.partials, g + (has_tail ? 1 : 0), agg_col ? agg_col->dev : nullptr, row_base, dtype_for_state,
                                  (AggStateDev *)res + slot, (int64_t *)((AggStateDev *)res + 16) + slot, a.tile_counts, ctx->tc_grid,
                                  ctx->tc_rounds, *per_pack_tiles, ctx->post_packs, (int64_t *)(res + ctx->post_packs_off));
        ctx->post_packs_done = e == hipSuccess;
    } else if (e == hipSuccess)
        e = launch_finalize(ctx->stream, width, a.partials, g + (has_tail ? 1 : 0), agg_col ? agg_col->dev : nullptr, row_base, dtype_for_state,
                            (AggStateDev *)res + slot, (int64_t *)((AggStateDev *)res + 16) + slot);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

int run_reduce(knox_ctx *ctx, const knox_col *col, const uint32_t *mask, unsigned agg, size_t n_rows, uint64_t row_base,
                      int slot, int grid) {
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.agg_col = col->dev;
    a.agg_flavor = dtype_flavor(col->dtype);
    a.iters = pick_iters(ctx, n_rows, col->width, 0);
    const uint64_t tr = (uint64_t)tile_rows(col->width) * a.iters;
    a.n_tiles = n_rows / tr;
    a.n_rows = n_rows;
    a.tail_first_row = a.n_tiles * tr;
    a.row_base = row_base;
    a.mask = mask;
    a.pack_shift = 255;
    a.partials = (AggPartial *)ctx->partials.p + (size_t)slot * (size_t)(grid + 1);
    const int want = ctx->blocks_per_cu > 0 ? grid : ctx->n_cus * (col->width == 1 || (col->width == 4 && dtype_is_float(col->dtype)) ? 8 : 3);
    const int g = (int)std::min<uint64_t>(a.n_tiles, (uint64_t)std::min(grid, want));
    hipError_t e = launch_reduce(ctx->stream, col->width, agg, g, a);
    if (e == hipSuccess)
        e = launch_finalize(ctx->stream, col->width, a.partials, g + 1, col->dev, row_base, col->dtype, (AggStateDev *)ctx->states.p + slot,
                            (int64_t *)((AggStateDev *)ctx->states.p + 16) + slot);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

// var/std second pass over the rows `mask` selects: centred on the mean of state `slot` (already
// finalized on the device, same stream), results into that state's moment fields
int run_moments(knox_ctx *ctx, const knox_col *col, const uint32_t *mask, size_t n_rows, int slot, int grid) {
    int rc = ensure(ctx, ctx->mom_partials, (size_t)(grid + 1) * 4 * sizeof(double));
    if (rc) return rc;
    AggStateDev *st = (AggStateDev *)ctx->states.p + slot;
    const int want = ctx->blocks_per_cu > 0 ? grid : ctx->n_cus * 4;
    hipError_t e = launch_moments(ctx->stream, col->width, dtype_flavor(col->dtype), std::min(grid, want), col->dev, mask, n_rows, st, 0.0,
                                  (double *)ctx->mom_partials.p, &st->m_shift, &st->flags);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

// one leaf -> its own bitset (multi-pass plan)

// Pruned packs (skip) cannot contain matches of the whole tree, so a leaf may leave them untouched (zero): bytes leaves
// walk only the surviving packs, numeric leaves go through the streaming kernel's compact mode where it applies; the
// caller clears the pruned ranges of the final result.
static int run_leaf_to_bits(knox_ctx *ctx, const knox_pred &p, const knox_col *col, size_t n_rows, uint64_t *bits, int grid,
                            SkipPlan *skip = nullptr) {
    if (col->dtype == KNOX_BYTES && (p.op == KNOX_IN || p.op == KNOX_NOTIN)) {
        // the set: val = packed list of [uint32 little-endian length][bytes]; sorted (bytes.Compare) + unique here
        if (p.val_len && !p.val) return KNOX_E_INVALID;
        std::vector<std::pair<const uint8_t *, uint32_t>> items;
        for (uint64_t off = 0; off < p.val_len;) {
            if (off + 4 > p.val_len) return KNOX_E_INVALID;
            uint32_t len;
            memcpy(&len, p.val + off, 4);
            off += 4;
            if (off + len > p.val_len) return KNOX_E_INVALID;
            items.push_back({p.val + off, len});
            off += len;
        }
        auto less = [](const std::pair<const uint8_t *, uint32_t> &x, const std::pair<const uint8_t *, uint32_t> &y) {
            const uint32_t c = std::min(x.second, y.second);
            const int r = c ? memcmp(x.first, y.first, c) : 0;
            return r != 0 ? r < 0 : x.second < y.second;
        };
        std::sort(items.begin(), items.end(), less);
        items.erase(std::unique(items.begin(), items.end(), [&](const auto &x, const auto &y) { return !less(x, y) && !less(y, x); }), items.end());
        if (items.size() > 0xffffffffull) return KNOX_E_RANGE;
        std::vector<uint64_t> sp(items.size());
        std::vector<uint8_t> pool;
        for (size_t i = 0; i < items.size(); i++) {
            if (pool.size() > 0xffffffffull) return KNOX_E_RANGE;
            sp[i] = ((uint64_t)pool.size() << 32) | items[i].second;
            pool.insert(pool.end(), items[i].first, items[i].first + items[i].second);
        }
        pool.resize(pool.size() + 8);  // 8-byte compare loads may touch the padding
        // the members' first 8 bytes as big-endian integers, zero padded: bytes.Compare order sorts by them first (knox_bytes.hip)
        std::vector<uint64_t> pre(items.size());
        for (size_t i = 0; i < items.size(); i++) {
            uint64_t be = 0;
            for (uint32_t k = 0; k < 8; k++) be = (be << 8) | (k < items[i].second ? items[i].first[k] : 0);
            pre[i] = be;
        }
        const size_t pre_off = (sp.size() * 8 + 63) & ~(size_t)63;
        const size_t pool_off = (pre_off + pre.size() * 8 + 63) & ~(size_t)63;
        int rc = ensure(ctx, ctx->bytes_val, pool_off + pool.size() + 64);
        if (rc) return rc;
        if (!sp.empty()) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, sp.data(), sp.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync((char *)ctx->bytes_val.p + pre_off, pre.data(), pre.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        HIPCHK(ctx, hipMemcpyAsync((char *)ctx->bytes_val.p + pool_off, pool.data(), pool.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vectors not retained
        hipError_t e = launch_bytes_inset(ctx->stream, p.op == KNOX_NOTIN, bytes_view(col, n_rows, skip), (const uint64_t *)ctx->bytes_val.p,
                                          (uint32_t)sp.size(), (const uint8_t *)ctx->bytes_val.p + pool_off,
                                          (const uint64_t *)((const char *)ctx->bytes_val.p + pre_off), (uint8_t *)bits, nullptr);
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            return KNOX_E_HIP;
        }
        return KNOX_OK;
    }
    if (p.op == KNOX_REGEXP) return run_regexp(ctx, col, p.val, p.val_len, nullptr, (uint8_t *)bits, skip);
    if (col->dtype == KNOX_BYTES) {
        const bool ordered = p.op == KNOX_LT || p.op == KNOX_LE || p.op == KNOX_GT || p.op == KNOX_GE || p.op == KNOX_RANGE;
        if (p.op != KNOX_EQ && p.op != KNOX_NE && !ordered) return KNOX_E_UNSUPPORTED;
        if (p.val_len > 0xffffffffull || (p.val_len && !p.val)) return KNOX_E_INVALID;
        const size_t len2 = p.op == KNOX_RANGE ? (size_t)p.val2_len : 0;
        if (len2 > 0xffffffffull || (len2 && !p.val2)) return KNOX_E_INVALID;
        const size_t off2 = ((size_t)p.val_len + 15) & ~(size_t)15;
        int rc = ensure(ctx, ctx->bytes_val, off2 + len2 + 16);
        if (rc) return rc;
        if (ctx->leaf_stage && ctx->leaf_stage_off + off2 + len2 <= ctx->leaf_stage_cap) {
            // through the scan's page-locked staging: one copy, the host values need not outlive this call, no synchronisation
            char *st = ctx->leaf_stage + ctx->leaf_stage_off;
            if (p.val_len) memcpy(st, p.val, (size_t)p.val_len);
            if (len2) memcpy(st + off2, p.val2, len2);
            if (p.val_len || len2) HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, st, off2 + len2, hipMemcpyHostToDevice, ctx->stream));
            ctx->leaf_stage_off += (off2 + len2 + 63) & ~(size_t)63;
        } else {
            if (p.val_len) HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, p.val, (size_t)p.val_len, hipMemcpyHostToDevice, ctx->stream));
            if (len2) HIPCHK(ctx, hipMemcpyAsync((char *)ctx->bytes_val.p + off2, p.val2, len2, hipMemcpyHostToDevice, ctx->stream));
            if (p.val_len || len2) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host values not retained
        }
        prof_mark(ctx, 0);  // the dominant kernel of a bytes-leaf plan: 8 B pointer + the value bytes per row
        hipError_t e = launch_bytes_match(ctx->stream, p.op, bytes_view(col, n_rows, skip), (const uint8_t *)ctx->bytes_val.p,
                                          (uint32_t)p.val_len, (const uint8_t *)ctx->bytes_val.p + off2, (uint32_t)len2, (uint8_t *)bits, nullptr,
                                          nullptr);
        prof_mark(ctx, 1);
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            return KNOX_E_HIP;
        }
        return KNOX_OK;
    }
    if (p.op == KNOX_IN || p.op == KNOX_NOTIN) {
        // sorted unique canonical order keys of the set (NaN dropped: never a member; +-0 one key)
        const size_t w = (size_t)col->width;
        if (p.val_len % w != 0 || (p.val_len && !p.val)) return KNOX_E_INVALID;
        const size_t cnt = (size_t)(p.val_len / w);
        if (cnt > 0xffffffffu) return KNOX_E_RANGE;
        const bool same = ctx->inset_keys.p && ctx->inset_dtype == col->dtype && ctx->inset_raw.size() == (size_t)p.val_len &&
                          (p.val_len == 0 || memcmp(ctx->inset_raw.data(), p.val, (size_t)p.val_len) == 0);
        if (!same) {
            std::vector<uint64_t> keys;
            keys.reserve(cnt);
            const bool isf = dtype_is_float(col->dtype);
            const uint64_t m = wmask((int)w), sign = 1ull << (8 * w - 1), expm = w == 8 ? 0x7ff0000000000000ull : 0x7f800000ull;
            for (size_t i = 0; i < cnt; i++) {
                uint64_t v = 0;
                memcpy(&v, p.val + i * w, w);
                if (isf && (v & (m ^ sign)) > expm) continue;
                keys.push_back(order_key(col->dtype, v));
            }
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            ctx->inset_dtype = -1;   // not valid while it is being replaced
            // large sets: radix buckets over the sorted keys (top bits of key - smallest member), two or three dependent L2 loads per
            // row instead of a 16-step search (knox_bitset.hip)
            std::vector<uint32_t> gstart;
            uint32_t nb = 0, shift = 0;
            if (keys.size() > 4096) {
                nb = 1;
                while (nb < keys.size() && nb < (1u << 24)) nb <<= 1;
                const uint64_t range = keys.back() - keys.front();
                while ((range >> shift) >= (uint64_t)nb) shift++;
                gstart.assign((size_t)nb + 1, (uint32_t)keys.size());
                size_t i = 0;
                for (uint32_t b = 0; b <= nb; b++) {    // gstart[b] = first key whose bucket is >= b
                    while (i < keys.size() && ((keys[i] - keys.front()) >> shift) < (uint64_t)b) i++;
                    gstart[b] = (uint32_t)i;
                }
            }
            const size_t goff = (keys.size() * 8 + 255) & ~(size_t)255;
            int rc = ensure(ctx, ctx->inset_keys, goff + gstart.size() * 4 + 16);
            if (rc) return rc;
            if (!keys.empty()) {
                HIPCHK(ctx, hipMemcpyAsync(ctx->inset_keys.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice, ctx->stream));
                if (!gstart.empty())
                    HIPCHK(ctx, hipMemcpyAsync((char *)ctx->inset_keys.p + goff, gstart.data(), gstart.size() * 4, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vectors not retained
            }
            ctx->inset_gstart_off = gstart.empty() ? 0 : goff;
            ctx->inset_buckets = nb;
            ctx->inset_shift = shift;
            ctx->inset_raw.assign(p.val, p.val + p.val_len);
            ctx->inset_n = keys.size();
            ctx->inset_dtype = col->dtype;
        }
        const uint32_t *gst = ctx->inset_gstart_off ? (const uint32_t *)((const char *)ctx->inset_keys.p + ctx->inset_gstart_off) : nullptr;
        hipError_t e = launch_inset(ctx->stream, col->width, dtype_flavor(col->dtype), p.op == KNOX_NOTIN, col->dev, n_rows,
                                    (const uint64_t *)ctx->inset_keys.p, (uint32_t)ctx->inset_n, gst, ctx->inset_buckets, ctx->inset_shift,
                                    (uint8_t *)bits, nullptr);
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            return KNOX_E_HIP;
        }
        return KNOX_OK;
    }
    LeafHost lh;
    int rc = make_leaf(col->dtype, p.op, p.a, p.b, lh);
    if (rc) return rc;
    const uint64_t tt = lh.neg ? 0x1ull : 0x2ull;
    const knox_col *lc[1] = {col};
    // slot 15 is a throw-away partial slot for intermediate leaves
    return run_fused(ctx, col->width, &lh, lc, 1, tt, nullptr, 0, n_rows, col->pack_rows, 0, skip, (uint32_t *)bits, 15, grid, col->dtype);
}

// ------------------------------------------------------------------------------------------------
// scan
// ------------------------------------------------------------------------------------------------
KNOX_API int knox_scan_async(knox_ctx *ctx, const knox_pred *nodes, int n_nodes, knox_col *const *cols, int n_cols,
                             const knox_agg *aggs, int n_aggs, uint64_t row_base, const uint8_t *pack_skip, knox_bits *bits_out,
                             int want_pack_counts) {
    if (!ctx || !nodes || n_nodes <= 0 || !cols || n_cols <= 0 || n_aggs < 0 || n_aggs > 14 || (n_aggs && !aggs)) return KNOX_E_INVALID;
    for (int i = 0; i < n_cols; i++)
        if (!cols[i] || cols[i]->ctx != ctx) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    for (int i_ = 0; i_ < n_cols; i_++) { const int rc_ = col_acquire(ctx, cols[i_]); if (rc_) return rc_; }  // behind the columns' asynchronous uploads
    if (ctx->res_head - ctx->res_tail >= (uint64_t)knox_ctx::kResRing) return KNOX_E_BUSY;  // collect first
    knox_ctx::ResSlot &slot = ctx->res[ctx->res_head % knox_ctx::kResRing];
    slot.kind = 0;
    const size_t n_rows = cols[0]->n_rows;
    const size_t pack_rows = cols[0]->pack_rows;
    for (int i = 0; i < n_cols; i++)
        if (!cols[i] || cols[i]->n_rows != n_rows || cols[i]->ctx != ctx) return KNOX_E_INVALID;
    if (bits_out && bits_out->size != n_rows) return KNOX_E_INVALID;
    if ((want_pack_counts || pack_skip) && (pack_rows == 0 || (pack_rows & 63))) return KNOX_E_INVALID;
    const size_t n_packs = pack_rows ? (n_rows + pack_rows - 1) / pack_rows : 0;

    // ---- leaves
    std::vector<int> leaf_of_node(n_nodes, -1);
    std::vector<int> leaf_nodes;
    for (int i = 0; i < n_nodes; i++) {
        if (nodes[i].kind == KNOX_NODE_LEAF) {
            if (nodes[i].col < 0 || nodes[i].col >= n_cols) return KNOX_E_INVALID;
            leaf_of_node[i] = (int)leaf_nodes.size();
            leaf_nodes.push_back(i);
        } else if (nodes[i].kind != KNOX_NODE_AND && nodes[i].kind != KNOX_NODE_OR) {
            return KNOX_E_INVALID;
        }
    }
    const int n_leaves = (int)leaf_nodes.size();
    if (n_leaves == 0) {
        // constant tree (empty AND = all rows, empty OR = none, match_core.go:45-56,131-138):
        // run it as one TRUE / FALSE leaf on the first numeric column
        uint64_t tt0 = 0;
        int rc0 = make_truth_table(nodes, n_nodes, leaf_of_node, std::vector<bool>(), 0, tt0);
        if (rc0) return rc0;
        for (int c = 0; c < n_cols; c++) {
            if (cols[c]->dtype == KNOX_BYTES) continue;
            knox_pred k;
            memset(&k, 0, sizeof k);
            k.kind = KNOX_NODE_LEAF;
            k.col = c;
            k.op = (tt0 & 1) ? KNOX_TRUE : KNOX_FALSE;
            return knox_scan_async(ctx, &k, 1, cols, n_cols, aggs, n_aggs, row_base, pack_skip, bits_out, want_pack_counts);
        }
        return KNOX_E_UNSUPPORTED;
    }

    // ---- aggregate columns: distinct columns, union of needed work
    int agg_cols[14], n_agg_cols = 0;
    unsigned agg_needs[14];
    for (int i = 0; i < n_aggs; i++) {
        if (aggs[i].col < 0 || aggs[i].col >= n_cols || cols[aggs[i].col]->dtype == KNOX_BYTES) return KNOX_E_INVALID;
        int s = -1;
        for (int j = 0; j < n_agg_cols; j++)
            if (agg_cols[j] == aggs[i].col) s = j;
        if (s < 0) {
            s = n_agg_cols++;
            agg_cols[s] = aggs[i].col;
            agg_needs[s] = 0;
        }
        int rc = agg_need(aggs[i].func, cols[aggs[i].col]->dtype, agg_needs[s]);
        if (rc) return rc;
        slot.agg_slot[i] = s;
    }
    slot.n_aggs = n_aggs;

    // ---- workspace
    const int grid = std::max(1, ctx->n_cus * (ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 8));
    int rc = ensure(ctx, ctx->partials, sizeof(AggPartial) * (size_t)(grid + 1) * 16);
    if (!rc) rc = ensure(ctx, ctx->states, sizeof(AggStateDev) * 16 + 8 * 16);
    if (rc) return rc;
    // device result block: [16 states][16 int64 counts][n_packs int64 pack counts] -> ONE D2H copy
    const size_t res_base = sizeof(AggStateDev) * 16 + 8 * 16;
    const size_t res_bytes = res_base + (want_pack_counts ? n_packs * 8 : 0);
    rc = ensure(ctx, ctx->states, res_bytes + 64);
    if (rc) return rc;
    SkipPlan skip;
    uint64_t rows_pruned = 0;
    // the slot's page-locked staging holds the skip list and, behind it, the value bytes of the bytes leaves (EQ / NE / ordered): their
    // uploads need no stream synchronisation -- the callers' arrays are not retained, the slot is, until this scan is collected
    struct StageGuard {
        knox_ctx *c;
        ~StageGuard() { c->leaf_stage = nullptr; }
    } stage_guard{ctx};
    size_t leaf_bytes = 0;
    for (int l = 0; l < n_leaves; l++) {
        const knox_pred &lp = nodes[leaf_nodes[l]];
        if (cols[lp.col]->dtype == KNOX_BYTES && lp.op != KNOX_IN && lp.op != KNOX_NOTIN && lp.op != KNOX_REGEXP)
            leaf_bytes += (((size_t)lp.val_len + 15) & ~(size_t)15) + (lp.op == KNOX_RANGE ? (size_t)lp.val2_len : 0) + 64;
    }
    if (leaf_bytes > ((size_t)1 << 20)) leaf_bytes = 0;   // huge values: the synchronising path
    std::vector<uint32_t> live;
    size_t skip_bytes = 0, live_off = 0;
    if (pack_skip && n_packs) {
        live.reserve(n_packs);
        for (size_t p = 0; p < n_packs; p++) {
            if (pack_skip[p]) rows_pruned += std::min(pack_rows, n_rows - p * pack_rows);
            else live.push_back((uint32_t)p);
        }
        live_off = (n_packs + 15) & ~(size_t)15;
        skip_bytes = (live_off + live.size() * 4 + 63) & ~(size_t)63;
    }
    if (skip_bytes + leaf_bytes) {
        rc = ensure_host(ctx, slot.h_skip, slot.h_skip_cap, skip_bytes + leaf_bytes + 64);
        if (rc) return rc;
        if (leaf_bytes) {
            ctx->leaf_stage = (char *)slot.h_skip;
            ctx->leaf_stage_off = skip_bytes;
            ctx->leaf_stage_cap = skip_bytes + leaf_bytes;
        }
    }
    if (pack_skip && n_packs) {
        rc = ensure(ctx, ctx->pack_skip, live_off + live.size() * 4 + 16);
        if (rc) return rc;
        memcpy(slot.h_skip, pack_skip, n_packs);
        if (!live.empty()) memcpy((char *)slot.h_skip + live_off, live.data(), live.size() * 4);
        HIPCHK(ctx, hipMemcpyAsync(ctx->pack_skip.p, slot.h_skip, live_off + live.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        skip.d_skip = (const uint8_t *)ctx->pack_skip.p;
        skip.d_live = (const uint32_t *)((const char *)ctx->pack_skip.p + live_off);
        skip.n_live = (uint32_t)live.size();
        skip.h_skip = pack_skip;
        skip.n_packs = n_packs;
    }
    slot.rows_scanned = n_rows;

    // ---- can the whole tree run as one fused pass?
    bool fused = n_leaves <= kMaxLeaves;
    int width = 0;
    for (int l = 0; l < n_leaves && fused; l++) {
        const knox_col *c = cols[nodes[leaf_nodes[l]].col];
        if (c->dtype == KNOX_BYTES || nodes[leaf_nodes[l]].op == KNOX_IN || nodes[leaf_nodes[l]].op == KNOX_NOTIN || nodes[leaf_nodes[l]].op == KNOX_REGEXP) fused = false;
        else if (width == 0) width = c->width;
        else if (c->width != width) fused = false;
    }
    bool any_moments = false;
    for (int s = 0; s < n_agg_cols; s++) any_moments = any_moments || (agg_needs[s] & A_NEED_MOMENTS) != 0;
    // "pruned = no row matches, not read" is part of the contract.  The fused kernel honours it by walking only the
    // surviving packs, which needs whole power-of-two packs of at least one tile; any other shape (a short last pack,
    // small or odd pack sizes) takes the multi-pass plan, which clears the pruned ranges of the result.
    if (fused && pack_skip && n_packs) {
        const bool pow2 = !(pack_rows & (pack_rows - 1));
        if (!pow2 || n_rows % pack_rows != 0 || (uint64_t)tile_rows(width) * pick_iters(ctx, n_rows, width, pack_rows) > pack_rows) fused = false;
    }
    const bool need_bits = bits_out != nullptr || want_pack_counts || !fused || n_agg_cols > 1 || any_moments ||
                           (n_agg_cols == 1 && fused && cols[agg_cols[0]]->width != width);
    uint64_t *bits = bits_out ? bits_out->dev : nullptr;
    if (need_bits && !bits) {
        rc = ensure_for(ctx, ctx->tmp_bits, (bits_padded(n_rows) + 64) * 8, cols[0]);  // result + up to 7 stack temporaries, written while the columns stream
        if (rc) return rc;
        bits = (uint64_t *)ctx->tmp_bits.p;
        HIPCHK(ctx, launch_zero_bytes(ctx->stream, bits, bits_padded(n_rows) + 64));
    }

    int first_reduce_slot = 0;
    uint32_t per_pack_tiles = 0;  // != 0: the fused kernel left per-wave-tile counts
    bool direct = false;          // the result block is written straight into the slot's page-locked mirror
    bool packs_done = false;      // ... and the per-pack counts came with the finalize launch
    static const bool direct_ok = !(getenv("KNOX_DIRECT_RESULTS") && atoi(getenv("KNOX_DIRECT_RESULTS")) == 0);  // 0: diagnostics
    if (fused) {
        LeafHost lh[kMaxLeaves];
        const knox_col *lc[kMaxLeaves];
        std::vector<bool> neg(n_leaves);
        for (int l = 0; l < n_leaves; l++) {
            const knox_pred &p = nodes[leaf_nodes[l]];
            lc[l] = cols[p.col];
            rc = make_leaf(lc[l]->dtype, p.op, p.a, p.b, lh[l]);
            if (rc) return rc;
            neg[l] = lh[l].neg;
        }
        // the aggregate column rides in the same pass when it has the leaves' width; put it
        // first among the leaves if it is one of them so its registers are reused
        const knox_col *ac = nullptr;
        unsigned acls = 0;
        if (n_agg_cols >= 1 && cols[agg_cols[0]]->width == width) {
            ac = cols[agg_cols[0]];
            acls = agg_class(agg_needs[0]);
            first_reduce_slot = 1;
        }
        std::vector<int> lof = leaf_of_node;
        if (ac) {
            for (int l = 1; l < n_leaves; l++) {
                if (lc[l]->dev == ac->dev && lc[0]->dev != ac->dev) {
                    std::swap(lc[0], lc[l]);
                    std::swap(lh[0], lh[l]);
                    const bool t = neg[0]; neg[0] = neg[l]; neg[l] = t;
                    for (int i = 0; i < n_nodes; i++) {
                        if (lof[i] == 0) lof[i] = l;
                        else if (lof[i] == l) lof[i] = 0;
                    }
                    break;
                }
            }
        }
        uint64_t tt;
        rc = make_truth_table(nodes, n_nodes, lof, neg, n_leaves, tt);
        if (rc) return rc;
        // nothing after this pass reads the result block on the device: it is written where the host will read it
        direct = n_agg_cols <= first_reduce_slot && !any_moments && direct_ok;
        if (direct) {
            rc = ensure_host(ctx, slot.h_res, slot.h_res_cap, res_bytes);
            if (rc) return rc;
            ctx->res_dst = (char *)slot.h_res;
            ctx->post_packs = want_pack_counts ? n_packs : 0;
            ctx->post_packs_off = res_base;
        }
        rc = run_fused(ctx, width, lh, lc, n_leaves, tt, ac, acls, n_rows, pack_rows, row_base, pack_skip ? &skip : nullptr, (uint32_t *)bits, 0,
                       grid, ac ? ac->dtype : lc[0]->dtype, want_pack_counts ? &per_pack_tiles : nullptr);
        ctx->res_dst = nullptr;
        ctx->post_packs = 0;
        packs_done = ctx->post_packs_done;
        ctx->post_packs_done = false;
        if (rc) return rc;
        if (skip.used) slot.rows_scanned = n_rows - rows_pruned;
    } else {
        // multi-pass plan: each leaf into a scratch bitset, postfix combine with the bitset
        // kernels (exactly the reference's MatchAnd / MatchOr data flow, match_core.go:45-215)
        if (bits_out == nullptr && bits == nullptr) return KNOX_E_INVALID;
        bool all_leaves_skipped = pack_skip != nullptr;
        const size_t stride = (bits_padded(n_rows) + 64) / 8;  // uint64 words per scratch bitset
        rc = ensure_for(ctx, ctx->tmp_bits, stride * 8 * 8, cols[0]);
        if (rc) return rc;
        if (!bits_out) bits = (uint64_t *)ctx->tmp_bits.p;
        uint64_t *pool = (uint64_t *)ctx->tmp_bits.p;
        // stack slot 0 aliases the result buffer
        auto slot_ptr = [&](int s) { return s == 0 ? bits : pool + (size_t)s * stride; };
        int sp = 0;
        for (int i = 0; i < n_nodes; i++) {
            const knox_pred &p = nodes[i];
            if (p.kind == KNOX_NODE_LEAF) {
                if (sp >= 7) return KNOX_E_UNSUPPORTED;
                uint64_t *dst = slot_ptr(sp);
                { const hipError_t ze = launch_zero_bytes(ctx->stream, dst, bits_padded(n_rows) + 8); if (ze != hipSuccess) { ctx->last_hip = (int)ze; return KNOX_E_HIP; } }
                SkipPlan leaf_skip = skip;  // per leaf: `used` reports whether THIS leaf left the pruned packs unread
                rc = run_leaf_to_bits(ctx, p, cols[p.col], n_rows, dst, grid, pack_skip ? &leaf_skip : nullptr);
                if (rc) return rc;
                all_leaves_skipped = all_leaves_skipped && leaf_skip.used;
                sp++;
            } else {
                const int k = p.nchild;
                if (k < 0 || k > sp) return KNOX_E_INVALID;
                if (k == 0) {
                    if (sp >= 7) return KNOX_E_UNSUPPORTED;
                    uint64_t *dst = slot_ptr(sp);
                    hipError_t e = p.kind == KNOX_NODE_AND ? launch_bitop(ctx->stream, BIT_ONE, dst, nullptr, n_rows, nullptr)
                                                           : launch_zero_bytes(ctx->stream, dst, bits_padded(n_rows) + 8);
                    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
                    sp++;
                    continue;
                }
                uint64_t *dst = slot_ptr(sp - k);
                for (int j = 1; j < k; j++) {
                    hipError_t e = launch_bitop(ctx->stream, p.kind == KNOX_NODE_AND ? BIT_AND : BIT_OR, dst, slot_ptr(sp - k + j), n_rows, nullptr);
                    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
                }
                sp -= (k - 1);
            }
        }
        if (sp != 1) return KNOX_E_INVALID;
        if (pack_skip && n_packs) {
            // "pruned = no row matches": leaves that were evaluated everywhere (or a NOT-style combination over
            // untouched ranges) may have left bits inside pruned packs -- clear those ranges of the result
            // (every leaf wrote only the surviving packs of a cleared scratch bitset, and AND / OR keep zeros zero: nothing to clear then)
            if (!all_leaves_skipped) {
                const hipError_t ze = launch_zero_pruned(ctx->stream, (uint8_t *)bits, skip.d_skip, (uint32_t)n_packs, pack_rows, n_rows);
                if (ze != hipSuccess) { ctx->last_hip = (int)ze; return KNOX_E_HIP; }
            }
            if (all_leaves_skipped) slot.rows_scanned = n_rows - rows_pruned;
        }
        // total count (slot 0 count word) from the final bitset -- together with the per-pack counts when those are wanted or packs
        // were pruned (then from the surviving packs only): one pass instead of two over the whole bitset
        unsigned long long *d_cnt = (unsigned long long *)((int64_t *)((AggStateDev *)ctx->states.p + 16));
        hipError_t e = launch_zero_bytes(ctx->stream, d_cnt, 8);   // (the runtime's fill kernel takes 11 us for these 8 bytes)
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        if ((pack_skip || want_pack_counts) && n_packs && pack_rows && !(pack_rows & 63)) {
            int64_t *pc = (int64_t *)((char *)ctx->states.p + res_base);
            if (!want_pack_counts) {
                rc = ensure(ctx, ctx->pack_counts, n_packs * 8 + 64);
                if (rc) return rc;
                pc = (int64_t *)ctx->pack_counts.p;
            }
            e = launch_popcount_segments(ctx->stream, bits, n_rows, pack_rows, pc, n_packs, pack_skip ? skip.d_skip : nullptr, d_cnt);
            packs_done = want_pack_counts;
        } else {
            e = launch_popcount(ctx->stream, bits, n_rows, d_cnt);
        }
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        first_reduce_slot = 0;
    }

    // ---- remaining aggregate columns: masked reduce over the result bitset
    for (int s = first_reduce_slot; s < n_agg_cols; s++) {
        rc = run_reduce(ctx, cols[agg_cols[s]], (const uint32_t *)bits, agg_class(agg_needs[s]) ? agg_class(agg_needs[s]) : A_ISUM, n_rows, row_base, s, grid);
        if (rc) return rc;
    }

    // ---- var/std: second pass centred on the mean the first pass just produced (no host round trip)
    for (int s = 0; s < n_agg_cols; s++) {
        if (!(agg_needs[s] & A_NEED_MOMENTS)) continue;
        rc = run_moments(ctx, cols[agg_cols[s]], (const uint32_t *)bits, n_rows, s, grid);
        if (rc) return rc;
    }

    // ---- per-pack counts from the bitset
    slot.has_pack_counts = want_pack_counts && n_packs == 0;  // an empty column has no packs: nothing to count, nothing to copy
    slot.n_packs = n_packs;
    if (want_pack_counts && n_packs && packs_done) {
        slot.has_pack_counts = true;
    } else if (want_pack_counts && n_packs) {
        char *res = direct ? (char *)slot.h_res : (char *)ctx->states.p;
        hipError_t e = per_pack_tiles
                           ? launch_pack_counts(ctx->stream, (const uint32_t *)ctx->tile_counts.p, ctx->tc_grid, ctx->tc_rounds, per_pack_tiles,
                                                n_packs, (int64_t *)(res + res_base))
                           : launch_popcount_segments(ctx->stream, bits, n_rows, pack_rows, (int64_t *)(res + res_base), n_packs,
                                                      (pack_skip && n_packs) ? skip.d_skip : nullptr);   // pruned packs hold no bit (the contract, kept above)
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        slot.has_pack_counts = true;
    }
    // ---- small results -> this scan's pinned mirror (one copy), then its completion event
    if (!direct) {
        rc = ensure_host(ctx, slot.h_res, slot.h_res_cap, res_bytes);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(slot.h_res, ctx->states.p, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!slot.done) HIPCHK(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(slot.done, ctx->stream));
    ctx->res_head++;
    return KNOX_OK;
}

KNOX_API int knox_scan_collect(knox_ctx *ctx, int64_t *pack_counts, knox_agg_state *states, int n_aggs, knox_scan_out *out) {
    if (!ctx || n_aggs < 0 || (n_aggs && !states)) return KNOX_E_INVALID;
    if (ctx->res_tail == ctx->res_head) return KNOX_E_INVALID;  // nothing in flight
    knox_ctx::ResSlot &slot = ctx->res[ctx->res_tail % knox_ctx::kResRing];
    if (slot.kind != 0) return KNOX_E_INVALID;    // the oldest result is a bitset call's: knox_bits_collect
    HIPCHK(ctx, hipEventSynchronize(slot.done));  // waits for THIS scan only, later ones keep running
    ctx->res_tail++;
    const AggStateDev *st = (const AggStateDev *)slot.h_res;
    const int64_t *counts = (const int64_t *)(st + 16);
    if (n_aggs > slot.n_aggs) return KNOX_E_RANGE;
    for (int i = 0; i < n_aggs; i++) memcpy(&states[i], &st[slot.agg_slot[i]], sizeof(knox_agg_state));
    if (out) {
        out->count = counts[0];
        out->rows_scanned = (int64_t)slot.rows_scanned;
    }
    if (pack_counts) {
        if (!slot.has_pack_counts) return KNOX_E_INVALID;
        memcpy(pack_counts, (const char *)slot.h_res + sizeof(AggStateDev) * 16 + 8 * 16, slot.n_packs * 8);
    }
    return KNOX_OK;
}

KNOX_API int knox_scan(knox_ctx *ctx, const knox_pred *nodes, int n_nodes, knox_col *const *cols, int n_cols, const knox_agg *aggs,
                       int n_aggs, uint64_t row_base, const uint8_t *pack_skip, knox_bits *bits_out, int64_t *pack_counts,
                       knox_agg_state *states, knox_scan_out *out) {
    if (!ctx) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;  // collect the async scans first (FIFO)
    int rc = knox_scan_async(ctx, nodes, n_nodes, cols, n_cols, aggs, n_aggs, row_base, pack_skip, bits_out, pack_counts != nullptr);
    if (rc) return rc;
    return knox_scan_collect(ctx, pack_counts, states, n_aggs, out);
}

// ------------------------------------------------------------------------------------------------
// zone-map test over the statistics columns (Matcher.MatchRangeVectors)
// ------------------------------------------------------------------------------------------------
KNOX_API int knox_match_range_vectors(knox_ctx *ctx, const knox_col *mins, const knox_col *maxs, int op, uint64_t a, uint64_t b,
                                      const knox_bits *mask, knox_bits *bits) {
    if (!ctx || !mins || !maxs || !bits || mins->n_rows != maxs->n_rows || mins->dtype != maxs->dtype || bits->size != mins->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != bits->size) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    knox_col *cols[2] = {const_cast<knox_col *>(mins), const_cast<knox_col *>(maxs)};
    knox_pred p[3];
    memset(p, 0, sizeof p);
    int n = 0;
    bool and_mask = false;
    switch (op) {
    case KNOX_EQ:     // min <= v && max >= v
        p[0].op = KNOX_LE; p[0].col = 0; p[0].a = a; p[1].op = KNOX_GE; p[1].col = 1; p[1].a = a; p[2].kind = KNOX_NODE_AND; p[2].nchild = 2; n = 3; and_mask = true; break;
    case KNOX_RANGE:  // min <= to && max >= from
        p[0].op = KNOX_LE; p[0].col = 0; p[0].a = b; p[1].op = KNOX_GE; p[1].col = 1; p[1].a = a; p[2].kind = KNOX_NODE_AND; p[2].nchild = 2; n = 3; and_mask = true; break;
    case KNOX_GT: p[0].op = KNOX_GT; p[0].col = 1; p[0].a = a; n = 1; break;  // max > v
    case KNOX_GE: p[0].op = KNOX_GE; p[0].col = 1; p[0].a = a; n = 1; break;  // max >= v
    case KNOX_LT: p[0].op = KNOX_LT; p[0].col = 0; p[0].a = a; n = 1; break;  // min < v
    case KNOX_LE: p[0].op = KNOX_LE; p[0].col = 0; p[0].a = a; n = 1; break;  // min <= v
    case KNOX_NE: case KNOX_NOTIN:  // undecided, always true: bits.Copy(mask) or bits.One()
        return mask ? knox_bits_copy(ctx, bits, mask) : knox_bits_one(ctx, bits);
    default: return KNOX_E_UNSUPPORTED;
    }
    knox_scan_out out;
    int rc = knox_scan(ctx, p, n, cols, 2, nullptr, 0, 0, nullptr, bits, nullptr, nullptr, &out);
    if (rc) return rc;
    if (and_mask && mask) rc = knox_bits_and(ctx, bits, mask);
    return rc;
}

// numInSetMatcher.MatchRangeVectors (match_num.go:693-727) on materialized statistics vectors: which packs hold a member of the
// set inside [min, max].  (The reference's other branch -- compressed statistics blocks -- answers with the coarser RANGE test
// [set min, set max]: knox_match_range_vectors with KNOX_RANGE.)  NOT IN is "undecided, always true" (:810-817).
KNOX_API int knox_match_range_vectors_set(knox_ctx *ctx, const knox_col *mins, const knox_col *maxs, int negate, const void *set, size_t n_set,
                                          const knox_bits *mask, knox_bits *bits) {
    if (!ctx || !mins || !maxs || !bits || mins->n_rows != maxs->n_rows || mins->dtype != maxs->dtype || bits->size != mins->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != bits->size) return KNOX_E_INVALID;
    if (n_set && !set) return KNOX_E_INVALID;
    if (mins->dtype == KNOX_BYTES || dtype_is_float(mins->dtype)) return KNOX_E_UNSUPPORTED;  // integer sets (xroar bitmaps of uint64(v))
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    if (negate) return mask ? knox_bits_copy(ctx, bits, mask) : knox_bits_one(ctx, bits);
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, mins); if (rc_) return rc_; }
    { const int rc_ = col_acquire(ctx, maxs); if (rc_) return rc_; }
    const int w = mins->width;
    const bool is_signed = dtype_flavor(mins->dtype) == FL_SIGNED;
    std::vector<uint64_t> keys(n_set);
    for (size_t i = 0; i < n_set; i++) {
        uint64_t v = 0;
        memcpy(&v, (const char *)set + i * (size_t)w, (size_t)w);
        if (is_signed && w < 8 && (v >> (8 * w - 1))) v |= ~0ull << (8 * w);  // Go's uint64(T(v)) sign-extends
        keys[i] = v;
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    if (keys.size() > 0xffffffffull) return KNOX_E_RANGE;
    int rc = ensure(ctx, ctx->bytes_val, keys.size() * 8 + 64);
    if (!rc && !keys.empty()) rc = upload(ctx, ctx->bytes_val.p, keys.data(), keys.size() * 8);
    if (rc) return rc;
    hipError_t e = launch_range_set(ctx->stream, w, is_signed, mins->dev, maxs->dev, mins->n_rows, (const uint64_t *)ctx->bytes_val.p, (uint32_t)keys.size(),
                                    bits->dev, mask ? mask->dev : nullptr);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    return KNOX_OK;
}

// The Int128 / Int256 matchers' MatchRangeVectors (match_i128.go:78-375, match_i256.go:64-229) over limb columns of the
// min / max statistics: compositions of the wide compares, WITH the mask everywhere (cmp_i128_* take one):
//   EQ     (min <= v) & (max >= v) & mask          RANGE  (min <= to) & (max >= from) & mask     (IN: RANGE [set min, set max], :371-376)
//   LT/LE  (min < / <= v) & mask                   GT/GE  (max > / >= v) & mask                  NE / NOT IN: mask, or all ones
KNOX_API int knox_match_range_vectors_wide(knox_ctx *ctx, knox_col *const *mins, knox_col *const *maxs, int n_limbs, int op, const uint64_t *a,
                                           const uint64_t *b, const knox_bits *mask, knox_bits *bits) {
    if (!ctx || !mins || !maxs || !bits || (n_limbs != 2 && n_limbs != 4)) return KNOX_E_INVALID;
    if (mask && mask->size != bits->size) return KNOX_E_INVALID;
    if (op == KNOX_NE || op == KNOX_NOTIN) return mask ? knox_bits_copy(ctx, bits, mask) : knox_bits_one(ctx, bits);
    if (!a || (op == KNOX_RANGE && !b)) return KNOX_E_INVALID;
    int64_t cnt = 0;
    switch (op) {
    case KNOX_LT: case KNOX_LE: return knox_match_wide(ctx, mins, n_limbs, op, a, nullptr, mask, bits, &cnt);
    case KNOX_GT: case KNOX_GE: return knox_match_wide(ctx, maxs, n_limbs, op, a, nullptr, mask, bits, &cnt);
    case KNOX_EQ: case KNOX_RANGE: {
        knox_bits *t = nullptr;
        int rc = knox_bits_alloc(ctx, bits->size, &t);
        if (rc) return rc;
        rc = knox_match_wide(ctx, mins, n_limbs, KNOX_LE, op == KNOX_RANGE ? b : a, nullptr, mask, t, &cnt);   // minBits = (min <= to) & mask
        if (!rc) rc = knox_match_wide(ctx, maxs, n_limbs, KNOX_GE, a, nullptr, t, bits, &cnt);                     // bits = (max >= from) & minBits
        knox_bits_free(t);
        return rc;
    }
    default: return KNOX_E_UNSUPPORTED;
    }
}

// the bytes forms (bytesEqualMatcher.. bytesRangeMatcher.MatchRangeVectors, match_bytes.go:97-299): same
// shapes over StringPool statistics columns, bytes.Compare order
KNOX_API int knox_match_range_vectors_bytes(knox_ctx *ctx, const knox_col *mins, const knox_col *maxs, int op, const uint8_t *val, uint64_t val_len,
                                            const uint8_t *val2, uint64_t val2_len, const knox_bits *mask, knox_bits *bits) {
    if (!ctx || !mins || !maxs || !bits || mins->n_rows != maxs->n_rows || mins->dtype != KNOX_BYTES || maxs->dtype != KNOX_BYTES ||
        bits->size != mins->n_rows)
        return KNOX_E_INVALID;
    if (mask && mask->size != bits->size) return KNOX_E_INVALID;
    if ((val_len && !val) || (op == KNOX_RANGE && val2_len && !val2)) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    knox_col *cols[2] = {const_cast<knox_col *>(mins), const_cast<knox_col *>(maxs)};
    knox_pred p[3];
    memset(p, 0, sizeof p);
    int n = 0;
    bool and_mask = false;
    auto leaf = [&](int i, int mode, int col, const uint8_t *v, uint64_t len) {
        p[i].op = mode; p[i].col = col; p[i].val = v; p[i].val_len = len;
    };
    switch (op) {
    case KNOX_EQ:     // min <= v && max >= v
        leaf(0, KNOX_LE, 0, val, val_len); leaf(1, KNOX_GE, 1, val, val_len); p[2].kind = KNOX_NODE_AND; p[2].nchild = 2; n = 3; and_mask = true; break;
    case KNOX_RANGE:  // min <= to && max >= from
        leaf(0, KNOX_LE, 0, val2, val2_len); leaf(1, KNOX_GE, 1, val, val_len); p[2].kind = KNOX_NODE_AND; p[2].nchild = 2; n = 3; and_mask = true; break;
    case KNOX_GT: leaf(0, KNOX_GT, 1, val, val_len); n = 1; break;  // max > v
    case KNOX_GE: leaf(0, KNOX_GE, 1, val, val_len); n = 1; break;  // max >= v
    case KNOX_LT: leaf(0, KNOX_LT, 0, val, val_len); n = 1; break;  // min < v
    case KNOX_LE: leaf(0, KNOX_LE, 0, val, val_len); n = 1; break;  // min <= v
    case KNOX_NE: case KNOX_NOTIN:  // undecided, always true
        return mask ? knox_bits_copy(ctx, bits, mask) : knox_bits_one(ctx, bits);
    default: return KNOX_E_UNSUPPORTED;
    }
    knox_scan_out out;
    int rc = knox_scan(ctx, p, n, cols, 2, nullptr, 0, 0, nullptr, bits, nullptr, nullptr, &out);
    if (rc) return rc;
    if (and_mask && mask) rc = knox_bits_and(ctx, bits, mask);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// knox_cmp: the finest boundary (host slices in, host bitset out)
// ------------------------------------------------------------------------------------------------
// Round 4: the call is latency, not bandwidth -- a cmp func var is called per vector (2^16 rows in production packs) and the
// reference's AVX2 kernel answers in 4.5-7.4 us there.  So the fixed part is as short as the boundary allows:
//   * a PAGE-LOCKED source (knox_host_alloc: the arena hook of a Go host) is not copied at all -- the scan kernel reads it
//     in place over PCIe (page-locked host memory is device addressable at its own address);
//   * a pageable source is staged as before (host copy into the pinned ring, DMA, scan from HBM);
//   * the bitset is written by the kernel straight into page-locked memory (the staging ring's second slot) and the count
//     and state block by the finalize launch into a page-locked result block (run_fused's res_dst), so the call ends in ONE
//     stream synchronisation and one host memcpy of (n + 7) / 8 bytes -- no device-to-host copy commands.
// tools/cmp_latency.py measures it against the AVX2 port; the table is in DESIGN.md 7 and sets the Go side's row floor.
KNOX_API int64_t knox_cmp(knox_ctx *ctx, int dtype, int op, const void *src, size_t n, const void *a, const void *b, uint8_t *bits) {
    if (!ctx) return KNOX_E_INVALID;
    if (n == 0 || src == nullptr) return 0;  // cmp/tests/int64.go:212-213
    const int w = dtype_width(dtype);
    if (w == 0 || !bits || !a || (op == KNOX_RANGE && !b)) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;  // the result block below is the scans' too
    hipSetDevice(ctx->device);
    uint64_t av = 0, bv = 0;
    memcpy(&av, a, (size_t)w);
    if (b) memcpy(&bv, b, (size_t)w);
    LeafHost lh;
    int rc = make_leaf(dtype, op, av, bv, lh);
    if (rc) return rc;
    rc = stage_init(ctx);
    if (rc) return rc;
    const size_t res_bytes = sizeof(AggStateDev) * 16 + 8 * 16;
    rc = ensure_host(ctx, ctx->cmp_res, ctx->cmp_res_cap, res_bytes);
    if (rc) return rc;
    const size_t out_bytes = (bits_padded(n) + 64) * 8;                 // what the kernel may touch (whole words + slack)
    const bool bits_pinned = out_bytes <= knox_ctx::kStageBytes;        // 2^28 rows: the ring slot holds the bitset
    const bool src_pinned = ((uintptr_t)src & 15) == 0 && is_pinned_host(src);
    if (!src_pinned) rc = ensure(ctx, ctx->tmp_vals, n * (size_t)w + 512);
    if (!rc && !bits_pinned) rc = ensure(ctx, ctx->tmp_bits, out_bytes);
    const int grid = std::max(1, ctx->n_cus * (ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 8));
    if (!rc) rc = ensure(ctx, ctx->partials, sizeof(AggPartial) * (size_t)(grid + 1) * 16);
    if (!rc) rc = ensure(ctx, ctx->states, res_bytes);
    if (rc) return rc;
    if (!src_pinned) {
        rc = upload(ctx, ctx->tmp_vals.p, src, n * (size_t)w);  // returns when the ring's DMAs are done: slot 1 is free again
        if (rc) return rc;
    }
    knox_col tmp;
    tmp.ctx = ctx; tmp.dtype = dtype; tmp.width = w; tmp.n_rows = n; tmp.pack_rows = n; tmp.dev = src_pinned ? const_cast<void *>(src) : ctx->tmp_vals.p;
    tmp.buf = nullptr; tmp.buf_len = 0; tmp.owned = false;
    const knox_col *lc[1] = {&tmp};
    const uint64_t tt = lh.neg ? 0x1ull : 0x2ull;
    uint32_t *out = bits_pinned ? (uint32_t *)ctx->stage[1] : (uint32_t *)ctx->tmp_bits.p;
    int64_t *h_cnt = (int64_t *)((AggStateDev *)ctx->cmp_res + 16);
    h_cnt[0] = -1;
    ctx->res_dst = (char *)ctx->cmp_res;
    ctx->post_packs = 0;
    rc = run_fused(ctx, w, &lh, lc, 1, tt, nullptr, 0, n, n, 0, nullptr, out, 0, grid, dtype);
    ctx->res_dst = nullptr;
    if (rc) return rc;
    if (bits_pinned) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        CopyPool::get().copy(bits, out, bits_bytes(n));  // exactly (n+7)/8 bytes: poison tail untouched
    } else {
        rc = download(ctx, bits, out, bits_bytes(n));    // synchronises the stream
        if (rc) return rc;
    }
    return h_cnt[0];
}

// ------------------------------------------------------------------------------------------------
// reducers
// ------------------------------------------------------------------------------------------------
KNOX_API int knox_reduce(knox_ctx *ctx, const knox_col *col, const knox_bits *mask, uint64_t row_base, knox_agg_state *state) {
    if (!ctx || !col || !state || col->dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (mask && mask->size != col->n_rows) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, col); if (rc_) return rc_; }
    const int grid = std::max(1, ctx->n_cus * (ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 8));
    int rc = ensure(ctx, ctx->partials, sizeof(AggPartial) * (size_t)(grid + 1) * 16);
    if (!rc) rc = ensure(ctx, ctx->states, sizeof(AggStateDev) * 16 + 8 * 16);
    if (rc) return rc;
    rc = run_reduce(ctx, col, mask ? (const uint32_t *)mask->dev : nullptr, A_ALL, col->n_rows, row_base, 0, grid);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(state, ctx->states.p, sizeof(knox_agg_state), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return KNOX_OK;
}

KNOX_API int knox_reduce_moments(knox_ctx *ctx, const knox_col *col, const knox_bits *mask, knox_agg_state *state) {
    if (!ctx || !col || !state || col->dtype == KNOX_BYTES || state->dtype != col->dtype) return KNOX_E_INVALID;
    if (mask && mask->size != col->n_rows) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    { const int rc_ = col_acquire(ctx, col); if (rc_) return rc_; }
    const int grid = std::max(1, ctx->n_cus * (ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 4));
    int rc = ensure(ctx, ctx->mom_partials, (size_t)(grid + 1) * 4 * sizeof(double) + 64);
    if (!rc) rc = ensure(ctx, ctx->misc, 256);
    if (rc) return rc;
    const double shift = state->count > 0 ? state->fsum_hi / (double)state->count : 0.0;
    double *d_out = (double *)((char *)ctx->misc.p + 64);
    hipError_t e = launch_moments(ctx->stream, col->width, dtype_flavor(col->dtype), grid, col->dev, mask ? (const uint32_t *)mask->dev : nullptr,
                                  col->n_rows, nullptr, shift, (double *)ctx->mom_partials.p, d_out, nullptr);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    double h[5];
    HIPCHK(ctx, hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    state->m_shift = h[0]; state->m_dev_hi = h[1]; state->m_dev_lo = h[2]; state->m_sq_hi = h[3]; state->m_sq_lo = h[4];
    state->flags |= 2u;
    return KNOX_OK;
}

