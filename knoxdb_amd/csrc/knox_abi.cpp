// knox_abi.hip -- C ABI (include/knoxgpu.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary: contexts, HBM-resident columns with a pinned staging
// ring (hipMemcpyAsync), the predicate-tree planner (postfix tree -> key-space leaves + truth
// table, or a multi-pass plan over scratch bitsets), result collection, exact finalisation of
// reducer values, and the RCCL exchange of partial aggregate states.
//
// There is NO CPU fallback: if the HIP runtime or the device is missing every entry point
// fails with KNOX_E_NODEVICE / KNOX_E_HIP.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include <rccl/rccl.h>  // types only: librccl is dlopen'ed on first knox_comm_* call

#include "../../include/knoxgpu.h"
#include "knox_arena.h"
#include "knox_finalize.h"
#include "knox_kernels.h"
#include "knox_regex.h"
#include "knox_xxh3.h"  // host flavour: KNOX_HD / KNOX_XXH3_CONST empty

using namespace knox;

#define KNOX_API extern "C" __attribute__((visibility("default")))

static_assert(sizeof(knox_agg_state) == sizeof(AggStateDev), "knox_agg_state layout");

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    const void *for_col = nullptr;  // the column region a verified write buffer was chosen for (knox_arena.h), nullptr = heap memory
};

struct knox_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int last_hip = 0;
    int n_cus = 256;
    int blocks_per_cu = 0;  // 0 = auto: 2 workgroups/CU for pure streaming passes, 3 when the pass carries
                            // float/min-max aggregates or several leaves (profiles/r01_sweep_blocks_per_cu.txt)
#ifdef KNOX_DIAG
    int variant = 1;  // diagnostics builds only (make diag): 2 bitset assembled but not stored, 3 plain stores, 4 write-through stores, 5 XCD-contiguous tiles
#endif
    // staging ring (pinned) for host -> HBM uploads
    static constexpr size_t kStageBytes = 32u << 20;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    // grow-only device scratch
    DevBuf partials;     // AggPartial[(grid+1) * slots]
    DevBuf states;       // AggStateDev[slots] + int64 count
    DevBuf tmp_vals;     // knox_cmp: staged values
    DevBuf tmp_bits;     // scratch bitset(s)
    DevBuf pack_skip;    // device copy of the caller's pack_skip bytes
    DevBuf pack_counts;  // int64 per pack
    DevBuf misc;         // flags / counters / small outputs
    DevBuf mom_partials; // var/std second pass: 4 doubles per workgroup
    DevBuf tile_counts;  // fused scan: popcount per wave tile -> per-pack counts
    uint32_t tc_grid = 0, tc_rounds = 0;  // layout of tile_counts as the last scan wrote it
    DevBuf idx_scratch;  // indexes: block offsets
    DevBuf idx_rows;     // indexes: row ids; window sort: the keys of all rows
    DevBuf sort_buf;     // window sort: the (key, row, value) triples, one or two sets
    DevBuf bytes_val;    // bytes leaf value
    // ring of pinned result mirrors: up to kResRing scans may be in flight (software
    // pipelining: enqueue scan i+1 before collecting scan i keeps the device busy)
    struct ResSlot {
        void *h_res = nullptr;
        size_t h_res_cap = 0;
        void *h_packs = nullptr;
        size_t h_packs_cap = 0;
        hipEvent_t done = nullptr;
        int agg_slot[16];
        int n_aggs = 0;
        size_t n_packs = 0;
        bool has_pack_counts = false;
        uint64_t rows_scanned = 0;
        int kind = 0;          // 0 scan, 1 bitset flags (any / all), 2 popcount
        size_t bits_size = 0;  // kind 1: the bitset's size ("all" of an empty bitset is false)
    };
    static constexpr int kResRing = 4;
    ResSlot res[kResRing + 1];  // [kResRing]: the blocking bitset calls' own slot, never part of the FIFO
    DevBuf ring_dev;            // 16 bytes of device results per slot (flags / popcount)
    // A single-pass scan without a var / std pass has nothing that reads its result block on the device: the block
    // [16 states][16 counts][pack counts] is then written straight into the slot's page-locked mirror (res_dst), and its
    // whole follow-up work -- finalize + per-pack counts -- is ONE launch (post_packs): what a step costs beyond the scan
    // kernel is ~6.5 us of dispatch gap per dependent launch, so two launches and a copy become one launch
    char *res_dst = nullptr;
    size_t post_packs = 0;       // != 0: run_fused also writes this many per-pack counts at res_dst + post_packs_off
    size_t post_packs_off = 0;
    bool post_packs_done = false;
    // large results (row ids, per-window values) land here; the *_view entry points hand out pointers into it
    void *h_view = nullptr;
    size_t h_view_cap = 0;
    uint64_t res_head = 0, res_tail = 0;  // head: next slot to fill, tail: oldest uncollected
    // per-kernel stopwatch (knox_prof_*)
    bool prof = false;
    static constexpr int kProfRing = 512;
    hipEvent_t prof_ev[2 * kProfRing] = {};
    int prof_n = 0;
    // All device memory of the context comes out of ONE bounded arena (knox_arena.h): physical chunks
    // classified once, columns and the bitsets / scratch written beside them carved out of different placement classes
    int placement = 1;             // 0 = plain hipMalloc
    Arena arena;
    void *probe_partials = nullptr;  // the classification probes' own partial buffer (plain allocation: the arena cannot allocate while it grows)
    hipEvent_t probe_ev[24] = {};
    // RCCL
    void *rccl = nullptr;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    DevBuf comm_buf;
};

struct knox_col {
    knox_ctx *ctx;
    int dtype;
    int width;  // bytes per value; bytes column: 8 (ptr words)
    size_t n_rows;
    size_t pack_rows;
    void *dev;      // values, or ptr[] for a bytes column
    uint8_t *buf;   // bytes column payload
    size_t buf_len;
    bool owned;
    // bytes column held the way the reference holds it, ONE StringPool per pack (pkg/stringx/pool.go:45-58: 32-bit
    // offsets are relative to the pack's own pool): pool_base[p] = where pack p's pool starts inside buf (device
    // array + host mirror, packs_cap + 1 entries); nullptr = the whole column is one pool
    uint64_t *pool_base = nullptr;
    uint64_t *h_pool_base = nullptr;
    size_t n_packs = 0;                               // packs present
    size_t packs_cap = 0, rows_cap = 0, buf_cap = 0;  // capacity of an appendable column
};

struct knox_bits {
    knox_ctx *ctx;
    size_t size;
    uint64_t *dev;
    size_t cap_bytes;
    bool owned = true;
    double pair_ratio = 0.0;  // knox_bits_alloc_for: scan / read-only of the (column, this bitset) pair as verified at allocation, 0 = not verified
    int pair_tried = 0;       // candidate windows looked at
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
#define HIPCHK(ctx, call)                              \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) {                        \
            if (ctx) (ctx)->last_hip = (int)_e;        \
            return KNOX_E_HIP;                         \
        }                                              \
    } while (0)

static int dtype_width(int dtype) {
    switch (dtype) {
    case KNOX_INT64: case KNOX_UINT64: case KNOX_FLOAT64: return 8;
    case KNOX_INT32: case KNOX_UINT32: case KNOX_FLOAT32: return 4;
    case KNOX_INT16: case KNOX_UINT16: return 2;
    case KNOX_INT8: case KNOX_UINT8: return 1;
    default: return 0;
    }
}
static bool dtype_is_float(int dtype) { return dtype == KNOX_FLOAT64 || dtype == KNOX_FLOAT32; }
static bool dtype_is_signed(int dtype) { return dtype >= KNOX_INT64 && dtype <= KNOX_INT8; }
static uint32_t dtype_flavor(int dtype) { return dtype_is_float(dtype) ? FL_FLOAT : dtype_is_signed(dtype) ? FL_SIGNED : FL_UNSIGNED; }

static double arena_probe(void *user, const void *col, uint64_t rows, void *bits);

// Device memory of a context: the arena when placement is on (kind READ = columns / payloads the scans stream through,
// kind WRITE = bitsets / scratch they write), plain hipMalloc when it is off or the device has no virtual-memory API.
// The arena's budget is a hard cap.
static hipError_t dev_malloc(knox_ctx *ctx, void **p, size_t bytes, int kind = ARENA_READ) {
    if (ctx && ctx->placement) {
        const hipError_t e = ctx->arena.alloc(p, bytes, kind);
        if (e != hipErrorNotSupported) return e;
        (void)hipGetLastError();
    }
    return hipMalloc(p, bytes);
}
static void dev_free(knox_ctx *ctx, void *p) {
    if (!p) return;
    if (ctx && ctx->arena.release(p)) return;
    (void)hipFree(p);
}

static int ensure(knox_ctx *ctx, DevBuf &b, size_t bytes, int kind = ARENA_WRITE) {
    if (b.cap >= bytes) return KNOX_OK;
    if (b.p) {
        // the stream may still be using the old buffer
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        dev_free(ctx, b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = std::max<size_t>(bytes, 4096);
    cap = (cap + 255) & ~(size_t)255;
    hipError_t e = dev_malloc(ctx, &b.p, cap, kind);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        b.p = nullptr;
        return KNOX_E_NOMEM;
    }
    b.cap = cap;
    b.for_col = nullptr;
    return KNOX_OK;
}

// scratch a scan WRITES while it streams through `col` (scratch bitsets of multi-pass plans, per-tile counts): large
// buffers are verified against that column like a knox_bits_alloc_for bitset (knox_arena.h)
struct knox_col;
static int ensure_for(knox_ctx *ctx, DevBuf &b, size_t bytes, const knox_col *col);

// RCCL send / receive buffers stay ordinary hipMalloc memory (a few hundred bytes)
static int ensure_plain(knox_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.cap >= bytes) return KNOX_OK;
    if (b.p) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t cap = (std::max<size_t>(bytes, 4096) + 255) & ~(size_t)255;
    if (hipMalloc(&b.p, cap) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        return KNOX_E_NOMEM;
    }
    b.cap = cap;
    return KNOX_OK;
}

static int ensure_host(knox_ctx *ctx, void *&p, size_t &cap, size_t bytes) {
    if (cap >= bytes) return KNOX_OK;
    if (p) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipHostFree(p));
        p = nullptr;
        cap = 0;
    }
    size_t c = std::max<size_t>(bytes, 4096);
    hipError_t e = hipHostMalloc(&p, c, hipHostMallocDefault);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        p = nullptr;
        return KNOX_E_NOMEM;
    }
    cap = c;
    return KNOX_OK;
}

static inline size_t bits_bytes(size_t size_bits) { return (size_bits + 7) >> 3; }
static inline size_t bits_padded(size_t size_bits) { return ((size_bits + 63) >> 6) << 3; }

// ------------------------------------------------------------------------------------------------
// library / context
// ------------------------------------------------------------------------------------------------
KNOX_API int knox_abi_version(void) { return KNOX_ABI_VERSION; }

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
    // arena budget: KNOX_ARENA_GIB, knox_ctx_set_arena, or three quarters of the device; physical memory is only taken
    // as allocations need it, 2 GiB at a time
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
    size_t budget = total_b / 4 * 3;
    env = getenv("KNOX_ARENA_GIB");
    if (env && atof(env) > 0) budget = (size_t)(atof(env) * (double)((size_t)1 << 30));
    c->arena.configure(device, budget, arena_probe, c);
    *out = c;
    return KNOX_OK;
}

KNOX_API int knox_ctx_open(int device, knox_ctx **out) { return ctx_open(device, nullptr, false, out); }
KNOX_API int knox_ctx_open_stream(int device, void *hip_stream, knox_ctx **out) { return ctx_open(device, hip_stream, true, out); }

KNOX_API int knox_ctx_close(knox_ctx *ctx) {
    if (!ctx) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm) knox_comm_destroy(ctx);
    DevBuf *bufs[] = {&ctx->partials, &ctx->states, &ctx->tmp_vals, &ctx->tmp_bits, &ctx->pack_skip, &ctx->pack_counts,
                      &ctx->misc, &ctx->mom_partials, &ctx->tile_counts, &ctx->idx_scratch, &ctx->idx_rows, &ctx->sort_buf, &ctx->bytes_val, &ctx->ring_dev};
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
        if (r.h_packs) hipHostFree(r.h_packs);
        if (r.done) hipEventDestroy(r.done);
    }
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return KNOX_OK;
}

KNOX_API int knox_ctx_sync(knox_ctx *ctx) {
    if (!ctx) return KNOX_E_INVALID;
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
    out->transient_spacer_bytes = ctx->arena.spacer_bytes_used;
    out->enabled = ctx->placement && ctx->arena.usable;
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
static void prof_mark(knox_ctx *ctx, int which) {
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

static int upload(knox_ctx *ctx, void *dst, const void *src, size_t bytes) {
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
        memcpy(ctx->stage[slot], (const char *)src + off, n);
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

static int download(knox_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return KNOX_OK;
    if (is_pinned_host(dst)) {
        HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return KNOX_OK;
    }
    int rc = stage_init(ctx);
    if (rc) return rc;
    // two ring slots: the DMA of chunk i+1 runs while chunk i is copied out to the caller's pageable memory
    const size_t chunk = std::min<size_t>(knox_ctx::kStageBytes, (size_t)4 << 20);
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
        memcpy((char *)dst + copied, ctx->stage[done], len[done]);
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
    if (!ctx->probe_ev[0])
        for (int i = 0; i <= kLaunches; i++)
            if (hipEventCreate(&ctx->probe_ev[i]) != hipSuccess) return 0.0;
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
    a.partials = (AggPartial *)ctx->probe_partials;
    const int g = (int)std::min<uint64_t>(a.n_tiles, (uint64_t)grid);
    if (hipEventRecord(ctx->probe_ev[0], ctx->stream) != hipSuccess) return 0.0;
    for (int i = 0; i < kLaunches; i++) {
        a.bits = (i & 1) ? (uint32_t *)bits : nullptr;
        if (launch_probe_scan(ctx->stream, g, a) != hipSuccess || hipEventRecord(ctx->probe_ev[i + 1], ctx->stream) != hipSuccess) return 0.0;
    }
    if (hipEventSynchronize(ctx->probe_ev[kLaunches]) != hipSuccess) return 0.0;
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
static double arena_probe(void *user, const void *col, uint64_t rows, void *bits) { return probe_pair((knox_ctx *)user, col, rows, bits); }

static int col_new(knox_ctx *ctx, int dtype, size_t n_rows, size_t pack_rows, bool alloc, knox_col **out) {
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
    return download(ctx, host, (const char *)col->dev + first_row * (size_t)col->width, n_rows * (size_t)col->width);
}

KNOX_API int knox_col_free(knox_col *col) {
    if (!col) return KNOX_E_INVALID;
    hipSetDevice(col->ctx->device);
    hipStreamSynchronize(col->ctx->stream);
    if (col->owned) dev_free(col->ctx, col->dev);
    dev_free(col->ctx, col->buf);
    dev_free(col->ctx, col->pool_base);
    free(col->h_pool_base);
    delete col;
    return KNOX_OK;
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

static int ensure_for(knox_ctx *ctx, DevBuf &b, size_t bytes, const knox_col *col) {
    const void *cp = nullptr;
    size_t cb = 0;
    if (col && col->n_rows) col_stream(col, &cp, &cb);
    if (!ctx->placement || !cp || cb < ((size_t)1 << 30) || bytes < ((size_t)16 << 20)) return ensure(ctx, b, bytes, ARENA_WRITE);
    if (b.cap >= bytes && b.for_col == cp) return KNOX_OK;
    if (b.p) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        dev_free(ctx, b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t cap = (std::max(bytes, b.cap) + 255) & ~(size_t)255;
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

static int bits_alloc(knox_ctx *ctx, size_t size_bits, const knox_col *for_col, knox_bits **out) {
    if (!ctx || !out) return KNOX_E_INVALID;
    *out = nullptr;
    hipSetDevice(ctx->device);
    knox_bits *b = new (std::nothrow) knox_bits();
    if (!b) return KNOX_E_NOMEM;
    b->ctx = ctx;
    b->size = size_bits;
    b->cap_bytes = bits_padded(size_bits) + 64;
    hipError_t e = hipErrorNotSupported;
    if (for_col && ctx->placement && for_col->n_rows) {
        // a write buffer of its own, verified against the column itself (knox_arena.h)
        (void)hipStreamSynchronize(ctx->stream);
        const void *cp;
        size_t cb;
        col_stream(for_col, &cp, &cb);
        e = ctx->arena.alloc_write_for((void **)&b->dev, b->cap_bytes, cp, cb, (double)cb / (double)for_col->n_rows, &b->pair_ratio, &b->pair_tried);
    }
    if (e == hipErrorNotSupported) e = dev_malloc(ctx, (void **)&b->dev, b->cap_bytes, ARENA_WRITE);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        delete b;
        return KNOX_E_NOMEM;
    }
    e = hipMemsetAsync(b->dev, 0, b->cap_bytes, ctx->stream);
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
    return bits_alloc(ctx, size_bits, col, out);
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
    HIPCHK(ctx, hipMemsetAsync(b->dev, 0, b->cap_bytes, ctx->stream));
    return upload(ctx, b->dev, host, bits_bytes(b->size));
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
    HIPCHK(ctx, hipMemsetAsync(*dev16, 0, 16, ctx->stream));
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
    HIPCHK(ctx, hipMemsetAsync(b->dev, 0, b->cap_bytes, ctx->stream));
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
    int rc = ensure(ctx, ctx->misc, 256);
    if (rc) return rc;
    unsigned long long *d_cnt = (unsigned long long *)((char *)ctx->misc.p + 64);
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
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
    // sorted unique keys in the kernel's order: unsigned lexicographic with the top limb's sign bit flipped
    const size_t L = (size_t)n_limbs;
    std::vector<size_t> idx(n_set);
    for (size_t i = 0; i < n_set; i++) idx[i] = i;
    auto key = [&](size_t i, size_t l) { return l == 0 ? set[i * L] ^ (1ull << 63) : set[i * L + l]; };
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
    for (int l = 0; l < n_limbs; l++) lp[l] = (const uint64_t *)limbs[l]->dev;
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
struct LeafHost {
    uint64_t lo = 0, diff = 0;
    bool neg = false;
    bool is_float = false;
};

static inline uint64_t wmask(int w) { return w == 8 ? ~0ull : ((1ull << (8 * w)) - 1); }

static uint64_t host_fkey(uint64_t u, int w) {
    if (w == 8) return u ^ ((uint64_t)((int64_t)u >> 63) | (1ull << 63));
    const uint32_t x = (uint32_t)u;
    return (uint64_t)(x ^ ((uint32_t)((int32_t)x >> 31) | 0x80000000u));
}

static int make_leaf(int dtype, int op, uint64_t a, uint64_t b, LeafHost &out) {
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

static uint32_t pack_shift_for(size_t pack_rows, int width) {
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
static uint32_t pick_iters(knox_ctx *ctx, size_t n_rows, int width, size_t pack_rows_or_0) {
    const char *env = getenv("KNOX_ITERS");
    uint32_t it = env && atoi(env) > 0 ? (uint32_t)atoi(env) : 2u;
    if (it > (uint32_t)kMaxIters) it = kMaxIters;
    const uint64_t base = (uint64_t)tile_rows(width);
    const uint64_t want_tiles = (uint64_t)ctx->n_cus * 8;
    while (it > 1 && n_rows / (base * it) < want_tiles) it >>= 1;
    if (pack_rows_or_0) while (it > 1 && (pack_rows_or_0 % (base * it)) != 0) it >>= 1;
    return it;
}

// Packs pruned by the caller's statistics.  d_skip: one byte per pack (tail kernel); d_live: the
// compact list of surviving packs the streaming kernel walks; h_skip: the caller's array (only
// read before the call returns) for zeroing the pruned bitset ranges.
struct SkipPlan {
    const uint8_t *d_skip = nullptr;
    const uint32_t *d_live = nullptr;
    uint32_t n_live = 0;
    const uint8_t *h_skip = nullptr;
    size_t n_packs = 0;
    bool used = false;  // out: the streaming kernel really skipped the pruned packs
};

// device view of a bytes column (knox_kernels.h BytesView); with a skip plan only the surviving packs are walked
static BytesView bytes_view(const knox_col *c, size_t n_rows, SkipPlan *skip = nullptr) {
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
    a.iters = pick_iters(ctx, n_rows, width, want_skip ? pack_rows : 0);
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
                size_t runs = 0;
                for (size_t p = 0; p < skip->n_packs; p++)
                    if (skip->h_skip[p] && (p == 0 || !skip->h_skip[p - 1])) runs++;
                const size_t pack_bytes = pack_rows >> 3;
                if (runs > 16) {
                    HIPCHK(ctx, hipMemsetAsync(bits, 0, skip->n_packs * pack_bytes, ctx->stream));
                } else {
                    for (size_t p = 0; p < skip->n_packs;) {
                        if (!skip->h_skip[p]) { p++; continue; }
                        size_t e = p;
                        while (e < skip->n_packs && skip->h_skip[e]) e++;
                        HIPCHK(ctx, hipMemsetAsync((uint8_t *)bits + p * pack_bytes, 0, (e - p) * pack_bytes, ctx->stream));
                        p = e;
                    }
                }
            }
        }
    } else {
        a.pack_shift = 255;
    }
    // auto occupancy: the pure streaming passes run best at 2 workgroups per CU, VALU-heavier ones at 3
    // (profiles/r01_sweep_blocks_per_cu.txt): 8-byte / 4-byte integer streaming 2 WG/CU; float or min-max
    // aggregates, several leaves, 2-byte types 3; 1-byte types and float32 sums (VALU bound) 8
    const bool heavy = n_leaves > 1 || (agg_col && (agg & ~A_ISUM) != 0) || width == 2;
    const bool very_heavy = width == 1 || (width == 4 && agg_col && (agg & A_FSUM) != 0);
    const int want = ctx->blocks_per_cu > 0 ? grid : ctx->n_cus * (very_heavy ? 8 : heavy ? 3 : 2);
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
    if (slot != 15) prof_mark(ctx, 0);
    hipError_t e = launch_scan(ctx->stream, width, n_leaves, agg_col ? agg : 0, sep, g, a);
    if (slot != 15) prof_mark(ctx, 1);
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

static int run_reduce(knox_ctx *ctx, const knox_col *col, const uint32_t *mask, unsigned agg, size_t n_rows, uint64_t row_base,
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
static int run_moments(knox_ctx *ctx, const knox_col *col, const uint32_t *mask, size_t n_rows, int slot, int grid) {
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

// bytes REGEXP: compile on the host, stage the DFA image in ctx->bytes_val, one lane per row
static int run_regexp(knox_ctx *ctx, const knox_col *col, const uint8_t *pat, uint64_t pat_len, const uint8_t *mask, uint8_t *bits,
                      SkipPlan *skip = nullptr) {
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

// one leaf -> its own bitset (multi-pass plan)
static uint64_t order_key(int dtype, uint64_t bits);

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
        const size_t pool_off = (sp.size() * 8 + 63) & ~(size_t)63;
        int rc = ensure(ctx, ctx->bytes_val, pool_off + pool.size() + 64);
        if (rc) return rc;
        if (!sp.empty()) HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, sp.data(), sp.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync((char *)ctx->bytes_val.p + pool_off, pool.data(), pool.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vectors not retained
        hipError_t e = launch_bytes_inset(ctx->stream, p.op == KNOX_NOTIN, bytes_view(col, n_rows, skip), (const uint64_t *)ctx->bytes_val.p,
                                          (uint32_t)sp.size(), (const uint8_t *)ctx->bytes_val.p + pool_off, (uint8_t *)bits, nullptr);
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
        if (p.val_len) HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, p.val, (size_t)p.val_len, hipMemcpyHostToDevice, ctx->stream));
        if (len2) HIPCHK(ctx, hipMemcpyAsync((char *)ctx->bytes_val.p + off2, p.val2, len2, hipMemcpyHostToDevice, ctx->stream));
        if (p.val_len || len2) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host values not retained
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
        std::vector<uint64_t> keys;
        keys.reserve(cnt);
        const bool isf = dtype_is_float(col->dtype);
        const uint64_t m = wmask((int)w), sign = 1ull << (8 * w - 1), expm = w == 8 ? 0x7ff0000000000000ull : 0x7f800000ull;
        for (size_t i = 0; i < cnt; i++) {
            uint64_t v = 0;
            memcpy(&v, p.val + i * w, w);
            if (isf) {
                if ((v & (m ^ sign)) > expm) continue;
                keys.push_back(order_key(col->dtype, v));
            } else {
                keys.push_back(order_key(col->dtype, v));
            }
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        int rc = ensure(ctx, ctx->bytes_val, keys.size() * 8 + 16);
        if (rc) return rc;
        if (!keys.empty()) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->bytes_val.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vector not retained
        }
        hipError_t e = launch_inset(ctx->stream, col->width, dtype_flavor(col->dtype), p.op == KNOX_NOTIN, col->dev, n_rows,
                                    (const uint64_t *)ctx->bytes_val.p, (uint32_t)keys.size(), (uint8_t *)bits, nullptr);
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
    hipSetDevice(ctx->device);
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
    if (pack_skip && n_packs) {
        std::vector<uint32_t> live;
        live.reserve(n_packs);
        for (size_t p = 0; p < n_packs; p++) {
            if (pack_skip[p]) rows_pruned += std::min(pack_rows, n_rows - p * pack_rows);
            else live.push_back((uint32_t)p);
        }
        const size_t live_off = (n_packs + 15) & ~(size_t)15;
        rc = ensure(ctx, ctx->pack_skip, live_off + live.size() * 4 + 16);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->pack_skip.p, pack_skip, n_packs, hipMemcpyHostToDevice, ctx->stream));
        if (!live.empty())
            HIPCHK(ctx, hipMemcpyAsync((char *)ctx->pack_skip.p + live_off, live.data(), live.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host arrays are not retained
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
        HIPCHK(ctx, hipMemsetAsync(bits, 0, bits_padded(n_rows) + 64, ctx->stream));
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
                HIPCHK(ctx, hipMemsetAsync(dst, 0, bits_padded(n_rows) + 8, ctx->stream));
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
                                                           : hipMemsetAsync(dst, 0, bits_padded(n_rows) + 8, ctx->stream);
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
            const size_t pack_bytes = pack_rows >> 3;
            for (size_t p0 = 0; p0 < n_packs;) {
                if (!pack_skip[p0]) { p0++; continue; }
                size_t e = p0;
                while (e < n_packs && pack_skip[e]) e++;
                const size_t b0 = p0 * pack_bytes, b1 = std::min(e * pack_bytes, bits_bytes(n_rows));
                HIPCHK(ctx, hipMemsetAsync((uint8_t *)bits + b0, 0, b1 - b0, ctx->stream));
                p0 = e;
            }
            if (all_leaves_skipped) slot.rows_scanned = n_rows - rows_pruned;
        }
        // total count (slot 0 count word) from the final bitset
        unsigned long long *d_cnt = (unsigned long long *)((int64_t *)((AggStateDev *)ctx->states.p + 16));
        HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
        hipError_t e = launch_popcount(ctx->stream, bits, n_rows, d_cnt);
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
    slot.has_pack_counts = false;
    slot.n_packs = n_packs;
    if (want_pack_counts && n_packs && packs_done) {
        slot.has_pack_counts = true;
    } else if (want_pack_counts && n_packs) {
        char *res = direct ? (char *)slot.h_res : (char *)ctx->states.p;
        hipError_t e = per_pack_tiles
                           ? launch_pack_counts(ctx->stream, (const uint32_t *)ctx->tile_counts.p, ctx->tc_grid, ctx->tc_rounds, per_pack_tiles,
                                                n_packs, (int64_t *)(res + res_base))
                           : launch_popcount_segments(ctx->stream, bits, n_rows, pack_rows, (int64_t *)(res + res_base), n_packs);
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

// ---- resident bit-packed columns: scanned as they are --------------------------------------------------
struct knox_packed {
    knox_ctx *ctx;
    int dtype;
    size_t n_rows, pack_rows, n_packs;
    uint64_t *d_words;
    PackedPack *d_packs;
    size_t words_bytes;
};

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
    for (size_t i = 0; i < n_bufs; i++) {
        if (!bufs[i] || lens[i] < 2) return KNOX_E_INVALID;
        if (bufs[i][0] != 4) return KNOX_E_UNSUPPORTED;  // only bit-packed containers stay packed
        size_t o = 1;
        uint64_t minv = 0, l2 = 0, n = 0;
        int k = knox_uvarint(bufs[i] + o, lens[i] - o, &minv);
        if (k == 0) return KNOX_E_INVALID;
        o += (size_t)k;
        k = knox_uvarint(bufs[i] + o, lens[i] - o, &l2);
        if (k == 0 || l2 > 64) return KNOX_E_INVALID;
        o += (size_t)k;
        k = knox_uvarint(bufs[i] + o, lens[i] - o, &n);
        if (k == 0) return KNOX_E_INVALID;
        o += (size_t)k;
        const uint64_t sz = ((l2 * n + 63) & ~63ull) / 8;
        if (o + sz > lens[i]) return KNOX_E_RANGE;
        if (n > pack_rows || (i + 1 < n_bufs && n != pack_rows) || n == 0) return KNOX_E_INVALID;  // only the last pack may be short
        packs[i] = {words, minv, (uint32_t)n, (uint32_t)l2};
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
    hipError_t e = launch_packed_scan(ctx->stream, grid, pa);
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
KNOX_API int64_t knox_cmp(knox_ctx *ctx, int dtype, int op, const void *src, size_t n, const void *a, const void *b, uint8_t *bits) {
    if (!ctx) return KNOX_E_INVALID;
    if (n == 0 || src == nullptr) return 0;  // cmp/tests/int64.go:212-213
    const int w = dtype_width(dtype);
    if (w == 0 || !bits || !a || (op == KNOX_RANGE && !b)) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    uint64_t av = 0, bv = 0;
    memcpy(&av, a, (size_t)w);
    if (b) memcpy(&bv, b, (size_t)w);
    LeafHost lh;
    int rc = make_leaf(dtype, op, av, bv, lh);
    if (rc) return rc;
    rc = ensure(ctx, ctx->tmp_vals, n * (size_t)w + 512);
    if (!rc) rc = ensure(ctx, ctx->tmp_bits, (bits_padded(n) + 64) * 8);
    const int grid = std::max(1, ctx->n_cus * (ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 8));
    if (!rc) rc = ensure(ctx, ctx->partials, sizeof(AggPartial) * (size_t)(grid + 1) * 16);
    if (!rc) rc = ensure(ctx, ctx->states, sizeof(AggStateDev) * 16 + 8 * 16);
    if (rc) return rc;
    rc = upload(ctx, ctx->tmp_vals.p, src, n * (size_t)w);
    if (rc) return rc;
    knox_col tmp;
    tmp.ctx = ctx; tmp.dtype = dtype; tmp.width = w; tmp.n_rows = n; tmp.pack_rows = n; tmp.dev = ctx->tmp_vals.p;
    tmp.buf = nullptr; tmp.buf_len = 0; tmp.owned = false;
    const knox_col *lc[1] = {&tmp};
    const uint64_t tt = lh.neg ? 0x1ull : 0x2ull;
    rc = run_fused(ctx, w, &lh, lc, 1, tt, nullptr, 0, n, n, 0, nullptr, (uint32_t *)ctx->tmp_bits.p, 0, grid, dtype);
    if (rc) return rc;
    int64_t cnt = 0;
    HIPCHK(ctx, hipMemcpyAsync(&cnt, (int64_t *)((AggStateDev *)ctx->states.p + 16), 8, hipMemcpyDeviceToHost, ctx->stream));
    rc = download(ctx, bits, ctx->tmp_bits.p, bits_bytes(n));  // exactly (n+7)/8 bytes: poison tail untouched
    if (rc) return rc;
    return cnt;
}

// ------------------------------------------------------------------------------------------------
// reducers
// ------------------------------------------------------------------------------------------------
KNOX_API int knox_reduce(knox_ctx *ctx, const knox_col *col, const knox_bits *mask, uint64_t row_base, knox_agg_state *state) {
    if (!ctx || !col || !state || col->dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (mask && mask->size != col->n_rows) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
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

// windows -> states on the device: each window's row range [d_start[k], d_start[k+1]) of `vals` is cut into work
// items (planned on the device: no host round trip of the bounds), reduced, and merged per window in row order.
// *states_dev = AggStateDev[n_buckets] inside ctx->tmp_vals.  n_rows bounds the item count.
static int reduce_windows(knox_ctx *ctx, const void *vals, int dtype, const uint32_t *mask, size_t n_rows, uint64_t row_base,
                          const uint64_t *d_start, size_t n_buckets, int flags, char **states_dev) {
    const int width = dtype_width(dtype);
    const uint32_t chunk_shift = 16;
    const size_t max_items = n_buckets + (n_rows >> chunk_shift) + 1;
    if (max_items > 0xfffffff0ull) return KNOX_E_RANGE;
    uint32_t plan_per = 0;
    const uint32_t plan_blocks = window_plan_blocks((uint32_t)n_buckets, &plan_per);
    const size_t off_istart = 0, off_bsum = ((n_buckets + 1) * 4 + 63) & ~(size_t)63;
    const size_t off_items = (off_bsum + ((size_t)plan_blocks + 1) * 4 + 63) & ~(size_t)63;
    const size_t off_partials = (off_items + max_items * sizeof(BucketItem) + 63) & ~(size_t)63;
    const size_t off_states = (off_partials + max_items * sizeof(AggPartial) + 63) & ~(size_t)63;
    int rc = ensure(ctx, ctx->tmp_vals, off_states + n_buckets * sizeof(AggStateDev) + 64);
    if (rc) return rc;
    char *base = (char *)ctx->tmp_vals.p;
    uint32_t *d_istart = (uint32_t *)(base + off_istart);
    hipError_t e = launch_window_plan(ctx->stream, d_start, (uint32_t)n_buckets, chunk_shift, (uint32_t *)(base + off_bsum), d_istart);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    uint32_t n_items32 = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n_items32, d_istart + n_buckets, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n_items = n_items32;
    if (n_items > max_items) return KNOX_E_RANGE;  // cannot happen: the bound above is exact arithmetic
    e = launch_window_items(ctx->stream, d_start, d_istart, (uint32_t)n_buckets, chunk_shift, n_items32, (BucketItem *)(base + off_items));
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    ScanArgs a;
    memset(&a, 0, sizeof a);
    a.agg_col = vals;
    a.agg_flavor = dtype_flavor(dtype);
    a.mask = mask;
    a.n_rows = n_rows;
    a.row_base = row_base;
    a.partials = (AggPartial *)(base + off_partials);
    const int grid = (int)std::min<size_t>(n_items, (size_t)ctx->n_cus * 8);
    // many small windows (a few hundred rows each): one wave per item instead of one workgroup per item
    const bool small_items = n_items > (size_t)ctx->n_cus * 8 && n_rows / n_items < 2048;
    if (small_items)
        e = launch_bucket_reduce_waves(ctx->stream, width, (int)std::min<size_t>((n_items + 3) / 4, (size_t)ctx->n_cus * 16), a,
                                       (const BucketItem *)(base + off_items), (uint32_t)n_items);
    else
        e = launch_bucket_reduce(ctx->stream, width, grid, a, (const BucketItem *)(base + off_items), (uint32_t)n_items);
    if (e == hipSuccess)
        e = launch_bucket_finalize(ctx->stream, width, a.partials, (const uint32_t *)(base + off_istart), (uint32_t)n_buckets, vals, row_base, dtype,
                                   (AggStateDev *)(base + off_states));
    if (e == hipSuccess && (flags & 2)) {
        // var/std: second pass over the same work items, each centred on its window's mean
        rc = ensure(ctx, ctx->mom_partials, (n_items + 1) * 4 * sizeof(double));
        if (rc) return rc;
        e = launch_bucket_moments(ctx->stream, width, a.agg_flavor, grid, vals, a.mask, (const BucketItem *)(base + off_items), (uint32_t)n_items,
                                  (const uint32_t *)(base + off_istart), (uint32_t)n_buckets, (AggStateDev *)(base + off_states),
                                  (double *)ctx->mom_partials.p);
    }
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    *states_dev = base + off_states;
    return KNOX_OK;
}

// out-of-order timestamps (bucket_native.go:123-141 finds / inserts the window's reducer by time): the selected rows
// are brought into window order by a STABLE radix sort of (window index, row id, value) triples (knox_sort.hip) -- rows
// of one window keep their stream order, which is all the reducers can observe -- and the ordered pipeline runs on the
// sorted copy of the values; the row ids inside the states are mapped back through the sorted row ids.
static int reduce_buckets_unordered(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                    int64_t step, size_t n_buckets, uint64_t row_base, int flags, char **states_dev) {
    const size_t n_rows = value_col->n_rows;
    if (n_rows > 0xfffffff0ull) return KNOX_E_UNSUPPORTED;
    const size_t w = (size_t)value_col->width;
    unsigned key_bits = 1;
    while ((1ull << key_bits) < n_buckets) key_bits++;  // keys 0..n_buckets-1 (rows outside every window leave in the first pass)
    uint32_t bits[4] = {0, 0, 0, 0};
    int n_passes = 0;
    window_sort_passes(key_bits, bits, &n_passes);
    const WindowSortGeom g0 = window_sort_geom(n_rows, ctx->n_cus);
    // scratch (idx_scratch): window bounds | histogram | digit totals;  idx_rows: keys of all rows;  sort_buf: the triples
    const size_t off_hist = ((n_buckets + 2) * 8 + 255) & ~(size_t)255;
    const size_t off_tot = off_hist + (((size_t)g0.nb * 256 * 4 + 255) & ~(size_t)255);
    int rc = ensure(ctx, ctx->idx_scratch, off_tot + 256 * 4 + 64);
    const size_t keys0_bytes = (n_rows * 4 + 255) & ~(size_t)255;
    if (!rc) rc = ensure(ctx, ctx->idx_rows, keys0_bytes);
    if (rc) return rc;
    char *sa = (char *)ctx->idx_scratch.p;
    uint64_t *d_start = (uint64_t *)sa;
    uint32_t *d_hist = (uint32_t *)(sa + off_hist), *d_tot = (uint32_t *)(sa + off_tot);
    hipError_t e = launch_window_sort_begin(ctx->stream, (const int64_t *)time_col->dev, n_rows, mask ? (const uint8_t *)mask->dev : nullptr, t_from,
                                            step, (uint32_t)n_buckets, g0, bits[0], (uint32_t *)ctx->idx_rows.p, d_hist, d_tot);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    uint32_t tot[256];
    HIPCHK(ctx, hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    size_t n_sel = 0;
    for (uint32_t t : tot) n_sel += t;
    // the triples: two sets when there is more than one pass
    const size_t arr4 = (n_sel * 4 + 255) & ~(size_t)255, arrw = (n_sel * w + 255) & ~(size_t)255;
    const size_t set = 2 * arr4 + arrw;
    rc = ensure(ctx, ctx->sort_buf, (n_passes > 1 ? 2 : 1) * set + 256);
    if (rc) return rc;
    const char *sb = (const char *)ctx->idx_rows.p;
    char *cur = (char *)ctx->sort_buf.p, *oth = cur + set;
    auto K = [&](char *b) { return (uint32_t *)b; };
    auto I = [&](char *b) { return (uint32_t *)(b + arr4); };
    auto V = [&](char *b) { return (void *)(b + 2 * arr4); };
    e = launch_window_sort_first(ctx->stream, (int)w, (const uint32_t *)sb, value_col->dev, n_rows, g0, bits[0], d_hist, d_tot, K(cur), I(cur), V(cur));
    const WindowSortGeom g1 = window_sort_geom(n_sel, ctx->n_cus);  // never more workgroups than g0: the histogram fits
    uint32_t shift = bits[0];
    for (int p = 1; p < n_passes && e == hipSuccess; p++) {
        e = launch_window_sort_pass(ctx->stream, (int)w, K(cur), I(cur), V(cur), n_sel, g1, shift, bits[p], d_hist, d_tot, K(oth), I(oth), V(oth));
        std::swap(cur, oth);
        shift += bits[p];
    }
    if (e == hipSuccess) e = launch_key_bounds(ctx->stream, K(cur), n_sel, (uint32_t)n_buckets, d_start);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    char *st = nullptr;
    rc = reduce_windows(ctx, V(cur), value_col->dtype, nullptr, n_sel, 0, d_start, n_buckets, flags, &st);
    if (rc) return rc;
    e = launch_remap_rows(ctx->stream, (AggStateDev *)st, (uint32_t)n_buckets, I(cur), row_base);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    *states_dev = st;
    return KNOX_OK;
}

// the windows' states on the device (ordered time: reduced in place; out-of-order time: the sorted copy)
static int reduce_buckets_dev(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                              int64_t step, size_t n_buckets, uint64_t row_base, int flags, char **states_dev) {
    if (!ctx || !time_col || !value_col || step <= 0 || n_buckets == 0 || n_buckets > (1u << 26)) return KNOX_E_INVALID;
    if (time_col->dtype != KNOX_INT64 || value_col->dtype == KNOX_BYTES || time_col->n_rows != value_col->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != value_col->n_rows) return KNOX_E_INVALID;
    // every window edge must be representable
    if ((__int128)t_from + (__int128)step * (__int128)n_buckets > (__int128)INT64_MAX) return KNOX_E_RANGE;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    const size_t n_rows = value_col->n_rows;
    int rc = ensure(ctx, ctx->misc, 256);
    if (!rc) rc = ensure(ctx, ctx->idx_scratch, (n_buckets + 2) * 8 + 64);
    if (rc) return rc;
    // 1. order check (one streaming pass over the time column) unless the caller vouches for it
    uint32_t *d_flag = (uint32_t *)((char *)ctx->misc.p + 32);
    uint32_t unsorted = 0;
    if (!(flags & 1)) {
        HIPCHK(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
        hipError_t e = launch_sorted_check(ctx->stream, (const int64_t *)time_col->dev, n_rows, d_flag);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
        HIPCHK(ctx, hipMemcpyAsync(&unsorted, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    // 2. first row of every window edge (stays on the device)
    uint64_t *d_start = (uint64_t *)ctx->idx_scratch.p;
    hipError_t e = launch_bucket_bounds(ctx->stream, (const int64_t *)time_col->dev, n_rows, t_from, step, n_buckets, d_start);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    if (!(flags & 1)) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (unsorted) return reduce_buckets_unordered(ctx, time_col, value_col, mask, t_from, step, n_buckets, row_base, flags, states_dev);
    // 3. reduce the windows' row ranges
    return reduce_windows(ctx, value_col->dev, value_col->dtype, mask ? (const uint32_t *)mask->dev : nullptr, n_rows, row_base, d_start, n_buckets,
                          flags, states_dev);
}

KNOX_API int knox_reduce_buckets(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                 int64_t step, size_t n_buckets, uint64_t row_base, int flags, knox_agg_state *states) {
    if (!states) return KNOX_E_INVALID;
    char *st = nullptr;
    int rc = reduce_buckets_dev(ctx, time_col, value_col, mask, t_from, step, n_buckets, row_base, flags, &st);
    if (rc) return rc;
    return download(ctx, states, st, n_buckets * sizeof(AggStateDev));
}

// The mergeable states themselves into the context's page-locked result buffer (a view, like knox_bits_indexes_view):
// for hosts that merge windows across GPUs.  144 B per window cross PCIe at DMA rate instead of through pageable memory.
KNOX_API int knox_reduce_buckets_view(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                      int64_t step, size_t n_buckets, uint64_t row_base, int flags, const knox_agg_state **states_view) {
    if (!states_view) return KNOX_E_INVALID;
    char *st = nullptr;
    int rc = reduce_buckets_dev(ctx, time_col, value_col, mask, t_from, step, n_buckets, row_base, flags, &st);
    if (rc) return rc;
    const size_t bytes = n_buckets * sizeof(AggStateDev);
    rc = ensure_host(ctx, ctx->h_view, ctx->h_view_cap, bytes);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_view, st, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *states_view = (const knox_agg_state *)ctx->h_view;
    return KNOX_OK;
}

// Reducer.Value() of every window, finalised on the device: 8 (value) + 1 (seen) bytes per window cross PCIe into the
// context's page-locked result buffer; the caller gets views (valid until the next *_view / *_values call on the ctx)
KNOX_API int knox_reduce_buckets_values(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask, int64_t t_from,
                                        int64_t step, size_t n_buckets, int func, int flags, const void **values_view,
                                        const uint8_t **seen_view) {
    if (!values_view || !seen_view || func < KNOX_R_SUM || func > KNOX_R_COUNT) return KNOX_E_INVALID;
    if (func == KNOX_R_VAR || func == KNOX_R_STD) flags |= 2;
    char *st = nullptr;
    int rc = reduce_buckets_dev(ctx, time_col, value_col, mask, t_from, step, n_buckets, 0, flags, &st);
    if (rc) return rc;
    const size_t w = (size_t)value_col->width;
    const size_t off_seen = (n_buckets * w + 63) & ~(size_t)63, bytes = off_seen + n_buckets;
    rc = ensure(ctx, ctx->pack_counts, bytes + 64);
    if (!rc) rc = ensure_host(ctx, ctx->h_view, ctx->h_view_cap, bytes);
    if (rc) return rc;
    char *d = (char *)ctx->pack_counts.p;
    hipError_t e = launch_window_values(ctx->stream, (int)w, (const AggStateDev *)st, (uint32_t)n_buckets, func, d, (uint8_t *)d + off_seen);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_view, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *values_view = ctx->h_view;
    *seen_view = (const uint8_t *)ctx->h_view + off_seen;
    return KNOX_OK;
}

// bytesRegexpMatcher.MatchVector (match_bytes.go:587-609): bits.Set(i) for every (masked) row the pattern matches
KNOX_API int knox_match_regexp(knox_ctx *ctx, const knox_col *col, const uint8_t *pattern, uint64_t pattern_len, const knox_bits *mask,
                               knox_bits *bits, int64_t *count) {
    if (!ctx || !col || !bits || bits->size != col->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != col->n_rows) return KNOX_E_INVALID;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
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

KNOX_API int knox_reduce_buckets_sequential(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col, const knox_bits *mask,
                                            int64_t t_from, int64_t step, size_t n_buckets, int func, int flags, void *values_out,
                                            uint8_t *seen_out) {
    if (!ctx || !time_col || !value_col || !values_out || !seen_out || step <= 0 || n_buckets == 0 || n_buckets > (1u << 26)) return KNOX_E_INVALID;
    if (time_col->dtype != KNOX_INT64 || value_col->dtype == KNOX_BYTES || time_col->n_rows != value_col->n_rows) return KNOX_E_INVALID;
    if (mask && mask->size != value_col->n_rows) return KNOX_E_INVALID;
    if (func < 0 || func > KNOX_R_STD_JOIN) return KNOX_E_INVALID;
    if ((__int128)t_from + (__int128)step * (__int128)n_buckets > (__int128)INT64_MAX) return KNOX_E_RANGE;
    if (ctx->res_head != ctx->res_tail) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    const size_t n_rows = value_col->n_rows;
    const size_t off_vals = ((n_buckets + 2) * 8 + 63) & ~(size_t)63;
    const size_t off_seen = off_vals + ((n_buckets * 8 + 63) & ~(size_t)63);
    int rc = ensure(ctx, ctx->misc, 256);
    if (!rc) rc = ensure(ctx, ctx->idx_scratch, off_seen + n_buckets + 64);
    if (rc) return rc;
    uint32_t *d_flag = (uint32_t *)((char *)ctx->misc.p + 32);
    if (!(flags & 1)) {
        HIPCHK(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
        hipError_t e = launch_sorted_check(ctx->stream, (const int64_t *)time_col->dev, n_rows, d_flag);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    }
    char *base = (char *)ctx->idx_scratch.p;
    uint64_t *d_start = (uint64_t *)base;
    hipError_t e = launch_bucket_bounds(ctx->stream, (const int64_t *)time_col->dev, n_rows, t_from, step, n_buckets, d_start);
    // the window bounds never leave the device: the reducer lanes read them in place
    if (e == hipSuccess)
        e = launch_bucket_sequential(ctx->stream, value_col->dtype, func, (const int64_t *)time_col->dev, value_col->dev,
                                     mask ? (const uint8_t *)mask->dev : nullptr, d_start, (uint32_t)n_buckets, base + off_vals,
                                     (uint8_t *)(base + off_seen));
    if (e != hipSuccess) { ctx->last_hip = (int)e; return KNOX_E_HIP; }
    uint32_t unsorted = 0;
    if (!(flags & 1)) {
        HIPCHK(ctx, hipMemcpyAsync(&unsorted, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (unsorted) return KNOX_E_UNSUPPORTED;
    }
    rc = download(ctx, values_out, base + off_vals, n_buckets * (size_t)value_col->width);
    if (!rc) rc = download(ctx, seen_out, base + off_seen, n_buckets);
    return rc;
}

KNOX_API int knox_reduce_sequential(knox_ctx *ctx, const knox_col *col, const knox_bits *mask, int func, void *value_out, int *seen,
                                    int64_t *row) {
    if (!ctx || !col || !value_out || col->dtype == KNOX_BYTES) return KNOX_E_INVALID;
    if (mask && mask->size != col->n_rows) return KNOX_E_INVALID;
    if (func < 0 || func > 8) return KNOX_E_INVALID;
    hipSetDevice(ctx->device);
    int rc = ensure(ctx, ctx->misc, 256);
    if (rc) return rc;
    uint64_t *d = (uint64_t *)((char *)ctx->misc.p + 192);
    hipError_t e = launch_reduce_sequential(ctx->stream, col->dtype, func, col->dev, col->n_rows, mask ? (const uint8_t *)mask->dev : nullptr, d);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    uint64_t h[3] = {0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(value_out, &h[0], (size_t)col->width);
    if (seen) *seen = (int)h[1];
    if (row) *row = h[2] == ~0ull ? -1 : (int64_t)h[2];
    return KNOX_OK;
}

// ---- Reducer.Value() from a merged state (pure host code) ------------------------------------
static void store_value(int dtype, void *out, uint64_t bits) { memcpy(out, &bits, (size_t)dtype_width(dtype)); }

// Reducer.Value(): the arithmetic lives in knox_finalize.h, shared with the device's per-window finaliser
KNOX_API int knox_agg_value(const knox_agg_state *st, int func, void *value_out, int *seen, int64_t *row) {
    if (!st || !value_out) return KNOX_E_INVALID;
    uint64_t bits = 0;
    int sn = 0;
    int64_t r = -1;
    const int rc = knox_fin::agg_value_bits(*st, func, &bits, &sn, &r);
    if (rc == -5) return KNOX_E_UNSUPPORTED;
    if (rc) return KNOX_E_INVALID;
    store_value(st->dtype, value_out, bits);
    if (seen) *seen = sn;
    if (row) *row = r;
    return KNOX_OK;
}

static uint64_t order_key(int dtype, uint64_t bits) {
    const int w = dtype_width(dtype);
    const uint64_t m = wmask(w), sign = 1ull << (8 * w - 1);
    bits &= m;
    if (dtype_is_float(dtype)) {
        if ((bits & (m ^ sign)) == 0) return host_fkey(0, w);  // +-0 share a key
        return host_fkey(bits, w);
    }
    return dtype_is_signed(dtype) ? (bits ^ sign) : bits;
}

// ---- NativeBucket.Emit's gap filling over dense per-window values (pure host code) --------------
// bucket_native.go:161-218 walks the windows from trange.From; a window without a reducer is filled by
// Fill(mode, step, last.Time(), next.Time(), lastVal, nextVal) (fill.go:59-85), where `last` starts as the
// FIRST reducer and `next` is the next reducer or, past the last one, `last` again -- quirks kept: a gap before
// the first window is "filled" from the first reducer, linear fill outside the reducers divides 0 by 0.
template <typename T>
static size_t fill_windows(int mode, const T *vals, const uint8_t *seen, size_t n, int64_t t_from, int64_t step, int64_t ups, size_t limit,
                           T *out, uint8_t *flags) {
    memset(flags, 0, n);
    size_t first = 0;
    while (first < n && !seen[first]) first++;
    if (first == n) return 0;  // no reducers: Emit writes nothing
    auto unix_s = [&](size_t k) {  // time.Time.Unix(): seconds, floored
        const int64_t t = t_from + (int64_t)k * step;
        int64_t q = t / ups;
        if ((t % ups != 0) && ((t < 0) != (ups < 0))) q--;
        return q;
    };
    size_t last = first, next_seen = first, count = 0;
    for (size_t k = 0; k < n && count < limit; k++) {
        while (next_seen < n && (next_seen < k || !seen[next_seen])) next_seen++;  // reducers[idx]
        const size_t next = next_seen < n ? next_seen : last;
        if (next != k) {
            switch (mode) {
            case KNOX_FILL_NULL: out[k] = 0; flags[k] = 2; count++; break;
            case KNOX_FILL_LAST: out[k] = vals[last]; flags[k] = 1; count++; break;
            case KNOX_FILL_ZERO: out[k] = 0; flags[k] = 1; count++; break;
            case KNOX_FILL_LINEAR: {
                // linearFill (fill.go:56-61): the slope between (prevTime, prevVal) and (nextTime, nextVal), in seconds
                const double m = (double)(T)(vals[next] - vals[last]) / (double)(unix_s(next) - unix_s(last));
                const double x = (double)(unix_s(k) - unix_s(last));
                const double b = (double)vals[last];
                out[k] = (T)(m * x + b);
                flags[k] = 1;
                count++;
                break;
            }
            default: break;  // none
            }
            continue;
        }
        out[k] = vals[k];
        flags[k] = 1;
        count++;
        last = k;
    }
    return count;
}

KNOX_API int knox_fill_windows(int dtype, int mode, const void *values, const uint8_t *seen, size_t n_buckets, int64_t t_from, int64_t step,
                               int64_t units_per_second, size_t limit, void *out_values, uint8_t *out_flags, size_t *n_out) {
    if (!values || !seen || !out_values || !out_flags || step <= 0 || units_per_second <= 0) return KNOX_E_INVALID;
    if (mode < KNOX_FILL_NONE || mode > KNOX_FILL_ZERO) return KNOX_E_INVALID;
    if ((__int128)t_from + (__int128)step * (__int128)n_buckets > (__int128)INT64_MAX) return KNOX_E_RANGE;
    size_t c;
#define KNOX_FILL(T) c = fill_windows<T>(mode, (const T *)values, seen, n_buckets, t_from, step, units_per_second, limit, (T *)out_values, out_flags); break
    switch (dtype) {
    case KNOX_INT64: KNOX_FILL(int64_t);
    case KNOX_INT32: KNOX_FILL(int32_t);
    case KNOX_INT16: KNOX_FILL(int16_t);
    case KNOX_INT8: KNOX_FILL(int8_t);
    case KNOX_UINT64: KNOX_FILL(uint64_t);
    case KNOX_UINT32: KNOX_FILL(uint32_t);
    case KNOX_UINT16: KNOX_FILL(uint16_t);
    case KNOX_UINT8: KNOX_FILL(uint8_t);
    case KNOX_FLOAT64: KNOX_FILL(double);
    case KNOX_FLOAT32: KNOX_FILL(float);
    default: return KNOX_E_INVALID;
    }
#undef KNOX_FILL
    if (n_out) *n_out = c;
    return KNOX_OK;
}

KNOX_API int knox_agg_combine(const knox_agg_state *states, int n, knox_agg_state *out) {
    if (!states || !out || n <= 0) return KNOX_E_INVALID;
    knox_agg_state r = states[0];
    // central moments: re-centre every part on the first non-empty part's shift K --
    //   sum (v-K) = D_i + n_i*d,  sum (v-K)^2 = Q_i + 2*d*D_i + n_i*d^2,  d = K_i - K  (binary128)
    bool mom_ok = true, mom_any = false;
    double mom_k = 0.0;
    __float128 mom_d = 0, mom_q = 0;
    for (int i = 0; i < n; i++) {
        const knox_agg_state &s = states[i];
        if (s.count <= 0) continue;
        if (!(s.flags & 2u)) { mom_ok = false; break; }
        if (!mom_any) { mom_k = s.m_shift; mom_any = true; }
        const __float128 d = (__float128)s.m_shift - (__float128)mom_k, ni = (__float128)s.count;
        const __float128 Di = (__float128)s.m_dev_hi + (__float128)s.m_dev_lo, Qi = (__float128)s.m_sq_hi + (__float128)s.m_sq_lo;
        mom_d += Di + ni * d;
        mom_q += Qi + 2 * d * Di + ni * d * d;
    }
    for (int i = 1; i < n; i++) {
        const knox_agg_state &s = states[i];
        if (s.dtype != r.dtype) return KNOX_E_INVALID;
        r.flags |= s.flags & 1u;
        r.count += s.count;
        r.isum += s.isum;
        {   // double-double add, same form as the device merge
            double hi = r.fsum_hi, lo = r.fsum_lo;
            double sum = hi + s.fsum_hi;
            double bb = sum - hi;
            double e = (hi - (sum - bb)) + (s.fsum_hi - bb);
            if (!(fabs(sum) <= 1.7976931348623157e308)) e = 0.0;
            r.fsum_lo = lo + s.fsum_lo + e;
            r.fsum_hi = sum;
        }
        if (s.min_row >= 0) {
            const uint64_t ks = order_key(r.dtype, s.min_bits), kr = order_key(r.dtype, r.min_bits);
            if (r.min_row < 0 || ks < kr || (ks == kr && s.min_row < r.min_row)) { r.min_bits = s.min_bits; r.min_row = s.min_row; }
        }
        if (s.max_row >= 0) {
            const uint64_t ks = order_key(r.dtype, s.max_bits), kr = order_key(r.dtype, r.max_bits);
            if (r.max_row < 0 || ks > kr || (ks == kr && s.max_row < r.max_row)) { r.max_bits = s.max_bits; r.max_row = s.max_row; }
        }
        if (s.first_row >= 0 && (r.first_row < 0 || s.first_row < r.first_row)) { r.first_row = s.first_row; r.first_bits = s.first_bits; }
        if (s.last_row >= 0 && (r.last_row < 0 || s.last_row > r.last_row)) { r.last_row = s.last_row; r.last_bits = s.last_bits; }
    }
    r.flags &= ~2u;
    r.m_shift = r.m_dev_hi = r.m_dev_lo = r.m_sq_hi = r.m_sq_lo = 0.0;
    if (mom_ok && mom_any) {
        r.flags |= 2u;
        r.m_shift = mom_k;
        r.m_dev_hi = (double)mom_d;
        r.m_dev_lo = (double)(mom_d - (__float128)r.m_dev_hi);
        r.m_sq_hi = (double)mom_q;
        r.m_sq_lo = (double)(mom_q - (__float128)r.m_sq_hi);
        if (!(fabs(r.m_dev_hi) <= 1.7976931348623157e308)) r.m_dev_lo = 0.0;
        if (!(fabs(r.m_sq_hi) <= 1.7976931348623157e308)) r.m_sq_lo = 0.0;
    }
    *out = r;
    return KNOX_OK;
}

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
    hipError_t e = hipMemsetAsync(b->d_filters, 0, b->bytes, ctx->stream);
    if (e == hipSuccess)
        e = col->dtype == KNOX_BYTES
                ? launch_bloom_build_bytes(ctx->stream, bytes_view(col, col->n_rows), b->d_filters, b->d_offsets)
                : launch_bloom_build(ctx->stream, col->width, col->dev, col->n_rows, col->pack_rows, b->d_filters, b->d_offsets);
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

// ------------------------------------------------------------------------------------------------
// RCCL: exchange of the final partial aggregates only (pack shards never move)
// ------------------------------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
    if (g_rccl.lib) return KNOX_OK;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return KNOX_E_COMM;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce || !g_rccl.CommDestroy) return KNOX_E_COMM;
    g_rccl.lib = lib;
    return KNOX_OK;
}

KNOX_API int knox_comm_unique_id(uint8_t id[KNOX_COMM_ID_BYTES]) {
    if (!id) return KNOX_E_INVALID;
    static_assert(sizeof(ncclUniqueId) == KNOX_COMM_ID_BYTES, "ncclUniqueId size");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return KNOX_E_COMM;
    memcpy(id, &u, sizeof u);
    return KNOX_OK;
}

KNOX_API int knox_comm_init(knox_ctx *ctx, int n_ranks, int rank, const uint8_t id[KNOX_COMM_ID_BYTES]) {
    if (!ctx || !id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return KNOX_E_INVALID;
    int rc = rccl_load();
    if (rc) return rc;
    hipSetDevice(ctx->device);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    if (g_rccl.CommInitRank(&ctx->comm, n_ranks, u, rank) != ncclSuccess) {
        ctx->comm = nullptr;
        return KNOX_E_COMM;
    }
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return KNOX_OK;
}

KNOX_API int knox_comm_allgather_states(knox_ctx *ctx, const knox_agg_state *mine, int n_states, knox_agg_state *merged, int64_t *count_inout) {
    if (!ctx || n_states < 0 || n_states > 14 || (n_states && (!mine || !merged))) return KNOX_E_INVALID;
    if (!ctx->comm) return KNOX_E_COMM;
    hipSetDevice(ctx->device);
    // payload per rank: n_states states + one int64 count
    const size_t per = sizeof(knox_agg_state) * (size_t)n_states + 8;
    const size_t total = per * (size_t)ctx->n_ranks;
    int rc = ensure_plain(ctx, ctx->comm_buf, per + total + 64);
    if (rc) return rc;
    std::vector<uint8_t> h(per + total);
    if (n_states) memcpy(h.data(), mine, sizeof(knox_agg_state) * (size_t)n_states);
    int64_t c = count_inout ? *count_inout : 0;
    memcpy(h.data() + per - 8, &c, 8);
    uint8_t *d_send = (uint8_t *)ctx->comm_buf.p;
    uint8_t *d_recv = d_send + ((per + 63) & ~(size_t)63);
    rc = ensure_plain(ctx, ctx->comm_buf, ((per + 63) & ~(size_t)63) + total + 64);
    if (rc) return rc;
    d_send = (uint8_t *)ctx->comm_buf.p;
    d_recv = d_send + ((per + 63) & ~(size_t)63);
    HIPCHK(ctx, hipMemcpyAsync(d_send, h.data(), per, hipMemcpyHostToDevice, ctx->stream));
    if (g_rccl.AllGather(d_send, d_recv, per, ncclUint8, ctx->comm, ctx->stream) != ncclSuccess) return KNOX_E_COMM;
    HIPCHK(ctx, hipMemcpyAsync(h.data() + per, d_recv, total, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // deterministic merge in rank order on every rank (rank order == global row order)
    std::vector<knox_agg_state> tmp((size_t)ctx->n_ranks);
    for (int s = 0; s < n_states; s++) {
        for (int r = 0; r < ctx->n_ranks; r++) memcpy(&tmp[(size_t)r], h.data() + per + per * (size_t)r + sizeof(knox_agg_state) * (size_t)s, sizeof(knox_agg_state));
        rc = knox_agg_combine(tmp.data(), ctx->n_ranks, &merged[s]);
        if (rc) return rc;
    }
    if (count_inout) {
        int64_t sum = 0;
        for (int r = 0; r < ctx->n_ranks; r++) {
            int64_t v;
            memcpy(&v, h.data() + per + per * (size_t)r + per - 8, 8);
            sum += v;
        }
        *count_inout = sum;
    }
    return KNOX_OK;
}

// The literal collective of the north star: ONE ncclAllReduce(ncclInt64, ncclSum) for the order-free integer
// aggregates {SumReducer (wrapping, reducer.go:168-178), CountReducer / CountResult}.  The 64-bit wrapping sum travels
// as two 32-bit halves in int64 lanes, so no lane can overflow for any number of ranks a node has and the result is
// exactly the sum mod 2^64 whatever the reduction order: bit-identical to knox_comm_allgather_states + combine.
KNOX_API int knox_comm_allreduce_sum_count(knox_ctx *ctx, uint64_t *isum_inout, int64_t *count_inout) {
    if (!ctx || !isum_inout || !count_inout) return KNOX_E_INVALID;
    if (!ctx->comm) return KNOX_E_COMM;
    hipSetDevice(ctx->device);
    int rc = ensure_plain(ctx, ctx->comm_buf, 64);
    if (rc) return rc;
    int64_t h[3] = {(int64_t)(*isum_inout & 0xffffffffull), (int64_t)(*isum_inout >> 32), *count_inout};
    int64_t *d = (int64_t *)ctx->comm_buf.p;
    HIPCHK(ctx, hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    if (g_rccl.AllReduce(d, d, 3, ncclInt64, ncclSum, ctx->comm, ctx->stream) != ncclSuccess) return KNOX_E_COMM;
    HIPCHK(ctx, hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *isum_inout = (uint64_t)h[0] + ((uint64_t)h[1] << 32);
    *count_inout = h[2];
    return KNOX_OK;
}

KNOX_API int knox_comm_destroy(knox_ctx *ctx) {
    if (!ctx) return KNOX_E_INVALID;
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return KNOX_OK;
}
