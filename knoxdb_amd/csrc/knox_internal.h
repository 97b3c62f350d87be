// knox_internal.h -- what the host-side translation units of libknoxgpu.so share: the handle structs behind
// include/knoxgpu.h (knox_ctx, knox_col, knox_bits), device / pinned-memory helpers, and the few internal entry points
// one unit needs from another.  Not installed; nothing here is exported (the library builds with -fvisibility=hidden).
#pragma once
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <new>
#include <vector>

#include <rccl/rccl.h>  // types only: librccl is dlopen'ed on first knox_comm_* call

#include "../../include/knoxgpu.h"
#include "knox_arena.h"
#include "knox_finalize.h"
#include "knox_kernels.h"

using namespace knox;

#define KNOX_API extern "C" __attribute__((visibility("default")))

static_assert(sizeof(knox_agg_state) == sizeof(AggStateDev), "knox_agg_state layout");

// ------------------------------------------------------------------------------------------------
// Host copies between the caller's pageable memory and the page-locked staging ring, spread over a few threads: one core
// moves 3-10 GB/s (less into memory it touches for the first time -- every 4 KiB page is a fault), the DMA engine 57 GB/s,
// so a large result (2^20 window states = 151 MB) or a pageable upload is bound by this copy.  One process-wide pool,
// started on first use; a copy below 1 MiB stays on the calling thread.  knox_copy_threads: KNOX_COPY_THREADS (default 8,
// 1 = the calling thread only).
// ------------------------------------------------------------------------------------------------
class CopyPool {
   public:
    static CopyPool &get() {
        static CopyPool pool;
        return pool;
    }
    void copy(void *dst, const void *src, size_t bytes) {
        if (bytes < ((size_t)1 << 20) || workers_.empty() || getpid() != pid_) {  // a forked child has no workers
            memcpy(dst, src, bytes);
            return;
        }
        std::lock_guard<std::mutex> one_at_a_time(call_mu_);
        const size_t parts = workers_.size() + 1;
        const size_t piece = ((bytes / parts) + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu_);
            dst_ = (char *)dst, src_ = (const char *)src, bytes_ = bytes, piece_ = piece;
            pending_ = (int)workers_.size();
            gen_++;
        }
        cv_.notify_all();
        run_piece(0);  // the caller takes the first piece
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
    }
    int threads() const { return (int)workers_.size() + 1; }

   private:
    CopyPool() : pid_(getpid()) {
        int n = 8;
        if (const char *e = getenv("KNOX_COPY_THREADS")) n = atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n > hw) n = (int)hw;
        for (int i = 1; i < n; i++) workers_.emplace_back([this, i] { loop(i); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    void run_piece(size_t i) {
        const size_t off = i * piece_;
        if (off < bytes_) memcpy(dst_ + off, src_ + off, std::min(piece_, bytes_ - off));
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            run_piece((size_t)id);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex call_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0, piece_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    const pid_t pid_;
};

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
    uint32_t defer_flags = 0; // ScanArgs.flags of this context's scans: KNOX_DEFER=0 at open -> 1 (bitset bursts tile by tile), =2 -> 2 (deferred, flushes not aligned in time); A/B of knox_scan.hip kDefer
    int blocks_per_cu = 0;  // 0 = auto: 2 workgroups/CU for pure streaming passes, 3 when the pass carries
                            // float/min-max aggregates or several leaves (profiles/r01_sweep_blocks_per_cu.txt)
#ifdef KNOX_DIAG
    int variant = 1;  // diagnostics builds only (make diag): 2 bitset assembled but not stored, 3 plain stores, 4 write-through stores, 5 XCD-contiguous tiles
#endif
    // staging ring (pinned) for host -> HBM uploads
    static constexpr size_t kStageBytes = 32u << 20;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    // ingest that overlaps the scans (knox_col_append_async & co.): a second stream for the H2D copies and a pinned ring of
    // its own for pageable sources; the scan stream waits on a column's `ready` event, never on the copy stream as a whole
    hipStream_t copy_stream = nullptr;
    void *cstage[2] = {nullptr, nullptr};
    hipEvent_t cstage_ev[2] = {nullptr, nullptr};
    bool cstage_used[2] = {false, false};
    int cstage_slot = 0;
    hipEvent_t fence_ev = nullptr;  // "every scan enqueued so far" for the copy stream (knox_col_reset)
    // grow-only device scratch
    DevBuf partials;     // AggPartial[(grid+1) * slots]
    DevBuf states;       // AggStateDev[slots] + int64 count
    DevBuf tmp_vals;     // knox_cmp: staged values
    DevBuf tmp_bits;     // scratch bitset(s)
    DevBuf tmp_bits_parked[2];  // verified scratch of the columns scanned before (ensure_for): switching back costs no probes
    int tmp_bits_park_next = 0;
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
    // the last numeric IN / NOT IN set as the device wants it (sorted unique order keys): a query runs the same set against every
    // run of its table, and sorting + uploading 64 Ki members took longer than matching 2^28 rows against them
    DevBuf inset_keys;
    // while a scan is being enqueued: room in its result slot's page-locked staging for the leaves' value bytes (behind the skip list),
    // so that their upload needs no stream synchronisation (knox_abi.cpp run_leaf_to_bits); null outside knox_scan_async
    // knox_reduce_buckets_values -> reduce_windows: where Reducer.Value() per window goes when the finalize kernel can write it itself
    void *win_values = nullptr;
    uint8_t *win_seen = nullptr;
    int win_func = 0;
    bool win_values_done = false;
    int win_path = 0;   // how the last window call reduced (knox_windows_last_path)
    char *leaf_stage = nullptr;
    size_t leaf_stage_off = 0, leaf_stage_cap = 0;
    std::vector<uint8_t> inset_raw;   // the caller's members as given: compared byte for byte before the prepared set is reused
    size_t inset_n = 0;
    int inset_dtype = -1;
    size_t inset_gstart_off = 0;      // byte offset of the radix bucket table behind the keys (0: none)
    uint32_t inset_buckets = 0, inset_shift = 0;
    // ring of pinned result mirrors: up to kResRing scans may be in flight (software
    // pipelining: enqueue scan i+1 before collecting scan i keeps the device busy)
    struct ResSlot {
        void *h_res = nullptr;
        size_t h_res_cap = 0;
        void *h_packs = nullptr;
        size_t h_packs_cap = 0;
        void *h_skip = nullptr;   // page-locked copy of the caller's pack_skip bytes + the live-pack list: the H2D copy needs no
        size_t h_skip_cap = 0;    // stream synchronisation (the slot outlives the scan), which a pruned point lookup would spend 30 us on
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
    void *bloom_flags = nullptr;  // page-locked overflow flags of a bloom build's batches
    size_t bloom_flags_cap = 0;
    int bloom_exact_batches = 0;  // batches of bloom builds that overflowed the one-read regions and were rebuilt exactly (diagnostics)
    void *cmp_res = nullptr;     // knox_cmp's page-locked result block (same layout as a slot's mirror)
    size_t cmp_res_cap = 0;
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
    int probe_launches = 0;    // launches and device time of the arena's probes (knox_arena_info)
    double probe_ms = 0.0;
    // RCCL
    void *rccl = nullptr;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    DevBuf comm_buf;
    // the sum / count all-reduce runs on a stream of its own behind a small ring of page-locked slots (knox_comm.cpp)
    static constexpr int kCommRing = 4;
    hipStream_t comm_stream = nullptr;
    int64_t *comm_h = nullptr, *comm_d = nullptr;  // 4 int64 per slot
    hipEvent_t comm_ev[kCommRing] = {};
    uint64_t comm_head = 0, comm_tail = 0;
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
    // asynchronous ingest (knox_col_create / _append_async / _upload_async): the last copy into this column as an event on the
    // context's copy stream; every reader makes the scan stream wait for it (col_acquire) until the host has seen it complete
    mutable hipEvent_t ready_ev = nullptr;
    mutable bool ready_pending = false;
    // time columns: what the order check of the window entry points found (-1 unknown, 0 out of order, 1 non-decreasing) for
    // the first `sorted_rows` rows.  A property of the column's CONTENT: kept only for columns the library owns and wrote
    // itself (never for knox_col_wrap), dropped by everything that writes into the column again (append, reset).
    mutable int8_t sorted_state = -1;
    mutable size_t sorted_rows = 0;
    // float columns: the largest / smallest finite non-zero biased float64 exponent of the values (-1 unknown), for the first `fexp_rows`
    // rows -- what the exact window sums size their fixed-point form by (knox_fold.hip window_fsum_kernel).  Kept and dropped like sorted_state.
    mutable int32_t fexp_max = -1, fexp_min = 0;
    mutable size_t fexp_rows = 0;
};
static inline int col_sorted_cached(const knox_col *c) { return (c->owned && c->sorted_state >= 0 && c->sorted_rows == c->n_rows) ? c->sorted_state : -1; }
static inline void col_sorted_set(const knox_col *c, bool sorted) {
    if (!c->owned) return;
    c->sorted_state = sorted ? 1 : 0;
    c->sorted_rows = c->n_rows;
}

static inline int col_fexp_cached(const knox_col *c) { return (c->owned && c->fexp_max >= 0 && c->fexp_rows == c->n_rows) ? c->fexp_max : -1; }
static inline void col_fexp_set(const knox_col *c, int emax, int emin) {
    if (!c->owned) return;
    c->fexp_max = emax;
    c->fexp_min = emin;
    c->fexp_rows = c->n_rows;
}

// a kernel on ctx->stream is about to read `col`: order it behind the column's last asynchronous upload
static inline int col_acquire(knox_ctx *ctx, const knox_col *col) {
    if (!col || !col->ready_pending) return KNOX_OK;
    if (hipEventQuery(col->ready_ev) == hipSuccess) {
        col->ready_pending = false;
        return KNOX_OK;
    }
    (void)hipGetLastError();  // hipErrorNotReady
    const hipError_t e = hipStreamWaitEvent(ctx->stream, col->ready_ev, 0);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

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

static inline int dtype_width(int dtype) {
    switch (dtype) {
    case KNOX_INT64: case KNOX_UINT64: case KNOX_FLOAT64: return 8;
    case KNOX_INT32: case KNOX_UINT32: case KNOX_FLOAT32: return 4;
    case KNOX_INT16: case KNOX_UINT16: return 2;
    case KNOX_INT8: case KNOX_UINT8: return 1;
    default: return 0;
    }
}
static inline bool dtype_is_float(int dtype) { return dtype == KNOX_FLOAT64 || dtype == KNOX_FLOAT32; }
static inline bool dtype_is_signed(int dtype) { return dtype >= KNOX_INT64 && dtype <= KNOX_INT8; }
static inline uint32_t dtype_flavor(int dtype) { return dtype_is_float(dtype) ? FL_FLOAT : dtype_is_signed(dtype) ? FL_SIGNED : FL_UNSIGNED; }

double arena_probe(void *user, const void *col, uint64_t rows, void *bits);
int arena_probe_multi(void *user, const void *col, uint64_t rows, void *const *bits, int n, double *ratios);

// Device memory of a context: the arena when placement is on (kind READ = columns / payloads the scans stream through,
// kind WRITE = bitsets / scratch they write), plain hipMalloc when it is off or the device has no virtual-memory API.
// The arena's budget is a hard cap.
static inline hipError_t dev_malloc(knox_ctx *ctx, void **p, size_t bytes, int kind = ARENA_READ) {
    if (ctx && ctx->placement) {
        const hipError_t e = ctx->arena.alloc(p, bytes, kind);
        if (e != hipErrorNotSupported) return e;
        (void)hipGetLastError();
    }
    return hipMalloc(p, bytes);
}
static inline void dev_free(knox_ctx *ctx, void *p) {
    if (!p) return;
    if (ctx && ctx->arena.release(p)) return;
    (void)hipFree(p);
}

static inline int ensure(knox_ctx *ctx, DevBuf &b, size_t bytes, int kind = ARENA_WRITE) {
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
int ensure_for(knox_ctx *ctx, DevBuf &b, size_t bytes, const knox_col *col);

// RCCL send / receive buffers stay ordinary hipMalloc memory (a few hundred bytes)
static inline int ensure_plain(knox_ctx *ctx, DevBuf &b, size_t bytes) {
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

static inline int ensure_host(knox_ctx *ctx, void *&p, size_t &cap, size_t bytes) {
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
// predicate planning types shared by the planner (knox_abi.cpp) and the packed-column scan (knox_containers.cpp)
// ------------------------------------------------------------------------------------------------
struct LeafHost {
    uint64_t lo = 0, diff = 0;
    bool neg = false;
    bool is_float = false;
};

static inline uint64_t wmask(int w) { return w == 8 ? ~0ull : ((1ull << (8 * w)) - 1); }

static inline uint64_t host_fkey(uint64_t u, int w) {
    if (w == 8) return u ^ ((uint64_t)((int64_t)u >> 63) | (1ull << 63));
    const uint32_t x = (uint32_t)u;
    return (uint64_t)(x ^ ((uint32_t)((int32_t)x >> 31) | 0x80000000u));
}

// the order key the device kernels compare in (floats: IEEE order key with +-0 folded, signed: sign flipped)
static inline uint64_t order_key(int dtype, uint64_t bits) {
    const int w = dtype_width(dtype);
    const uint64_t m = wmask(w), sign = 1ull << (8 * w - 1);
    bits &= m;
    if (dtype_is_float(dtype)) {
        if ((bits & (m ^ sign)) == 0) return host_fkey(0, w);  // +-0 share a key
        return host_fkey(bits, w);
    }
    return dtype_is_signed(dtype) ? (bits ^ sign) : bits;
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

// ------------------------------------------------------------------------------------------------
// internal entry points used across translation units (defined in knox_abi.cpp unless noted)
// ------------------------------------------------------------------------------------------------
void prof_mark(knox_ctx *ctx, int which);                                       // knox_prof_*: event before / after the timed kernel
int upload(knox_ctx *ctx, void *dst, const void *src, size_t bytes);            // host -> HBM through the pinned ring (in place from page-locked memory)
int download(knox_ctx *ctx, void *dst, const void *src, size_t bytes);
int copy_stream_init(knox_ctx *ctx);                                            // the context's copy stream (ingest that overlaps the scans)
int upload_async(knox_ctx *ctx, void *dst, const void *src, size_t bytes);      // enqueue only, on the copy stream
int col_new(knox_ctx *ctx, int dtype, size_t n_rows, size_t pack_rows, bool alloc, knox_col **out);
int bits_alloc(knox_ctx *ctx, size_t size_bits, const knox_col *for_col, knox_bits **out);
int make_leaf(int dtype, int op, uint64_t a, uint64_t b, LeafHost &out);
BytesView bytes_view(const knox_col *c, size_t n_rows, SkipPlan *skip = nullptr);
uint32_t pack_shift_for(size_t pack_rows, int width);
uint32_t pick_iters(knox_ctx *ctx, size_t n_rows, int width, size_t pack_rows_or_0, uint32_t dflt = 2);
int run_reduce(knox_ctx *ctx, const knox_col *col, const uint32_t *mask, unsigned agg, size_t n_rows, uint64_t row_base, int slot, int grid);
int run_moments(knox_ctx *ctx, const knox_col *col, const uint32_t *mask, size_t n_rows, int slot, int grid);
int run_regexp(knox_ctx *ctx, const knox_col *col, const uint8_t *pat, uint64_t pat_len, const uint8_t *mask, uint8_t *bits,
               SkipPlan *skip = nullptr);                                       // knox_regexp_host.cpp
