// knox_comm.cpp -- RCCL exchange of the final partial aggregates (pack shards never move).  librccl is dlopen'ed on the
// first knox_comm_* call so that a single-GPU host needs no RCCL at all.
#include "knox_internal.h"

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
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
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
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(lib, "ncclCommCount");
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
//
// The exchange has nothing to wait for on the device (its input is the host's copy of a collected scan), so it runs on a
// stream of its own behind page-locked slots: _start enqueues {H2D, ncclAllReduce, D2H, event} and returns, _finish
// reads the oldest one -- a host keeps the collective of step i in flight behind the scan of step i + 1.
static void comm_ring_drop(knox_ctx *ctx) {
    for (hipEvent_t &e : ctx->comm_ev)
        if (e) { hipEventDestroy(e); e = nullptr; }
    if (ctx->comm_h) hipHostFree(ctx->comm_h);
    if (ctx->comm_d) hipFree(ctx->comm_d);
    if (ctx->comm_stream) hipStreamDestroy(ctx->comm_stream);
    ctx->comm_h = nullptr;
    ctx->comm_d = nullptr;
    ctx->comm_stream = nullptr;
    ctx->comm_head = ctx->comm_tail = 0;
}

// All or nothing: `comm_stream` doubles as the "ring is ready" flag, so a failure half way (page-locked slots, device
// slots, an event) tears the partial state down again and the next call starts over instead of writing through a null
// slot pointer.
static int comm_ring_init(knox_ctx *ctx) {
    if (ctx->comm_stream && ctx->comm_h && ctx->comm_d && ctx->comm_ev[knox_ctx::kCommRing - 1]) return KNOX_OK;
    comm_ring_drop(ctx);
    bool ok = hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->comm_h, sizeof(int64_t) * 4 * knox_ctx::kCommRing, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->comm_d, sizeof(int64_t) * 4 * knox_ctx::kCommRing) == hipSuccess;
    for (int i = 0; ok && i < knox_ctx::kCommRing; i++) ok = hipEventCreateWithFlags(&ctx->comm_ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        ctx->last_hip = (int)hipGetLastError();
        comm_ring_drop(ctx);
        return KNOX_E_HIP;
    }
    return KNOX_OK;
}

KNOX_API int knox_comm_allreduce_sum_count_start(knox_ctx *ctx, uint64_t isum, int64_t count) {
    if (!ctx) return KNOX_E_INVALID;
    if (!ctx->comm) return KNOX_E_COMM;
    if (ctx->comm_head - ctx->comm_tail >= (uint64_t)knox_ctx::kCommRing) return KNOX_E_BUSY;
    hipSetDevice(ctx->device);
    int rc = comm_ring_init(ctx);
    if (rc) return rc;
    const int k = (int)(ctx->comm_head % knox_ctx::kCommRing);
    int64_t *h = ctx->comm_h + 4 * k, *d = ctx->comm_d + 4 * k;
    h[0] = (int64_t)(isum & 0xffffffffull);
    h[1] = (int64_t)(isum >> 32);
    h[2] = count;
    HIPCHK(ctx, hipMemcpyAsync(d, h, 3 * sizeof(int64_t), hipMemcpyHostToDevice, ctx->comm_stream));
    if (g_rccl.AllReduce(d, d, 3, ncclInt64, ncclSum, ctx->comm, ctx->comm_stream) != ncclSuccess) return KNOX_E_COMM;
    HIPCHK(ctx, hipMemcpyAsync(h, d, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->comm_stream));
    HIPCHK(ctx, hipEventRecord(ctx->comm_ev[k], ctx->comm_stream));
    ctx->comm_head++;
    return KNOX_OK;
}

KNOX_API int knox_comm_allreduce_sum_count_finish(knox_ctx *ctx, uint64_t *isum_out, int64_t *count_out) {
    if (!ctx || !isum_out || !count_out) return KNOX_E_INVALID;
    if (ctx->comm_head == ctx->comm_tail) return KNOX_E_INVALID;  // nothing in flight
    const int k = (int)(ctx->comm_tail % knox_ctx::kCommRing);
    HIPCHK(ctx, hipEventSynchronize(ctx->comm_ev[k]));
    const int64_t *h = ctx->comm_h + 4 * k;
    *isum_out = (uint64_t)h[0] + ((uint64_t)h[1] << 32);
    *count_out = h[2];
    ctx->comm_tail++;
    return KNOX_OK;
}

KNOX_API int knox_comm_allreduce_sum_count(knox_ctx *ctx, uint64_t *isum_inout, int64_t *count_inout) {
    if (!ctx || !isum_inout || !count_inout) return KNOX_E_INVALID;
    if (!ctx->comm) return KNOX_E_COMM;
    if (ctx->comm_head != ctx->comm_tail) return KNOX_E_BUSY;  // finish the exchanges in flight first
    const int rc = knox_comm_allreduce_sum_count_start(ctx, *isum_inout, *count_inout);
    if (rc) return rc;
    return knox_comm_allreduce_sum_count_finish(ctx, isum_inout, count_inout);
}

// the communicator's own view of its size (ncclCommCount) and this context's rank in it
KNOX_API int knox_comm_ranks(knox_ctx *ctx, int *n_ranks, int *rank) {
    if (!ctx || !n_ranks) return KNOX_E_INVALID;
    if (!ctx->comm || !g_rccl.CommCount) return KNOX_E_COMM;
    if (g_rccl.CommCount(ctx->comm, n_ranks) != ncclSuccess) return KNOX_E_COMM;
    if (rank) *rank = ctx->rank;
    return KNOX_OK;
}

KNOX_API int knox_comm_destroy(knox_ctx *ctx) {
    if (!ctx) return KNOX_E_INVALID;
    if (ctx->comm_stream) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->comm_stream);
    }
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    comm_ring_drop(ctx);
    ctx->comm = nullptr;
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return KNOX_OK;
}

