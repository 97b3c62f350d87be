/*
 * knoxgpu.h -- C ABI of libknoxgpu.so: the MI355X (gfx950) pack-scan path for KnoxDB.
 *
 * Drop-in boundary for ONE path of blockwatch-cc/knoxdb: predicate comparators
 * (internal/cmp) -> bitset combinators (internal/bitset) -> aggregate reducers
 * (internal/reducer), as entered through filter.Match (internal/operator/filter/
 * match_core.go:14).  Everything is plain C: pointers, sizes, integer codes.  A Go host
 * binds these with cgo (INTEGRATION.md shows the shim); tests bind them with ctypes.
 *
 * Citations are file:line relative to the reference checkout.
 *
 * Conventions
 *   - every function returns int: 0 = KNOX_OK, <0 = knox_err (knox_strerror names it),
 *     except knox_cmp which returns the match count (>=0) or a negative knox_err.
 *   - dtype codes are types.BlockType (internal/types/block.go:20-36); op codes are
 *     types.FilterMode (internal/types/mode.go:13-27); reducer codes follow
 *     internal/reducer/reducer.go:25-45.
 *   - bitsets are LSB-first: row i <-> byte i>>3, bit i&7 (internal/bitset/bitset.go:23-29);
 *     (n+7)/8 bytes; bits >= n in the last byte are 0; nothing is written past that byte
 *     (the reference's poison-tail test, internal/cmp/tests/tests.go:24-48).
 *   - host pointers passed in are never retained after the call returns (cgo rule);
 *     device memory is owned by the knox_* handle that allocated it.
 *   - one knox_ctx = one device + one HIP stream; a ctx is not re-entrant, use one ctx per
 *     host thread (the reference runs one goroutine per query, pack/table/query.go:195-219).
 */
#ifndef KNOXGPU_H
#define KNOXGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNOX_ABI_VERSION 7

/* ---- error codes ------------------------------------------------------------------- */
enum knox_err {
    KNOX_OK = 0,
    KNOX_E_INVALID = -1,     /* bad argument (null handle, unknown dtype/op, misaligned ptr) */
    KNOX_E_NODEVICE = -2,    /* no HIP device / HIP runtime failure at open */
    KNOX_E_HIP = -3,         /* a HIP call failed (knox_last_hip_error has the code)    */
    KNOX_E_NOMEM = -4,       /* host or device allocation failed                          */
    KNOX_E_UNSUPPORTED = -5, /* valid request this build does not implement               */
    KNOX_E_COMM = -6,        /* RCCL failure / not initialised                            */
    KNOX_E_RANGE = -7,       /* row/size out of range                                     */
    KNOX_E_BUSY = -8         /* result ring full: knox_scan_collect before the next async */
};

/* ---- types.BlockType (internal/types/block.go:20-36) -------------------------------- */
enum knox_dtype {
    KNOX_INT64 = 1, KNOX_INT32 = 2, KNOX_INT16 = 3, KNOX_INT8 = 4,
    KNOX_UINT64 = 5, KNOX_UINT32 = 6, KNOX_UINT16 = 7, KNOX_UINT8 = 8,
    KNOX_FLOAT64 = 9, KNOX_FLOAT32 = 10, KNOX_BOOL = 11, KNOX_BYTES = 12
};

/* ---- types.FilterMode (internal/types/mode.go:13-27) -------------------------------- */
enum knox_op {
    KNOX_EQ = 1, KNOX_NE = 2, KNOX_GT = 3, KNOX_GE = 4, KNOX_LT = 5, KNOX_LE = 6,
    KNOX_IN = 7, KNOX_NOTIN = 8 /* value in / not in a set: val = array of column-type values */,
    KNOX_RANGE = 9 /* Between [a,b] */,
    KNOX_REGEXP = 10 /* bytes columns: val = the pattern text (Go regexp / RE2 syntax) */,
    KNOX_TRUE = 11, KNOX_FALSE = 12
};

/* ---- reducer functions (internal/reducer/reducer.go:25-45) -------------------------- */
enum knox_reducer {
    KNOX_R_SUM = 0, KNOX_R_FIRST = 1, KNOX_R_LAST = 2, KNOX_R_MIN = 3, KNOX_R_MAX = 4,
    KNOX_R_MEAN = 5, KNOX_R_VAR = 6, KNOX_R_STD = 7, KNOX_R_COUNT = 8,
    /* join reducers (reducer.go:432-731): rows sharing one raw timestamp are summed before the
     * reducer proper sees them; only knox_reduce_buckets_sequential evaluates these           */
    KNOX_R_FIRST_JOIN = 9, KNOX_R_LAST_JOIN = 10, KNOX_R_MIN_JOIN = 11, KNOX_R_MAX_JOIN = 12,
    KNOX_R_MEAN_JOIN = 13, KNOX_R_VAR_JOIN = 14, KNOX_R_STD_JOIN = 15
};

typedef struct knox_ctx knox_ctx;   /* device + stream + workspace                       */
typedef struct knox_col knox_col;   /* one column resident in HBM (many packs, contiguous)*/
typedef struct knox_bits knox_bits; /* one device-resident bitset                         */

/* ---- library / context -------------------------------------------------------------- */
int knox_abi_version(void);
const char *knox_strerror(int err);
int knox_device_count(int *count);
int knox_last_hip_error(const knox_ctx *ctx); /* hipError_t of the last failing HIP call  */

/* Opens device `device`, creates the ctx's own stream and workspace.                    */
int knox_ctx_open(int device, knox_ctx **out);
/* Same, but launches on an existing hipStream_t (e.g. the caller's stream).              */
int knox_ctx_open_stream(int device, void *hip_stream, knox_ctx **out);
int knox_ctx_close(knox_ctx *ctx);
int knox_ctx_sync(knox_ctx *ctx);
void *knox_ctx_stream(knox_ctx *ctx); /* hipStream_t */
/* HIP-event stopwatch on the ctx stream (for achieved-GB/s measurement).                 */
int knox_timer_start(knox_ctx *ctx);
int knox_timer_stop(knox_ctx *ctx, float *elapsed_ms); /* records, syncs, returns ms      */
/* Tuning knob: workgroups per CU of the streaming kernels (0 = auto, <0 keeps the current value).  */
int knox_ctx_set_tuning(knox_ctx *ctx, int blocks_per_cu);
/* Page-locked host memory owned by the library (hipHostMalloc): the one kind of host memory the
 * library may be handed repeatedly without staging.  A Go host builds its block vectors in it
 * (unsafe.Slice over the pointer; arena hook, SURVEY.md 8b "Ownership") and knox_col_upload /
 * knox_bits_upload / the download calls then DMA straight from / into it at PCIe rate; any other
 * host pointer goes through the library's pinned staging ring.  Never retained after a call.  */
int knox_host_alloc(knox_ctx *ctx, size_t bytes, void **out);
int knox_host_free(knox_ctx *ctx, void *p);

/* ---- device memory: one bounded arena per context (knoxdb_amd/csrc/knox_arena.h) ---------------- *
 * Measured on MI355X: for the scan's read+write mix it matters where the bitset lies relative to the column.  Physical
 * HBM falls into a few classes; a scan whose column and bitset lie in the SAME class runs at ~6.0 TB/s, in DIFFERENT
 * classes at ~7.05 TB/s; read-only passes do not see it.  All device memory a context owns comes out of ONE arena:
 * physical memory is taken from the driver only as allocations need it (HIP virtual-memory API), up to a hard budget,
 * and is not returned before knox_ctx_close (or an explicit trim); every 1 GiB chunk is labelled with its class once
 * (probed against one reference window per class met so far) and mapped into that class's zone, so columns of >= 1 GiB
 * are class-pure; a large write buffer allocated FOR a column (knox_bits_alloc_for, the scan's scratch bitsets) comes
 * from the top of ANOTHER class's zone (else: spare or fresh physical allocations of its own) and is verified by running
 * the scan kernel over every GiB of that very (column, buffer)
 * pair.  Nothing is held outside the arena's accounting.  Budget: KNOX_ARENA_GIB in the environment,
 * knox_ctx_set_arena, default 3/4 of the device.  Mode 0 (or KNOX_PLACEMENT=0): plain hipMalloc / hipFree.
 * Results never depend on any of this.                                                                           */
int knox_ctx_set_placement(knox_ctx *ctx, int mode /* 0 plain allocations, 1 arena */);
int knox_ctx_set_arena(knox_ctx *ctx, size_t budget_bytes);
typedef struct knox_arena_info {
    uint64_t budget_bytes;     /* the cap                                                   */
    uint64_t mapped_bytes;     /* physical memory taken from the driver and kept so far     */
    uint64_t used_bytes;       /* handed out to columns / bitsets / scratch right now       */
    uint64_t peak_used_bytes;
    uint64_t spare_write_buffer_bytes; /* write buffers that were candidates or were freed: the first candidates of the next request */
    uint64_t mixed_zone_bytes;         /* chunks moved out of their class zones into one mixed-class region: columns that fit
                                          no single class zone although the arena had the room (classes interleaving every
                                          few GiB); 0 on every device seen so far                                        */
    int32_t write_buffers, chunks, probes; /* write buffers created, 1 GiB chunks created, probe measurements taken */
    int32_t enabled;           /* 0: plain allocations (mode 0 or no virtual-memory API)    */
    int32_t probe_launches;    /* kernel launches the probe measurements above took ...      */
    float probe_ms;            /* ... and the device time they took (HIP events), since the context was opened */
    int32_t unprobed_chunks;   /* chunks that took the label of the chunk before them, and ...                  */
    int32_t unverified_write_buffers; /* ... write buffers taken from another class's zone by label alone: the probe budget was spent */
} knox_arena_info;
int knox_ctx_arena_info(knox_ctx *ctx, knox_arena_info *out);
/* What the arena may spend on probe measurements from now on: at most max_launches kernel launches and max_ms of device time
 * (counted like probe_launches / probe_ms above).  A context starts with 200 launches / 50 ms -- placement buys 1-6 points of
 * the HBM peak on multi-GiB scans and must not cost more than that by itself; a long-lived host grants more when it has idle
 * time.  With the budget spent nothing is measured: new chunks inherit the previous chunk's class, write buffers are placed
 * by label.  (0, 0) = measure nothing at all.                                                                        */
int knox_ctx_set_probe_budget(knox_ctx *ctx, int max_launches, double max_ms);
/* Column space mapped per placement class (chunks labelled so far, used or free) and the number of classes met.
 * Diagnostics: a context whose classes_seen stays 1 while bitsets are placed has found one class of memory only.        */
int knox_ctx_arena_classes(knox_ctx *ctx, uint64_t class_bytes[4], int *classes_seen /* nullable */);
/* Room for columns that will be SCANNED TOGETHER (the columns of one table run, the leaves of one filter tree): grows the
 * arena until one placement class has `bytes` contiguous free and makes that class the preferred one of the following
 * column allocations.  Two read streams of one scan run 4-5 % faster when they lie in ONE class (the bitset they write in
 * ANOTHER one: knox_bits_alloc_for_cols).  Allocates nothing; without it columns simply follow the previous column's class
 * while it has room.  KNOX_E_NOMEM: the budget cannot hold that much in one class (allocations still work, anywhere).    */
int knox_ctx_reserve(knox_ctx *ctx, size_t bytes);
/* The placement class (0 .. 3) the column's values (which = 0; a bytes column: its pointer array) or its bytes pool
 * (which = 1) lie in; -1: small allocations, the mixed region, borrowed memory, placement off.  Diagnostics.            */
int knox_col_placement_class(const knox_col *col, int which);
/* Returns every FREE column chunk and every spare write buffer to the driver.  Explicit only: returning multi-GiB blocks
 * slows live neighbours for a second or two (the driver clears them).                                             */
int knox_ctx_arena_trim(knox_ctx *ctx, size_t *freed_bytes /* nullable */);
/* Bytes / regions the context keeps away from the driver OUTSIDE the arena's accounting: always 0 (kept so that a
 * host can assert it).                                                                                            */
int knox_ctx_placement_info(knox_ctx *ctx, int64_t *held_bytes, int *held_regions);
/* What knox_bits_alloc_for measured for this (column, bitset) pair when it chose the buffer: the scan of up to four
 * 1 GiB samples of the column into the bitset over the read-only form of the same samples (worst sample; >= 0.95 = a
 * fast pair; 0 = column below 1 GiB or bitset below 16 MiB: not measured) and the candidates it looked at.        */
int knox_bits_pair_check(const knox_bits *bits, double *scan_over_read_only, int *candidates);
/* Per-kernel stopwatch: while enabled, every launch of the dominant streaming kernel
 * (scan / reduce) is bracketed by HIP events on the ctx stream; knox_prof_read syncs, returns
 * the summed kernel time and launch count since the last read, and resets.                */
int knox_prof_enable(knox_ctx *ctx, int on);
int knox_prof_read(knox_ctx *ctx, double *total_ms, int *launches);
/* Same, but returns each launch's duration (up to cap entries, in launch order).             */
int knox_prof_read_each(knox_ctx *ctx, float *ms, int cap, int *launches);
/* Scan-kernel instantiations launched so far (process-wide) whose resident workgroups per CU, as the HIP runtime
 * reports them, are BELOW what the launch shape assumes (knox_kernels.h scan_shape: the deferred bitset bursts park LDS
 * sized for exactly that many).  0 on gfx950 with the shipped build; anything else means a changed kernel, table or part
 * and a measured loss (results stay correct).                                                                      */
int knox_scan_occupancy_mismatches(void);
/* The launch-shape table itself (diagnostics): workgroups per CU and parked bitset words per wave of the scan instantiation
 * for `width`-byte columns, n_leaves leaves, aggregates beyond the wrapping sum (heavy), a float aggregate column, a float
 * sum among the aggregates.  tests/test_scan_shapes.py holds it against the compiler's occupancy figures of the build.  */
int knox_scan_launch_shape(int width, int n_leaves, int heavy, int is_float, int float_sum, int *per_cu, uint32_t *park_words);

/* ---- finest boundary: the cmp func vars --------------------------------------------- *
 * Replaces `func(src []T, val T, bits []byte) int64` / `func(src []T, a, b T, bits []byte)
 * int64` (internal/cmp/cmp.go:6-114, rebound in cmp_amd64.go:14-218, called from
 * operator/filter/match_num.go:348-566).  HOST pointers; synchronous; `a`/`b` point to one
 * value of the column type (`b` only for KNOX_RANGE).  `bits` must hold (n+7)/8 bytes and be
 * zeroed by the caller (reader.go:334).  n == 0 or src == NULL returns 0 and writes nothing.
 * Stages src through pinned memory to HBM, runs the same kernels as knox_scan.            */
int64_t knox_cmp(knox_ctx *ctx, int dtype, int op, const void *src, size_t n, const void *a,
                 const void *b, uint8_t *bits);

/* ---- columns resident in HBM -------------------------------------------------------- *
 * A knox_col is the device image of one column of a run of packs: the decoded vectors
 * block.NewAccessor[T](b).Slice() (internal/block/access.go:87-91) of consecutive packs,
 * concatenated.  pack_rows is the reference's PackSize (a multiple of 64; the last pack may
 * be short).  Upload goes host -> pinned ring -> hipMemcpyAsync.                           */
int knox_col_upload(knox_ctx *ctx, int dtype, const void *host, size_t n_rows, size_t pack_rows,
                    knox_col **out);
/* Ingest that overlaps the scans -- the device form of the reference's load / match interleave
 * (internal/pack/table/reader.go:329-334 calls loadPack, :484, then filter.Match, pack by pack).  Here a RUN of packs is
 * copied host -> HBM on the context's copy stream while the scan of the previous run is in flight on the scan stream:
 *   knox_col_create        an empty numeric column with room for capacity_rows rows (n_rows = 0)
 *   knox_col_append_async  enqueue the copy of n_rows more rows behind the ones present and return.  A page-locked source
 *                          (knox_host_alloc) is read in place by the DMA engine and must stay untouched until
 *                          knox_col_wait / knox_col_ready says so; any other source is copied through a pinned ring
 *                          inside the call and is not referenced after return (cgo rule)
 *   knox_col_reset         n_rows = 0, to refill the column with the next run.  The copies enqueued afterwards wait for
 *                          every scan enqueued SO FAR (they may still read the old rows) and for nothing enqueued later:
 *                          call it before enqueueing the scan of the other buffer
 *   knox_col_upload_async  create + append in one call
 *   knox_col_wait / knox_col_ready   block until / ask whether (1 / 0) the column's uploads have landed
 * Every entry point that reads a column orders its kernels behind the column's pending upload ON THE DEVICE
 * (hipStreamWaitEvent), so `append_async(B); scan_async(A); collect(A)` overlaps B's copy with A's scan and
 * `append_async(A); scan_async(A)` is simply correct.  KNOX_E_RANGE: more rows than the capacity.                      */
int knox_col_create(knox_ctx *ctx, int dtype, size_t capacity_rows, size_t pack_rows, knox_col **out);
int knox_col_append_async(knox_ctx *ctx, knox_col *col, const void *host, size_t n_rows);
int knox_col_reset(knox_ctx *ctx, knox_col *col);
int knox_col_upload_async(knox_ctx *ctx, int dtype, const void *host, size_t n_rows, size_t pack_rows, knox_col **out);
int knox_col_wait(knox_ctx *ctx, knox_col *col);
int knox_col_ready(const knox_col *col);
/* Bytes column in StringPool layout (pkg/stringx/pool.go:45-58): ptr[i] = off<<32 | len.  */
int knox_col_upload_bytes(knox_ctx *ctx, const uint64_t *ptr, size_t n_rows, const uint8_t *buf,
                          size_t buf_len, size_t pack_rows, knox_col **out);
/* Bytes column held the way the reference holds it: ONE StringPool per pack (block) -- block.Bytes() of pack p is
 * its own pool, ptr[i] = off<<32 | len with off relative to THAT pool (pkg/stringx/pool.go:45-58; matchers run per
 * pool, match.go:20-51).  The column keeps the row pointers of all packs in one array and the pools concatenated in
 * one payload buffer with a 64-bit base per pack, so only a single pack's pool is limited to 4 GiB -- 2^30 rows of
 * 32-byte values (BASELINE configs[3]) are one resident column.  Create reserves room for max_packs packs of
 * pack_rows rows (a multiple of 64) and max_pool_bytes of payload in total; append_pack adds the next pack (host
 * slices, never retained): every pack but the last holds exactly pack_rows rows, a short pack ends the column.     */
int knox_col_bytes_create(knox_ctx *ctx, size_t pack_rows, size_t max_packs, size_t max_pool_bytes, knox_col **out);
int knox_col_bytes_append_pack(knox_ctx *ctx, knox_col *col, const uint64_t *ptr, size_t n_rows, const uint8_t *buf,
                               size_t buf_len);
/* Pack `pack` back as the reference would hold it: row pointers (relative to the pack's pool) and the pool.  Either
 * output may be NULL; *n_rows_out / *buf_len_out (nullable) report the sizes.  A single-pool column is pack 0.       */
int knox_col_bytes_download_pack(knox_ctx *ctx, const knox_col *col, size_t pack, uint64_t *ptr_out, size_t ptr_cap,
                                 size_t *n_rows_out, uint8_t *buf_out, size_t buf_cap, size_t *buf_len_out);
size_t knox_col_pool_bytes(const knox_col *col); /* payload bytes resident in HBM (bytes columns) */
/* Synthetic bytes column generated on the device (DESIGN.md "synthetic data", kind 5; SURVEY.md 8d cfg 4): every row
 * holds one of 2^window_bits * ceil(n_packs / packs_per_window) distinct keys of key_len bytes, clustered so that a
 * key only occurs inside its window of packs_per_window consecutive packs:
 *   key index of row i (pack p) = (p / packs_per_window) << window_bits | (r(i) & (2^window_bits - 1)),
 *   key k = the first key_len bytes of the little-endian words mix64(8k + j + mix64(seed ^ 0x6b657973)), j = 0, 1, ..
 * One pool per pack, rows stored in row order (ptr[i] = (i % pack_rows) * key_len << 32 | key_len).                  */
int knox_col_generate_bytes(knox_ctx *ctx, uint64_t seed, size_t n_rows, size_t pack_rows, uint32_t key_len,
                            uint32_t window_bits, uint32_t packs_per_window, knox_col **out);
/* Ingest the ON-DISK containers of consecutive packs without a host decode pass ("next" row
 * f2): bufs[i] is the container of pack i exactly as stored by the reference:
 *   raw ints    [7][uvarint n values][LE values]            internal/encode/int_raw.go:73-94
 *   raw floats  [15][uvarint n bytes][LE values]            internal/encode/float_raw.go:73-89
 *   bit-packed  [4][uvarint For][uvarint log2][uvarint N][packed bits]   int_bitpack.go:93-133
 *   constant    [1][uvarint Val][uvarint N]                 int_const.go:67-71   (floats: [10][LE bits][N])
 *   delta       [2][uvarint For][uvarint Delta][uvarint N]  int_delta.go:70-75   value i = T(i)*Delta + For
 *   simple8     [6][uvarint For][uvarint N][uvarint len][64-bit words]    int_s8b.go:87-115, s8b/generic/decode.go
 *   ALP         [13][uvarint Exponent][uvarint Factor][flags][int64|int32 values][patches][positions]  float_alp.go:109-165
 *   ALP-RD      [14][container of uint16 left parts][container of uint64|uint32 right parts][uvarint Shift]  float_alprd.go:80-110
 *   dictionary  [5|12][container of T][container of uint16 codes]         int_dict.go:76-103, float_dict.go:75-102
 *   run-end     [3|11][container of T][container of uint32 inclusive ends] int_runend.go:84-112, float_runend.go:80-110
 * (uvarint = pkg/num/varint.go, a SQLite4-style prefix varint).  Raw payloads are staged straight
 * into the HBM column; bit-packed payloads cross PCIe packed and are decoded ON THE DEVICE
 * (bitpack/decode.go:56-74).  Any other container type (128/256-bit, strings, bitmaps): KNOX_E_UNSUPPORTED (host decode as today). */
int knox_col_upload_raw_containers(knox_ctx *ctx, int dtype, const uint8_t *const *bufs, const size_t *lens,
                                   size_t n_bufs, size_t pack_rows, knox_col **out);
/* The same for bytes columns: one string container per pack -> one StringPool column.  Payloads are
 * uploaded as they are, the row pointers are built on the device from the decoded children:
 *   constant    [16][uvarint n][uvarint len][val]                          string_const.go:51-70
 *   fixed       [17][uvarint n][uvarint sz][n*sz bytes]                    string_fixed.go:59-82
 *   compact     [18][uint32 container ofs][uint32 container len][uvarint buflen][buf]   string_compact.go:64-97
 *   dictionary  [19][ofs][len][uint32 container code][uvarint dictlen][dict]            string_dict.go:70-108
 * A row that would leave its payload returns KNOX_E_RANGE (nothing on the device ever reads out of
 * bounds).  When every container but the last holds exactly pack_rows rows (a multiple of 64) each pack keeps
 * its own pool (see knox_col_bytes_create: no 4 GiB limit on the column); ragged containers share one pool of at
 * most 4 GiB (32-bit StringPool offsets).                                                       */
int knox_col_upload_string_containers(knox_ctx *ctx, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs,
                                      size_t pack_rows, knox_col **out);
/* Int128 / Int256 containers ([8][X0][X1], [9][X0][X1][X2][X3]: internal/encode/int128.go:115-137,
 * int256.go:121-150; the children are ordinary 64-bit integer containers of any type above) -> the
 * stride layout knox_match_wide takes: out_limbs[0..n_limbs) receive one 8-byte column per limb.  */
int knox_col_upload_wide_containers(knox_ctx *ctx, int n_limbs /* 2 | 4 */, const uint8_t *const *bufs, const size_t *lens,
                                    size_t n_bufs, size_t pack_rows, knox_col **out_limbs);
/* Borrow memory that already lives on the device (16-byte aligned); never freed by knox.  */
int knox_col_wrap(knox_ctx *ctx, int dtype, void *device_ptr, size_t n_rows, size_t pack_rows,
                  knox_col **out);
/* Synthetic column generated on the device (DESIGN.md "synthetic data"; counter based):
 * kind 0 uniform integer in [lo, lo+span) (span 0 = raw 64 bits), 1 float grid k*2^-10,
 * 2 float uniform [0,1), 3 row id + lo, 4 float mix with NaN/Inf/+-0.                    */
int knox_col_generate(knox_ctx *ctx, int dtype, int kind, uint64_t seed, uint64_t first_row,
                      size_t n_rows, int64_t lo, uint64_t span, size_t pack_rows, knox_col **out);
int knox_col_download(knox_ctx *ctx, const knox_col *col, size_t first_row, size_t n_rows, void *host);
int knox_col_free(knox_col *col);
size_t knox_col_rows(const knox_col *col);
int knox_col_dtype(const knox_col *col);
void *knox_col_device_ptr(const knox_col *col);

/* ---- device bitsets: internal/bitset/bitset.go:30-35 + generic/bitset.go:13-435 ------ */
int knox_bits_alloc(knox_ctx *ctx, size_t size_bits, knox_bits **out);
/* Borrowed device memory as a bitset (counterpart of knox_col_wrap): device_ptr is 128-byte aligned
 * and at least ((size_bits + 63) / 64) * 8 + 64 bytes long; tail bits must be zero; never freed
 * by the library.                                                                            */
int knox_bits_wrap(knox_ctx *ctx, void *device_ptr, size_t size_bits, knox_bits **out);
/* As knox_bits_alloc, but for scans over `col`: for columns of >= 1 GiB the buffer is a physical allocation of its own,
 * chosen among candidates by scanning samples of `col` into it (knox_bits_pair_check reports the measurement); works
 * for borrowed columns (knox_col_wrap) too.  Plain knox_bits_alloc takes ordinary arena memory, unverified.       */
int knox_bits_alloc_for(knox_ctx *ctx, size_t size_bits, const knox_col *col, knox_bits **out);
/* The same for a scan whose tree reads SEVERAL columns (<= 4: the leaves of one fused pass): the buffer is verified against
 * samples of every one of them and taken from a placement class none of them lies in when there is one.             */
int knox_bits_alloc_for_cols(knox_ctx *ctx, size_t size_bits, const knox_col *const *cols, int n_cols, knox_bits **out); /* zeroed           */
int knox_bits_free(knox_bits *b);
size_t knox_bits_size(const knox_bits *b);
void *knox_bits_device_ptr(const knox_bits *b);
int knox_bits_upload(knox_ctx *ctx, knox_bits *b, const uint8_t *host);
/* Bitmap container of a bool column -> bits (size must equal the container's N): [20] zero, [21] one,
 * [22] dense (internal/encode/bitmap.go:103-135); [23] sparse = a serialized xroar bitmap (internal/xroar:
 * key node + array / bitmap containers), decoded with bounds checks on the host into the dense form (it is chosen for
 * < 1 bit in 16 set, so the dense bytes are at most twice its size). */
int knox_bits_upload_bitmap_container(knox_ctx *ctx, knox_bits *bits, const uint8_t *buf, size_t len);  /* (size+7)/8 bytes */
int knox_bits_download(knox_ctx *ctx, const knox_bits *b, uint8_t *host);
int knox_bits_zero(knox_ctx *ctx, knox_bits *b);                         /* Bitset.Zero :533 */
int knox_bits_one(knox_ctx *ctx, knox_bits *b);                          /* Bitset.One  :520 */
int knox_bits_copy(knox_ctx *ctx, knox_bits *dst, const knox_bits *src);
int knox_bits_and(knox_ctx *ctx, knox_bits *dst, const knox_bits *src);     /* And    :13  */
int knox_bits_andnot(knox_ctx *ctx, knox_bits *dst, const knox_bits *src);  /* AndNot :115 */
int knox_bits_or(knox_ctx *ctx, knox_bits *dst, const knox_bits *src);      /* Or     :153 */
int knox_bits_xor(knox_ctx *ctx, knox_bits *dst, const knox_bits *src);     /* Xor    :257 */
int knox_bits_neg(knox_ctx *ctx, knox_bits *b);                             /* Neg    :295 */
/* AndFlag :51 / OrFlag :191 -- also report any / all over the valid bits.                 */
int knox_bits_and_flag(knox_ctx *ctx, knox_bits *dst, const knox_bits *src, int *any, int *all);
int knox_bits_or_flag(knox_ctx *ctx, knox_bits *dst, const knox_bits *src, int *any, int *all);
int knox_bits_popcount(knox_ctx *ctx, const knox_bits *b, int64_t *count);  /* PopCount :332 */
/* popcount of consecutive segments of seg_bits (a multiple of 64): the per-pack counts.   */
int knox_bits_popcount_segments(knox_ctx *ctx, const knox_bits *b, size_t seg_bits, int64_t *counts,
                                size_t n_counts);
/* Indexes :355 / Bitset.Indexes iterator.go:269: ascending uint32 row ids.  `capacity`
 * entries available in `rows` (host); *n_out receives the number written (= popcount).     */
int knox_bits_indexes(knox_ctx *ctx, const knox_bits *b, uint32_t *rows, size_t capacity, size_t *n_out);
/* The same into the context's own page-locked result buffer -- the default for large selections: no capacity to
 * guess (the set bits are counted first) and no copy into pageable memory, whose first touch costs several times the
 * PCIe transfer.  *rows_view points at *n_out row ids and stays valid until the next knox_bits_indexes_view /
 * knox_reduce_buckets_values call on this ctx (Go: unsafe.Slice(view, n), copy out if it must outlive that).        */
int knox_bits_indexes_view(knox_ctx *ctx, const knox_bits *b, const uint32_t **rows_view, size_t *n_out);
/* Enqueue-only forms of the flag / count producing calls: the result lands in the SAME 4-slot result ring
 * knox_scan_async uses (FIFO; a 5th uncollected result returns KNOX_E_BUSY), so a host walking a filter tree the way
 * MatchAnd / MatchOr do (match_core.go:45-190) can enqueue several leaf merges and read their flags afterwards without
 * a blocking read-back per call.  knox_bits_collect returns the OLDEST uncollected result, which must be a bitset
 * result (a scan is collected by knox_scan_collect; the wrong collector returns KNOX_E_INVALID and consumes nothing):
 * and_flag / or_flag fill any / all (count = -1), popcount fills count (any = all = -1).  All three nullable.        */
int knox_bits_and_flag_async(knox_ctx *ctx, knox_bits *dst, const knox_bits *src);
int knox_bits_or_flag_async(knox_ctx *ctx, knox_bits *dst, const knox_bits *src);
int knox_bits_popcount_async(knox_ctx *ctx, const knox_bits *b);
int knox_bits_collect(knox_ctx *ctx, int *any, int *all, int64_t *count);

/* Bool columns are bitsets; their vector matchers (internal/bitset/match.go:14-62 MatchEqual ..
 * MatchBetween) are copies / negations of the column.  `bits` must be zeroed by the caller; modes
 * that cannot match leave it untouched, exactly as in the reference.  a / b are 0 or 1.         */
int knox_bits_match_bool(knox_ctx *ctx, const knox_bits *column, int op, int a, int b, knox_bits *bits);

/* ---- predicate tree + fused scan ---------------------------------------------------- *
 * Replaces filter.Match(node, pkg, stats, bits) (operator/filter/match_core.go:14-215) for
 * every pack of the resident columns in ONE pass, plus the reducers that the reference
 * feeds row-at-a-time afterwards (query/result.go:134-150 -> reducer.go:138-349).
 *
 * The tree is flattened to postfix: leaves push, AND/OR pop `nchild` operands.
 * Leaf = filter.Filter{Type, Mode, Index, Value} (operator/filter/filter.go:22-30):
 * `col` indexes the cols[] array, a/b hold the value bits of the column type (little-endian,
 * zero-extended to 64 bit; b only for KNOX_RANGE).  Bytes leaves pass val/val_len -- EQ/NE
 * (StringPool.MatchEqual/NotEqual, pkg/stringx/match.go:20-84) and the bytes.Compare ordered modes
 * LT/LE/GT/GE/RANGE (:86-251, RANGE = [val, val2] inclusive);
 * KNOX_IN / KNOX_NOTIN leaves pass the set as val = array of column-type values, val_len =
 * its length in BYTES (numInSetMatcher / floatInSetMatcher, match_num.go:593-808); on a bytes
 * column the set is a packed list of [uint32 little-endian length][bytes] entries (any order,
 * duplicates allowed: bytesInSetMatcher / bytesNotInSetMatcher, match_bytes.go:300-540).       */
enum knox_node_kind { KNOX_NODE_LEAF = 0, KNOX_NODE_AND = 1, KNOX_NODE_OR = 2 };
typedef struct knox_pred {
    int32_t kind;   /* knox_node_kind */
    int32_t nchild; /* AND / OR: number of operands popped */
    int32_t col;    /* leaf: index into cols[] */
    int32_t op;     /* leaf: knox_op */
    uint64_t a, b;  /* leaf: value bits */
    const uint8_t *val; /* bytes leaf: value (RANGE: lower bound); IN / NOT IN: the set */
    uint64_t val_len;
    const uint8_t *val2; /* bytes leaf, KNOX_RANGE: upper bound (inclusive) */
    uint64_t val2_len;
} knox_pred;

typedef struct knox_agg {
    int32_t func; /* knox_reducer */
    int32_t col;  /* index into cols[] */
} knox_agg;

/* Partial aggregate state of one column over the selected rows: everything any reducer of
 * internal/reducer/reducer.go needs, in a form that merges exactly across blocks and GPUs. */
typedef struct knox_agg_state {
    int32_t dtype;
    uint32_t flags;     /* bit0: state valid; bit1: the m_* moments are valid */
    int64_t count;      /* selected rows */
    uint64_t isum;      /* wrapping sum of the raw value bits (integers: SumReducer mod 2^64) */
    double fsum_hi;     /* double-double sum of float64(v): hi + lo */
    double fsum_lo;
    uint64_t min_bits, max_bits;     /* value at min_row / max_row (raw bits, zero-extended) */
    uint64_t first_bits, last_bits;  /* value at first_row / last_row */
    int64_t min_row, max_row;        /* global row id of the adopted value, -1 = none       */
    int64_t first_row, last_row;     /* first / last selected global row id, -1 = none      */
    /* central moments for VarReducer / StdReducer (reducer.go:352-429), second pass:         */
    double m_shift;                  /* K = the first pass's mean estimate                   */
    double m_dev_hi, m_dev_lo;       /* double-double sum of (float64(v) - K)                */
    double m_sq_hi, m_sq_lo;         /* double-double sum of (float64(v) - K)^2              */
} knox_agg_state;

typedef struct knox_scan_out {
    int64_t count;          /* total selected rows (sum of per-pack popcounts) */
    int64_t rows_scanned;   /* rows actually read (pruned packs excluded)      */
} knox_scan_out;

/* One scan over rows [0, n_rows) of the given columns (all the same length).
 *   bits_out     nullable; receives the selection bitset of all rows (pack p's bitset is the
 *                slice starting at bit p*pack_rows -- pack_rows is a multiple of 8).
 *   pack_counts  nullable HOST array of n_packs int64: per-pack popcount (Bitset.ResetCount).
 *   pack_skip    nullable HOST array of n_packs bytes: 1 = pack pruned by the caller's
 *                statistics (zone map / bloom, pack/stats/match.go:112-195): treated as
 *                "no row matches" and not read.
 *   aggs/states  n_aggs reducers; states[i] receives the merged state for aggs[i].col
 *                (finalize with knox_agg_value).  row ids are offset by row_base.           */
int knox_scan(knox_ctx *ctx, const knox_pred *nodes, int n_nodes, knox_col *const *cols, int n_cols,
              const knox_agg *aggs, int n_aggs, uint64_t row_base, const uint8_t *pack_skip,
              knox_bits *bits_out, int64_t *pack_counts, knox_agg_state *states, knox_scan_out *out);
/* Enqueue-only variant: identical work; the results land in one of 4 ctx-owned pinned result
 * slots.  knox_scan_collect waits for and returns the OLDEST uncollected scan (FIFO), so a
 * host can keep up to 4 scans in flight -- enqueue scan i+1, then collect scan i -- and the
 * device never idles between packs of work.  A 5th uncollected async returns KNOX_E_BUSY.   */
int knox_scan_async(knox_ctx *ctx, const knox_pred *nodes, int n_nodes, knox_col *const *cols, int n_cols,
                    const knox_agg *aggs, int n_aggs, uint64_t row_base, const uint8_t *pack_skip,
                    knox_bits *bits_out, int want_pack_counts);
int knox_scan_collect(knox_ctx *ctx, int64_t *pack_counts, knox_agg_state *states, int n_aggs,
                      knox_scan_out *out);

/* Int128 / Int256 matchers ("next" row f4): cmp_i128_eq..bw / cmp_i256_eq..bw
 * (internal/cmp/int128.go:14-471, int256.go:14-270) over the reference's stride layout
 * num.Int128Stride{X0 []int64, X1 []uint64} / Int256Stride{X0..X3} (pkg/num/int128_stride.go:19-22,
 * int256_stride.go:22-27): limbs[l] is the resident 8-byte integer column of limb l, MOST
 * significant first; a / b hold n_limbs uint64, most significant first (num.Int128 = [2]uint64).
 * op = KNOX_EQ..KNOX_LE or KNOX_RANGE (a <= v && v <= b, signed).  bits = hits & mask (mask
 * nullable; bits is pre-zeroed by the caller per the Matcher contract).  *count (nullable) =
 * popcount of the result -- NOT the reference's return value when a mask is given: its unrolled
 * i128 loops add the popcount of the unmasked byte (int128.go:40-41), documented in DESIGN.md. */
int knox_match_wide(knox_ctx *ctx, knox_col *const *limbs, int n_limbs /* 2 | 4 */, int op, const uint64_t *a,
                    const uint64_t *b /* RANGE only */, const knox_bits *mask, knox_bits *bits, int64_t *count);
/* Int128 / Int256 IN / NOT IN (i128InSetMatcher / i128NotInSetMatcher and the i256 twins,
 * operator/filter/match_i128.go:318-430, match_i256.go): set = n_set values of n_limbs uint64, most
 * significant first, any order, duplicates allowed.  Matching rows are OR-ed into bits (bits.Set,
 * bits pre-zeroed by the caller); with a mask only mask-selected rows are tested.               */
int knox_match_wide_set(knox_ctx *ctx, knox_col *const *limbs, int n_limbs /* 2 | 4 */, int negate, const uint64_t *set,
                        size_t n_set, const knox_bits *mask, knox_bits *bits);

/* ---- bit-packed columns scanned as they are ------------------------------------------------- *
 * The reference compares on the packed form (internal/encode/bitpack/cmp_*.go behind
 * BitpackContainer.Match*, int_bitpack.go); so does the device: the [4] containers of consecutive
 * packs stay packed in HBM (log2/8 bytes per value instead of 8) and knox_packed_scan evaluates one
 * numeric leaf on them -- selection bitset, count, wrapping sum, per-pack counts -- without ever
 * materializing the values.  Any integer dtype; pack_rows a multiple of 4096; every
 * container but the last holds exactly pack_rows values.  Results are identical to
 * knox_col_upload_raw_containers + knox_scan on the same containers.                          */
typedef struct knox_packed knox_packed;
int knox_packed_upload(knox_ctx *ctx, int dtype, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs,
                       size_t pack_rows, knox_packed **out);
/* A packed column filled on the context's COPY stream (the cold-scan form: bit-packed data crosses PCIe at log2 / 64 of the
 * bytes of the decoded vectors, so a table that is not resident scans 64 / log2 times faster than knox_col_append_async can
 * feed it): knox_packed_create reserves room for max_packs packs and max_payload_bytes of packed words;
 * knox_packed_append_async enqueues the containers of the NEXT packs (same rules as knox_packed_upload: every pack but the
 * column's last holds pack_rows values; page-locked sources are read in place and must stay untouched until
 * knox_packed_wait, others are copied inside the call); knox_packed_scan orders itself behind pending appends on the
 * device; knox_packed_reset empties the column for the next run (call it before enqueueing the other buffer's scan).     */
int knox_packed_create(knox_ctx *ctx, int dtype, size_t pack_rows, size_t max_packs, size_t max_payload_bytes, knox_packed **out);
int knox_packed_append_async(knox_ctx *ctx, knox_packed *p, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs);
int knox_packed_reset(knox_ctx *ctx, knox_packed *p);
int knox_packed_wait(knox_ctx *ctx, knox_packed *p);
int knox_packed_free(knox_packed *p);
size_t knox_packed_rows(const knox_packed *p);
size_t knox_packed_bytes(const knox_packed *p); /* packed payload resident in HBM */
/* op / a / b as in a knox_pred leaf.  bits_out nullable; pack_counts (host, n_packs) needs bits_out;
 * sum_state (nullable unless want_sum) receives count + isum (knox_agg_value: Sum, Count).      */
int knox_packed_scan(knox_ctx *ctx, const knox_packed *p, int op, uint64_t a, uint64_t b, int want_sum,
                     knox_bits *bits_out, int64_t *pack_counts, int64_t *count, knox_agg_state *sum_state);

/* ---- integer columns kept ENCODED in HBM and scanned as they are (round 4) --------------------- *
 * The reference matches on containers without materializing them: DictionaryContainer.Match* looks the value up
 * in the sorted dictionary and compares the uint16 CODES (internal/encode/int_dict.go:181-380), RunEndContainer.Match*
 * compares the run VALUES and turns matching runs into row ranges (int_runend.go:224-330), constant and delta
 * containers answer from their header (int_const.go, int_delta.go).  So does the device, PACK BY PACK -- the
 * reference's encoder picks the scheme per block, a run of packs is usually mixed:
 *   dictionary [5]   2 B of codes per row + the dictionary (predicate evaluated on the dictionary once per pack)
 *   run-end [3]      (W + 4) B per run (predicate on the run values, matching runs expanded into the bitset)
 *   constant [1], delta [2]   nothing resident: row i = T(i * delta + base)
 *   every other integer container (raw, bit-packed, simple8, nested dictionary / run-end): decoded on the device
 *                    at upload, W B per row, compared there
 * knox_enc_scan evaluates one numeric leaf on such a column: selection bitset, count, wrapping sum, per-pack counts.
 * The children of dictionary / run-end containers are decoded on the device; what crosses PCIe is the containers'
 * own bytes.  Any integer dtype (floats: KNOX_E_UNSUPPORTED, use knox_col_upload_raw_containers); pack_rows a
 * multiple of 8192; every pack but the last holds exactly pack_rows rows.  Results are identical to
 * knox_col_upload_raw_containers + knox_scan on the same containers (tests/test_gpu_encoded.py).
 * op = KNOX_EQ .. KNOX_LE, KNOX_RANGE; a / b as in a knox_pred leaf.                                              */
typedef struct knox_enc knox_enc;
int knox_enc_upload(knox_ctx *ctx, int dtype, const uint8_t *const *bufs, const size_t *lens, size_t n_bufs,
                    size_t pack_rows, knox_enc **out);
int knox_enc_free(knox_enc *e);
size_t knox_enc_rows(const knox_enc *e);
size_t knox_enc_bytes(const knox_enc *e);           /* bytes resident in HBM (codes + dictionaries, run values + ends, decoded packs) */
size_t knox_enc_container_bytes(const knox_enc *e); /* bytes of the containers as they came (what crossed PCIe)        */
/* how many packs are scanned in which form (each pointer nullable) */
int knox_enc_kinds(const knox_enc *e, size_t *n_dictionary, size_t *n_runend, size_t *n_const_delta, size_t *n_raw);
/* bits_out nullable; pack_counts (host, n_packs) needs bits_out; sum_state (nullable unless want_sum) receives
 * count + isum (knox_agg_value: Sum, Count).                                                                  */
int knox_enc_scan(knox_ctx *ctx, const knox_enc *e, int op, uint64_t a, uint64_t b, int want_sum,
                  knox_bits *bits_out, int64_t *pack_counts, int64_t *count, knox_agg_state *sum_state);

/* Matcher.MatchRangeVectors (operator/filter/match.go:66-71; numeric forms match_num.go:357-371,
 * 396-403, 429-434, 460-465, 491-496, 522-527, 573-588): vectorized zone-map test over the
 * per-pack min / max statistics columns (one row per pack) -- which packs MAY contain a match.
 *   EQ: min <= v && max >= v (AND mask)      RANGE: min <= to && max >= from (AND mask)
 *   NE: mask, or all ones                    GT/GE: max > / >= v      LT/LE: min < / <= v
 * (the single-column forms ignore the mask exactly like the reference's MatchVector does).
 * The result feeds knox_scan's pack_skip ("next" row f3: pruning without a host round trip). */
int knox_match_range_vectors(knox_ctx *ctx, const knox_col *mins, const knox_col *maxs, int op, uint64_t a,
                             uint64_t b, const knox_bits *mask /* nullable */, knox_bits *bits);

/* The same for bytes statistics columns (match_bytes.go:97-117, 138-145, 165-170, 190-195, 215-220,
 * 240-245, 282-299): bytes.Compare order; RANGE tests min <= val2 && max >= val.           */
int knox_match_range_vectors_bytes(knox_ctx *ctx, const knox_col *mins, const knox_col *maxs, int op,
                                   const uint8_t *val, uint64_t val_len, const uint8_t *val2, uint64_t val2_len,
                                   const knox_bits *mask /* nullable */, knox_bits *bits);
/* numInSetMatcher.MatchRangeVectors (match_num.go:693-727), materialized statistics vectors: bits[p] = the set holds a
 * member in [uint64(min[p]), uint64(max[p])] (Go's conversion: signed values sign-extend; the pair is swapped when reversed),
 * and-ed with the mask.  set = n_set values of the (integer) column type, any order, duplicates allowed.  negate != 0 =
 * numNotInSetMatcher: undecided, bits = mask or all ones (:810-817).  The reference's branch for COMPRESSED statistics
 * blocks answers with the coarser range test [set min, set max]: knox_match_range_vectors(KNOX_RANGE).               */
int knox_match_range_vectors_set(knox_ctx *ctx, const knox_col *mins, const knox_col *maxs, int negate, const void *set,
                                 size_t n_set, const knox_bits *mask /* nullable */, knox_bits *bits);
/* The Int128 / Int256 forms (match_i128.go:78-375, match_i256.go:64-229) over the limb columns of the min / max statistics
 * (most significant limb first, as knox_match_wide): EQ (min <= v) & (max >= v), RANGE (min <= to) & (max >= from) -- also
 * what the IN matchers ask with [set min, set max] (:371-376) --, LT / LE on min, GT / GE on max, all and-ed with the mask
 * (the wide compare kernels take one); NE / NOT IN: the mask, or all ones.                                          */
int knox_match_range_vectors_wide(knox_ctx *ctx, knox_col *const *mins, knox_col *const *maxs, int n_limbs /* 2 | 4 */, int op,
                                  const uint64_t *a, const uint64_t *b /* RANGE only */, const knox_bits *mask /* nullable */,
                                  knox_bits *bits);

/* bytesRegexpMatcher (operator/filter/match_bytes.go:543-609): MatchVector = regexp.Match(row) for every
 * (masked) row of a bytes column.  The pattern is compiled on the host into a DFA over rune classes and
 * the rows are walked one per lane, decoding runes exactly as Go does (invalid byte = U+FFFD, width 1).
 * bits: overwritten when mask is NULL; with a mask only masked rows are evaluated and their hits OR-ed in
 * (the reference's bits.Set(i) loop).  Syntax: Go's regexp/syntax (RE2) -- literals, escapes, classes,
 * Perl and POSIX classes, \pN \p{Name} \P{Name} (general categories and their aliases, scripts, Any / Assigned /
 * ASCII), . * + ? {n,m} (lazy forms too), groups and named groups, | , ^ $ \A \z \b \B, flags i m s U ((?i) =
 * Unicode simple case folding, unicode.SimpleFold).  KNOX_E_UNSUPPORTED = outside that subset (a construct
 * regexp.Compile rejects, a DFA above 4096 states or 255 rune classes): the caller keeps regexp.Match for that
 * condition.  The same program serves a KNOX_REGEXP leaf of knox_scan (val/val_len = pattern).          */
int knox_match_regexp(knox_ctx *ctx, const knox_col *col, const uint8_t *pattern, uint64_t pattern_len,
                      const knox_bits *mask /* nullable */, knox_bits *bits, int64_t *count /* nullable */);
/* MatchValue (match_bytes.go:578-580) for one value, host only: 1 match, 0 no match, <0 error.  */
int knox_regexp_match_value(const uint8_t *pattern, uint64_t pattern_len, const uint8_t *value, uint64_t value_len);
/* The Unicode version of the tables behind \p{..} and (?i) ("17.0.0").  A host whose own tables differ (Go:
 * unicode.Version) keeps patterns that use \p{..} or fold non-ASCII runes on its own regexp: code points assigned or
 * re-categorised between the two versions would match differently.                                         */
const char *knox_regexp_unicode_version(void);

/* Masked reduction of one column by an existing bitset (reduce_* kernels of the plan).      */
int knox_reduce(knox_ctx *ctx, const knox_col *col, const knox_bits *mask /*nullable = all rows*/,
                uint64_t row_base, knox_agg_state *state);
/* Var / Std second pass: adds the central moments to a state knox_reduce (or knox_scan) produced
 * over the SAME column and selection (the pass is centred on that state's mean).  knox_scan runs
 * it by itself for KNOX_R_VAR / KNOX_R_STD aggregates.                                       */
int knox_reduce_moments(knox_ctx *ctx, const knox_col *col, const knox_bits *mask /*nullable = all rows*/,
                        knox_agg_state *state /* in/out */);
/* Time-window aggregation ("next" row f1): what pkg/series feeds row-at-a-time into
 * NativeBucket.Push (internal/reducer/bucket_native.go:104-159): one reducer per window
 * [t_from + k*step, t_from + (k+1)*step), k < n_buckets, over the rows `mask` selects.
 * time_col is an int64 column (the reference's time field is int64 internally).  When it is
 * NON-DECREASING over the scanned rows every window is one contiguous row range and is reduced in
 * place.  Out-of-order timestamps (the reference finds / inserts the window's reducer by time,
 * bucket_native.go:123-141) take a second path: the selected rows are sorted by window index
 * (stable, so each window keeps its stream order), the values gathered, and the same reduction runs
 * on the gathered rows; row ids in the states are the caller's.  Either way window k's state is
 * what the reference's reducer for that window saw.
 * states[k] is the window's mergeable state (count == 0: the reference created no reducer).
 * flags bit0: the caller guarantees the order, skip the verification pass.
 * flags bit1: also run the var/std second pass per window (states carry the m_* moments).
 * flags bit4 (16): run the verification pass even when this column's order is already known.  (What the pass finds is
 * kept with the column -- columns the library filled itself, until rows are appended or the column is reset; never for
 * knox_col_wrap -- so a time column that is queried again and again is checked once: the pass reads 8 B per row, half of
 * what an ordered window query reads.)                                                          */
int knox_reduce_buckets(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col,
                        const knox_bits *mask /* nullable = all rows */, int64_t t_from, int64_t step,
                        size_t n_buckets, uint64_t row_base, int flags, knox_agg_state *states);

/* The same states as a VIEW into the context's page-locked result buffer (valid until the next *_view / *_values call on
 * this ctx): 144 B per window at DMA rate -- for hosts that merge windows across GPUs with knox_agg_combine.         */
int knox_reduce_buckets_view(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col,
                             const knox_bits *mask /* nullable = all rows */, int64_t t_from, int64_t step,
                             size_t n_buckets, uint64_t row_base, int flags, const knox_agg_state **states_view);
/* Values only: Reducer.Value() of reducer `func` (KNOX_R_SUM .. KNOX_R_COUNT) for every window, finalised ON THE DEVICE
 * by the same code knox_agg_value runs on the host (bit-identical by construction), so 8 + 1 bytes per window cross
 * PCIe instead of a 144-byte state: values_view = n_buckets values of the column's type, seen_view[k] = Value()'s ok
 * (0: the reference created no reducer for the window).  Both point into the context's page-locked result buffer and
 * stay valid until the next knox_reduce_buckets_values / knox_bits_indexes_view call on this ctx.  flags bit0 as above;
 * the var / std pass runs by itself for KNOX_R_VAR / KNOX_R_STD.  Feed the views to knox_fill_windows for Emit's gaps.
 * Out-of-order time: every reducer has an order-free form (first / last as the lowest / highest row, float sums and the sums behind
 * var / std as exact fixed-point numbers) and is folded where the rows lie -- what NativeBucket.Push does -- as long as the windows'
 * table fits a workgroup's LDS (a few thousand windows); more windows go through the stable sort.  The results are the same bytes. */
int knox_reduce_buckets_values(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col,
                               const knox_bits *mask /* nullable = all rows */, int64_t t_from, int64_t step,
                               size_t n_buckets, int func, int flags, const void **values_view, const uint8_t **seen_view);

/* Which way the last knox_reduce_buckets* call on this ctx took (diagnostics, tests): the time column was non-decreasing and the
 * windows were reduced in place; out-of-order time through the stable sort (after a compaction of the selected rows); out-of-order
 * time without a sort -- Value() of an order-free reducer, whole states, exact float sums / means.                                */
enum knox_windows_path {
    KNOX_WINDOWS_ORDERED = 1, KNOX_WINDOWS_SORTED = 2, KNOX_WINDOWS_SORTED_COMPACTED = 3,
    KNOX_WINDOWS_DIRECT_VALUES = 4, KNOX_WINDOWS_DIRECT_STATES = 5, KNOX_WINDOWS_DIRECT_FLOAT_SUMS = 6
};
int knox_windows_last_path(const knox_ctx *ctx);

/* The literal form of the same: window k's reducer is the reference's own, applied row by row
 * (one lane per window), so Value() is bit-exact for EVERY reducer -- Welford mean / var / std
 * and the join reducers KNOX_R_*_JOIN (reducer.go:432-731; join = the row's raw timestamp equals
 * the previous selected row's, pkg/series/series.go:200-203) included.  values_out holds
 * n_buckets values of the column's type, seen_out[k] = Value()'s ok (0: no reducer / no group).
 * Needs a non-decreasing time column (KNOX_E_UNSUPPORTED otherwise); flags bit0 as above.
 * The recurrence of a window is serial by definition; what varies is who fetches its rows: one lane per
 * window when there are enough windows for every lane of the device, a whole wave per window below that
 * (64 rows per coalesced fetch, handed to the SAME recurrence in row order -- identical results).  flags
 * bit2 (4) forces the one-lane form, bit3 (8) the wave form (tests and A/B).                      */
int knox_reduce_buckets_sequential(knox_ctx *ctx, const knox_col *time_col, const knox_col *value_col,
                                   const knox_bits *mask /* nullable = all rows */, int64_t t_from, int64_t step,
                                   size_t n_buckets, int func, int flags, void *values_out, uint8_t *seen_out);

/* Literal sequential reducer (one lane, row order) -- reproduces reducer.go bit for bit,
 * including Welford mean/var; slow, used for parity tests on small inputs.                  */
int knox_reduce_sequential(knox_ctx *ctx, const knox_col *col, const knox_bits *mask, int func,
                           void *value_out /* one T */, int *seen, int64_t *row);

/* Reducer.Value() from a state: writes one value of the column type to value_out, *seen as
 * Value()'s second result, *row = the row whose timestamp Time() would report.
 * sum/mean of float columns are the correctly rounded exact sum / exact mean.   */
int knox_agg_value(const knox_agg_state *state, int func, void *value_out, int *seen, int64_t *row);
/* Merge n states of consecutive row ranges (rank order = row order) -- pure host code.      */
int knox_agg_combine(const knox_agg_state *states, int n, knox_agg_state *out);

/* NativeBucket.Emit's gap filling (bucket_native.go:161-218, Fill fill.go:59-85) over the dense per-window
 * values of knox_reduce_buckets / _sequential -- pure host code, everything but the string formatting:
 * window k (time t_from + k*step) with seen[k] emits its value; a window without a reducer is filled per
 * `mode` from the previous and the next reducer (the reference's start / end behaviour kept: `last`
 * starts as the first reducer, `next` past the last reducer is `last`).  Linear fill interpolates over
 * time.Time.Unix() seconds: units_per_second converts the raw time unit (1e9 for nanoseconds).
 * out_flags[k]: 0 nothing emitted, 1 out_values[k] emitted, 2 null emitted.  At most `limit` entries
 * are emitted; *n_out = that count.                                                           */
enum knox_fill_mode { KNOX_FILL_NONE = 0, KNOX_FILL_NULL = 1, KNOX_FILL_LAST = 2, KNOX_FILL_LINEAR = 3, KNOX_FILL_ZERO = 4 };
int knox_fill_windows(int dtype, int mode, const void *values, const uint8_t *seen, size_t n_buckets,
                      int64_t t_from, int64_t step, int64_t units_per_second, size_t limit, void *out_values,
                      uint8_t *out_flags, size_t *n_out);

/* ---- bloom pre-filter: internal/filter/bloom/bloom.go:136-150,182 -------------------- *
 * A knox_bloom holds the per-pack filters of one column resident on the device: n_packs
 * serialized filters [k][m/8 bytes] (bloom.go:48-60,83-101) concatenated, offsets[n_packs+1]
 * giving each filter's byte range (offsets[p] == offsets[p+1]: pack has no filter -> maybe).  */
typedef struct knox_bloom knox_bloom;
int knox_bloom_upload(knox_ctx *ctx, const uint8_t *filters, const uint64_t *offsets, size_t n_packs,
                      knox_bloom **out);
/* Build the per-pack filters on the device: numeric columns of any width hash the fixed-width
 * xxh3 of the raw value bits (hash.HashT, internal/hash/hash.go:67-92, xxh3.go:22-58), bytes columns
 * the XXH3-64 of every row's bytes (hash.Hash; pack/stats/filter.go:352-354); k = 4,
 * m = pow2(pack_rows * factor * 8) bits (pack/stats/filter.go:296-360; factor =
 * FilterType.Factor(), bytes per value).  knox_bloom_build_u64 is the 8-byte form (config 4). */
int knox_bloom_build(knox_ctx *ctx, const knox_col *col, int factor, knox_bloom **out);
int knox_bloom_build_u64(knox_ctx *ctx, const knox_col *col, int factor, knox_bloom **out);
size_t knox_bloom_packs(const knox_bloom *b);
/* How a built filter set came about (diagnostics): batches of packs the build ran in, and how many of them took the exact
 * count / scan / bin / fill passes -- because a fixed entry region of the one-read build overflowed (heavy hitters in a
 * column whose filters are sized for many distinct values) or the first batch said the column is of that kind.  The bytes
 * are the same either way.                                                                                           */
int knox_bloom_build_info(const knox_bloom *b, int *batches, int *exact_batches);
size_t knox_bloom_bytes(const knox_bloom *b);
int knox_bloom_download(knox_ctx *ctx, const knox_bloom *b, uint8_t *filters, size_t filters_cap,
                        uint64_t *offsets /* n_packs+1 */);
/* Filter.Contains(h) for every pack: maybe_out[p] = 1 if pack p may contain hash h.          */
int knox_bloom_probe(knox_ctx *ctx, const knox_bloom *b, uint64_t h, uint8_t *maybe_out);
int knox_bloom_free(knox_bloom *b);
/* hash.HashT for fixed-width values (internal/hash/hash.go:67-92, xxh3.go:22-58): the hash a
 * numeric Equal matcher probes with (match_num.go:339-346).  Float types hash their bytes.   */
uint64_t knox_hash_value(int dtype, const void *val);
/* hash.Hash (internal/hash/hash.go:26 = zeebo/xxh3.Hash): XXH3-64, seed 0, of a byte string -- the
 * probe value for filters built over bytes columns (bytesMatcher.WithValue, match_bytes.go:56-59). */
uint64_t knox_hash_bytes(const uint8_t *val, size_t len);

/* ---- multi-GPU: final aggregate only ------------------------------------------------- *
 * One process per GPU.  Rank 0 creates an id, every rank receives it out of band (the host's
 * own rendezvous) and joins; knox_comm_allgather_states gathers each rank's state over
 * RCCL (xGMI) and merges them in rank order with knox_agg_combine on every rank.            */
#define KNOX_COMM_ID_BYTES 128
int knox_comm_unique_id(uint8_t id[KNOX_COMM_ID_BYTES]);
int knox_comm_init(knox_ctx *ctx, int n_ranks, int rank, const uint8_t id[KNOX_COMM_ID_BYTES]);
int knox_comm_allgather_states(knox_ctx *ctx, const knox_agg_state *mine, int n_states,
                               knox_agg_state *merged /* n_states */, int64_t *count_inout);
/* The order-free integer aggregates in ONE ncclAllReduce(ncclInt64, ncclSum) over xGMI: the wrapping integer sum
 * (SumReducer, reducer.go:168-178; state.isum) and the selected-row count.  The sum crosses as two 32-bit halves, so
 * the result is exactly the sum mod 2^64 for any reduction order -- bit-identical to the all-gather + combine above.
 * min / max / first / last and float sums need rank order and take knox_comm_allgather_states.                   */
int knox_comm_allreduce_sum_count(knox_ctx *ctx, uint64_t *isum_inout, int64_t *count_inout);
/* The same exchange, enqueue only: _start puts {sum mod 2^64, count} on the context's collective stream (page-locked
 * slot -> ncclAllReduce -> page-locked slot) and returns; _finish blocks for the OLDEST exchange in flight (FIFO, up to
 * 4; a 5th _start returns KNOX_E_BUSY).  The exchange waits for nothing on the scan stream, so a host keeps step i's
 * collective in flight behind step i + 1's scan.  The blocking call above is _start + _finish on an empty FIFO.       */
int knox_comm_allreduce_sum_count_start(knox_ctx *ctx, uint64_t isum, int64_t count);
int knox_comm_allreduce_sum_count_finish(knox_ctx *ctx, uint64_t *isum_out, int64_t *count_out);
/* The communicator's own size (ncclCommCount) and this context's rank.                                                 */
int knox_comm_ranks(knox_ctx *ctx, int *n_ranks, int *rank);
int knox_comm_destroy(knox_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* KNOXGPU_H */
