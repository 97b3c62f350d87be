// knox_kernels.h -- internal interface between the host side of the C ABI (knox_abi.cpp and its sibling .cpp files) and
// the gfx950 kernel translation units (knox_scan / knox_agg / knox_window / knox_bitset / knox_bytes / knox_bloom /
// knox_decode / knox_sort .hip).  Not installed; the public surface is include/knoxgpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace knox {

constexpr int kBlock = 256;          // threads per workgroup = 4 wave64
constexpr int kWavesPerBlock = 4;
constexpr int kUnroll = 4;           // 16-byte loads in flight per lane per column
constexpr int kMaxIters = 16;        // wave iterations whose bitset words are staged before one burst store
constexpr int kMaxLeaves = 4;        // leaves evaluated by one fused pass (truth table 2^4 bits)
constexpr uint64_t kNoRow = ~0ull;

// aggregate work a pass performs (template mask)
enum : unsigned { A_ISUM = 1, A_FSUM = 2, A_MINMAX = 4, A_FIRSTLAST = 8, A_ALL = 15 };
// value flavour of the aggregated column
enum : unsigned { FL_SIGNED = 0, FL_UNSIGNED = 1, FL_FLOAT = 2 };

// ONE table for the two sides that must agree on a scan instantiation's launch shape: the workgroups per CU run_fused
// (knox_abi.cpp) launches and the LDS words per wave the kernel (knox_scan.hip) may park deferred bitset bursts in.  The
// deferral's chip-wide phases only line up when the whole grid is resident, i.e. per_cu workgroups really fit a CU:
// per_cu * kWavesPerBlock * park_words * 4 B must stay inside the CU's LDS (static_assert below, gfx950: 160 KiB), and
// knox_scan_check_occupancy (knox_scan.hip) asks the runtime once per instantiation that it agrees.
//   heavy  = aggregates beyond the wrapping integer sum (float sums, min / max, first / last)
//   flt    = the aggregated column is float,  fsum = a float sum is among the aggregates
// Measured choices (tools/sweep_narrow.py, profiles/r01_sweep_blocks_per_cu.txt, profiles/r03_narrow.txt):
// streaming passes 2 per CU (1-byte columns 3); aggregate-heavy single leaves 2 (2-byte columns 3, float32 sums 8); two
// aggregate-heavy leaves 2 (pipelined, 177 VGPRs); every other tree 3.
struct ScanShape {
    int per_cu;            // workgroups per CU the launch asks for
    uint32_t park_words;   // bitset words per WAVE parked in LDS before one flush (0: bursts leave tile by tile)
};
constexpr uint32_t kLdsPerCu = 160u * 1024u;
constexpr ScanShape scan_shape(int W, int NL, bool heavy, bool flt, bool fsum) {
    int per_cu = 0;
    // (two plain leaves on 2-byte columns -- pipelined, 210 VGPRs -- and three / four aggregate-heavy leaves on 1-byte columns
    //  -- 199-218 VGPRs -- hold two workgroups per CU: round 4, from the compiler's own occupancy figures)
    if (NL >= 2) per_cu = ((NL == 2 && heavy) || (NL == 2 && W == 2) || (NL >= 3 && W == 1 && heavy)) ? 2 : 3;
    else if (!heavy) per_cu = W == 1 ? 3 : 2;
    else if (W == 4 && flt && fsum) per_cu = 8;
    else per_cu = W == 2 ? 3 : 2;
    // float32 aggregates keep their tile-by-tile bursts (the sums want 8 workgroups per CU)
    const bool f32_heavy = NL == 1 && heavy && W == 4 && flt;
    const uint32_t park = f32_heavy ? 0u : per_cu == 2 ? 3840u : per_cu == 3 ? 3072u : 0u;   // 15 KiB / 12 KiB per wave
    return ScanShape{per_cu, park};
}
constexpr bool scan_shape_fits(int W, int NL, bool heavy, bool flt, bool fsum) {
    return (uint32_t)scan_shape(W, NL, heavy, flt, fsum).per_cu * kWavesPerBlock * scan_shape(W, NL, heavy, flt, fsum).park_words * 4u <= kLdsPerCu;
}
// instantiations whose resident workgroups per CU (asked of the runtime at first launch) are below scan_shape's (knox_agg.hip)
extern std::atomic<int> g_scan_occupancy_mismatches;
static_assert(scan_shape_fits(8, 1, false, false, false) && scan_shape_fits(1, 1, false, false, false) && scan_shape_fits(2, 1, true, false, false) &&
                  scan_shape_fits(8, 2, true, true, true) && scan_shape_fits(8, 2, false, false, false) && scan_shape_fits(4, 4, true, true, true),
              "parked bitset words of per_cu workgroups exceed the CU's LDS");

// One leaf in "key space": hit <=> U(key(x) - lo) <= diff, key = float order key or raw bits.
struct LeafDev {
    const void *col;
    uint64_t lo;
    uint64_t diff;
    uint32_t is_float;
    uint32_t _pad;
};

// Mergeable partial aggregate (one per workgroup, merged in block order by finalize).
struct AggPartial {
    uint64_t count;
    uint64_t isum;
    double hi, lo;
    uint64_t minkey, minrow;
    uint64_t maxkey, maxrow;
    uint64_t firstrow, lastrow;
    uint64_t xlo;   // integer columns: exact 128-bit sum (xhi:xlo)
    int64_t xhi;
};

struct ScanArgs {
    LeafDev leaf[kMaxLeaves];
    uint64_t tt;            // truth table over the leaf hit bits (bit index = sum hit_l << l)
    const void *agg_col;    // aggregated column (may alias leaf[0].col)
    uint32_t agg_flavor;
    uint32_t pack_shift;    // log2(pack_rows), or 255: no per-pack skipping (tail kernel only)
    const uint8_t *pack_skip;
    const uint32_t *live_packs;  // streaming kernel: compact list of packs to scan (nullable = all)
    uint32_t tpp_shift;          // log2(tiles per pack) when live_packs is set
    uint64_t n_tiles;       // full tiles handled by the streaming kernel
    uint64_t n_rows;        // all rows
    uint64_t tail_first_row; // tail kernel covers [tail_first_row, n_rows)
    uint64_t row_base;      // global id of local row 0
    uint32_t *bits;         // output bitset (nullable)
    const uint32_t *mask;   // reduce pass: input selection bitset (nullable = all rows)
    uint32_t debug;         // read by -DKNOX_DIAG builds only (make diag); the product kernels ignore it and the ABI leaves it 0
    uint32_t flags;         // bit0: bitset bursts leave tile by tile (no deferral: the arena's placement probes, KNOX_DEFER=0); bit1: deferred bursts are not aligned in time (KNOX_DEFER=2)
    uint32_t iters;         // wave iterations per tile (1..kMaxIters): tile = iters * tile_rows(width)
    uint32_t *tile_counts;  // nullable: [(workgroup * kWavesPerBlock + wave) * tc_rounds + round] = popcount of that wave tile's burst
    uint32_t tc_rounds;     // rounds per workgroup, padded to a multiple of 32 (one 128-byte line of counts)
    AggPartial *partials;   // [grid] (+1 slot for the tail at index grid)
};

// mirror of knox_agg_state (include/knoxgpu.h) written by the finalize kernel
struct AggStateDev {
    int32_t dtype;
    uint32_t flags;
    int64_t count;
    uint64_t isum;
    double fsum_hi, fsum_lo;
    uint64_t min_bits, max_bits, first_bits, last_bits;
    int64_t min_row, max_row, first_row, last_row;
    // second-pass central moments (var/std), valid when flags bit1 is set
    double m_shift;             // K: the first pass's mean estimate
    double m_dev_hi, m_dev_lo;  // sum of (v - K)
    double m_sq_hi, m_sq_lo;    // sum of (v - K)^2
};

int tile_rows(int width);  // rows per workgroup tile of the streaming kernels

// Streaming scan: evaluates the leaves' truth table for n_tiles full tiles, writes bitset
// words, accumulates count + aggregates into partials[0..grid).  width = bytes per value.
hipError_t launch_scan(hipStream_t s, int width, int n_leaves, unsigned agg, bool agg_separate,
                       int grid, const ScanArgs &a);
// placement probe: scan_kernel<8,1,ISUM> under a separate symbol
hipError_t launch_probe_scan(hipStream_t s, int grid, const ScanArgs &a);
// Tail / small-n path: rows [n_tiles*tile_rows, n_rows), byte-granular stores, one workgroup,
// partial written to partials[tail_slot].
hipError_t launch_scan_tail(hipStream_t s, int width, int n_leaves, unsigned agg, int tail_slot,
                            const ScanArgs &a);
// Masked reduce: aggregates agg_col over rows selected by a.mask (or all rows).
hipError_t launch_reduce(hipStream_t s, int width, unsigned agg, int grid, const ScanArgs &a);
// Merge partials[0..n_partials) in order, fetch the adopted values, write the state and the
// total count.
// finalize + per-pack counts from the per-tile counts as ONE launch (a scan's whole follow-up work)
hipError_t launch_finalize_packs(hipStream_t s, int width, const AggPartial *partials, int n_partials, const void *agg_col, uint64_t row_base,
                                 int dtype, AggStateDev *state_out, int64_t *count_out, const uint32_t *tile_counts, uint32_t grid,
                                 uint32_t rounds, uint32_t tiles_per_pack, uint64_t n_packs, int64_t *pack_counts);
hipError_t launch_finalize(hipStream_t s, int width, const AggPartial *partials, int n_partials,
                           const void *agg_col, uint64_t row_base, int dtype, AggStateDev *state_out,
                           int64_t *count_out);

// Second pass for VarReducer / StdReducer (internal/reducer/reducer.go:352-429): double-double sums
// of (v - K) and (v - K)^2 over the selected rows, K = state_in->fsum_hi / count (the first pass's
// mean) or `shift` when state_in == nullptr.  partials: 4 doubles per workgroup.  out5 receives
// {K, dev_hi, dev_lo, sq_hi, sq_lo}; *flags_out |= 2 when given.
hipError_t launch_moments(hipStream_t s, int width, uint32_t flavor, int grid, const void *col, const uint32_t *mask, uint64_t n_rows,
                          const AggStateDev *state_in, double shift, double *partials, double *out5, uint32_t *flags_out);

// ---- time-window (bucket) aggregation over a non-decreasing int64 time column ----------------
struct BucketItem {        // one work item: a chunk of one window's contiguous row range
    uint64_t begin, end;   // rows [begin, end)
    uint32_t bucket;
    uint32_t _pad;
};
// flag[0] |= 1 if time[i] > time[i+1] for some i
hipError_t launch_sorted_check(hipStream_t s, const int64_t *time, uint64_t n_rows, uint32_t *flag);
// row_start[k] = first row with time >= t_from + k*step, k = 0..n_buckets
hipError_t launch_bucket_bounds(hipStream_t s, const int64_t *time, uint64_t n_rows, int64_t t_from, int64_t step,
                                uint64_t n_buckets, uint64_t *row_start);
// one partial per work item (a.partials[item]); a.agg_col / a.agg_flavor / a.mask / a.row_base as in launch_reduce
hipError_t launch_bucket_reduce(hipStream_t s, int width, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items);
// state_out[k] = ordered merge of partials[item_start[k] .. item_start[k+1])
// the same with one wave per work item: many small windows
hipError_t launch_bucket_reduce_waves(hipStream_t s, int width, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items);
hipError_t launch_bucket_reduce_groups(hipStream_t s, int width, int grid, const ScanArgs &a, const BucketItem *items, uint32_t n_items, bool sums_only);
hipError_t launch_bucket_finalize(hipStream_t s, int width, const AggPartial *partials, const uint32_t *item_start, uint32_t n_buckets,
                                  const void *agg_col, uint64_t row_base, int dtype, AggStateDev *state_out, int func = 0, void *values = nullptr, uint8_t *seen = nullptr);

// var/std per window (second pass over the same work items); partials: 4 doubles per work item
// group_rows != 0: the 16-lanes-per-window form (grid = workgroups of 16 windows; group_rows = the column's rows, the bitset's bound)
hipError_t launch_bucket_moments(hipStream_t s, int width, uint32_t flavor, int grid, const void *col, const uint32_t *mask, uint64_t group_rows,
                                 const BucketItem *items, uint32_t n_items, const uint32_t *item_start, uint32_t n_buckets,
                                 AggStateDev *states, double *partials);

// Int128 / Int256 compare over limb columns (most significant first; a / b already sign-flipped in limb 0)
struct WideArgs {
    const uint64_t *limb[4];
    uint64_t a[4], b[4];
    uint32_t op;             // FilterMode: 1 eq 2 ne 3 gt 4 ge 5 lt 6 le 9 range
    uint32_t _pad;
    const uint32_t *mask;    // nullable
    uint32_t *bits;
    uint64_t n_tiles;        // full tiles of tile_rows(8) rows
    uint64_t n_rows;
    unsigned long long *count;  // pre-zeroed, nullable
};
hipError_t launch_wide_cmp(hipStream_t s, int n_limbs, int grid, const WideArgs &w);
// wide IN / NOT IN: keys = sorted unique set, n_limbs words per value, top limb sign-flipped; matches OR-ed into bits
hipError_t launch_wide_inset(hipStream_t s, int n_limbs, bool negate, const uint64_t *const *limbs, uint64_t n_rows, const uint64_t *keys,
                             uint32_t n_keys, uint8_t *bits, const uint8_t *mask);

// bitset combinators on 64-bit words (buffers padded to a multiple of 8 bytes)
enum BitOp { BIT_AND = 0, BIT_ANDNOT = 1, BIT_OR = 2, BIT_XOR = 3, BIT_NEG = 4, BIT_COPY = 5, BIT_ONE = 6 };
hipError_t launch_bitop(hipStream_t s, int op, uint64_t *dst, const uint64_t *src, uint64_t size_bits,
                        uint32_t *flags_any_all /* nullable: [0]=any (or), [1]=not-all (or) */);
hipError_t launch_popcount(hipStream_t s, const uint64_t *bits, uint64_t size_bits,
                           unsigned long long *count_out /* pre-zeroed */);
// counts[p] = sum of the wave-tile counts of pack p's tiles_per_pack tiles (scan launched with `grid` workgroups)
hipError_t launch_pack_counts(hipStream_t s, const uint32_t *tile_counts, uint32_t grid, uint32_t rounds, uint32_t tiles_per_pack,
                              uint64_t n_packs, int64_t *counts);
hipError_t launch_popcount_segments(hipStream_t s, const uint64_t *bits, uint64_t size_bits, uint64_t seg_bits,
                                    int64_t *counts, uint64_t n_seg, const uint8_t *skip = nullptr, unsigned long long *total = nullptr);
hipError_t launch_zero_bytes(hipStream_t s, void *p, uint64_t bytes);
// bitset -> ascending uint32 row ids; scratch holds ceil(size/16384)+1 uint32 block offsets
hipError_t launch_idx_scan(hipStream_t s, uint32_t *counts, uint64_t n, unsigned long long *n_out);  // in-place exclusive scan, one workgroup
hipError_t launch_indexes(hipStream_t s, const uint64_t *bits, uint64_t size_bits, uint32_t *scratch,
                          uint32_t *rows_out, uint64_t capacity, unsigned long long *n_out);

// A bytes column on the device.  The reference holds ONE StringPool per pack (block): ptr[i] = off << 32 | len
// with `off` relative to that pack's own buffer (pkg/stringx/pool.go:45-58).  A resident column keeps the row
// pointers of all packs in one array and the pools concatenated in one payload buffer; pool_base[p] is where
// pack p's pool starts inside it (64 bit: the column as a whole is not limited to 4 GiB, only one pack's pool is).
// pool_base == nullptr: the whole column is one pool (offsets relative to buf).  `live` (nullable) is the compact
// list of packs a pruned scan evaluates; rows of other packs are neither read nor written.
struct BytesView {
    const uint64_t *ptr;
    const uint8_t *buf;
    const uint64_t *pool_base;  // [n_packs + 1] or nullptr
    uint64_t n_rows;
    uint64_t pack_rows;         // a multiple of 64 whenever pool_base or live is set
    uint32_t pack_shift;        // log2(pack_rows), 255 = not a power of two (divide)
    uint32_t n_live;
    const uint32_t *live;
};

// bytes column matchers (StringPool layout): mode = FilterMode 1 eq 2 ne 3 gt 4 ge 5 lt 6 le 9 range [val, val2]
hipError_t launch_bytes_match(hipStream_t s, int mode, const BytesView &v, const uint8_t *val, uint32_t vlen, const uint8_t *val2,
                              uint32_t vlen2, uint8_t *bits, const uint8_t *mask, unsigned long long *count_out /* pre-zeroed */);

// IN / NOT IN: membership of the value's canonical order key in a sorted key set (binary search);
// matching rows are OR-ed into bits (bits.Set), only mask-selected rows are tested when mask != null
// numInSetMatcher.MatchRangeVectors: bits[row] = the sorted u64 key set holds a member in [uint64(min[row]), uint64(max[row])] (swapped if reversed), & mask
hipError_t launch_range_set(hipStream_t s, int width, bool is_signed, const void *mins, const void *maxs, uint64_t n_rows, const uint64_t *keys,
                            uint32_t n_keys, uint64_t *bits, const uint64_t *mask);
// zeroes the bitset ranges of the packs with skip[p] != 0 (pack_rows a multiple of 64), one launch
hipError_t launch_zero_pruned(hipStream_t s, uint8_t *bits, const uint8_t *d_skip, uint32_t n_packs, uint64_t pack_rows, uint64_t n_rows);
hipError_t launch_inset(hipStream_t s, int width, uint32_t flavor, bool negate, const void *col, uint64_t n_rows,
                        const uint64_t *keys, uint32_t n_keys, const uint32_t *gstart, uint32_t g_buckets, uint32_t g_shift, uint8_t *bits, const uint8_t *mask);

// bytes IN / NOT IN: the set sorted + unique in StringPool form (set_ptr[i] = off << 32 | len into set_buf)
hipError_t launch_bytes_inset(hipStream_t s, bool negate, const BytesView &v, const uint64_t *set_ptr, uint32_t n_set,
                              const uint8_t *set_buf, const uint64_t *set_prefix, uint8_t *bits, const uint8_t *mask);

// bloom probe / build
hipError_t launch_bloom_probe(hipStream_t s, const uint8_t *filters, const uint64_t *offsets, uint64_t n_packs,
                              uint64_t h, uint8_t *maybe_out);
hipError_t launch_bloom_build_bytes(hipStream_t s, const BytesView &v, uint8_t *filters, const uint64_t *offsets);
// partitioned (atomic-free) build of packs [p0, p0 + np): packs of >= 4096 rows, filters of <= 2 MiB (256 segments of 2^16 bits)
size_t bloom_part_scratch_bytes(uint64_t packs, uint64_t pack_rows);
// the one-read build (round 4): fixed entry regions per (pack, segment), no count pass; *overflow_flag_dev != 0 afterwards =
// a region overflowed (heavy hitters), rebuild the batch with launch_bloom_build_part
size_t bloom_one_scratch_bytes(uint64_t packs, uint64_t pack_rows, uint64_t max_fbytes);
hipError_t launch_bloom_build_one(hipStream_t s, int width /* 0 = bytes */, const void *col, const BytesView *v, uint64_t n_rows, uint64_t pack_rows,
                                  uint64_t p0, uint64_t np, uint64_t max_fbytes, uint8_t *filters, const uint64_t *offsets, void *scratch,
                                  uint32_t **overflow_flag_dev);
uint32_t bloom_part_tiles(uint64_t pack_rows);
hipError_t launch_bloom_build_part(hipStream_t s, int width /* 0 = bytes */, const void *col, const BytesView *v, uint64_t n_rows, uint64_t pack_rows,
                                   uint64_t p0, uint64_t np, uint64_t max_filter_bytes, uint8_t *filters, const uint64_t *offsets, void *scratch);
hipError_t launch_bloom_build(hipStream_t s, int width, const void *col, uint64_t n_rows, uint64_t pack_rows,
                              uint8_t *filters, const uint64_t *offsets);

// ---- scan directly on bit-packed containers (no materialization) -----------------------------------------
struct PackedPack {        // one bit-packed container ([4][For][log2][N][words], int_bitpack.go:93-133) resident in HBM
    uint64_t word_off;     // first 64-bit word of the pack inside the words buffer
    uint64_t minv;         // For
    uint32_t n;            // values
    uint32_t log2;         // bits per value
};
struct PackedScanArgs {
    const uint64_t *words;
    const PackedPack *packs;
    uint32_t n_packs;
    uint32_t groups_per_pack;  // pack_rows / 4096
    uint64_t pack_rows;
    uint64_t n_rows;
    uint64_t lo, diff;         // key space of the 8-byte column type: hit = U(key(v) - lo) <= diff
    uint64_t key_xor;          // sign flip for signed columns (0 for unsigned)
    uint64_t wmask;            // value mask of the column type (2^(8W) - 1): T wraps, and so does the compare
    uint32_t negate;           // 1: hit = !in_range
    uint32_t want_sum;
    uint64_t *bits;            // nullable
    AggPartial *partials;      // [grid]
};
// one wave per 4096 values: lane l evaluates values l, l + 64, ...; a wave64 ballot IS the bitset word
hipError_t launch_packed_scan(hipStream_t s, int grid, const PackedScanArgs &a);
// the same for columns whose fields are at most 32 bits wide (and whose packs start 16-byte aligned): chunks staged in LDS
hipError_t launch_packed_scan_lds(hipStream_t s, int grid, const PackedScanArgs &a, uint32_t max_log2);

// ---- dictionary / run-end containers scanned as they are (round 4; knox_enc_*) -------------------------------------------
// dictionary [5]: row i of pack p = dict[dict_off[p] + codes[p * pack_rows + i]] (int_dict.go:76-103); the predicate is
// evaluated on the pack's dictionary once (a hit bitmap in LDS), then the uint16 codes stream: 2.125 B per row whatever
// the value width.  run-end [3]: run r of pack p covers rows (ends[r-1], ends[r]] of the pack (int_runend.go:84-112); the
// predicate is evaluated on the run values, hit runs are expanded into the bitset: (W + 4) B per run + 1/8 B per row.
struct EncPack {
    uint64_t off;       // dictionary: first entry; run-end: first run; raw: first value -- element index into the value pool
    uint64_t code_off;  // dictionary: first code inside the code pool (a multiple of 8: 16-byte aligned loads); run-end: first end inside the end pool
    uint64_t base, delta;  // constant / delta packs: row i = T(i * delta + base), wrapping in T (int_const.go, int_delta.go:109)
    uint32_t len;       // dictionary entries / runs / raw rows
    uint32_t rows;      // rows of the pack (pack_rows except for the column's last)
    uint32_t kind;      // encode.ContainerType the pack is SCANNED as: 1 constant, 2 delta, 3 run-end, 5 dictionary, 7 raw (every other
                        // container: decoded on the device at upload)
    uint32_t _pad;
};
struct EncScanArgs {
    const void *vals;        // dictionary entries / run values / raw values, W bytes each, one pool for all packs
    const uint16_t *codes;   // dictionary packs: pack_rows codes per pack
    const uint32_t *ends;    // run-end packs: inclusive last row (inside its pack) of every run, a pool of their own (EncPack.code_off)
    const EncPack *packs;    // [n_packs], by pack number
    const uint32_t *list;    // the packs THIS launch handles (all of one scanned kind), ascending
    uint32_t n_list;
    uint32_t width;          // W
    uint64_t pack_rows;
    uint64_t n_rows;
    uint64_t lo, diff, wmask;  // hit = ((raw(v) - lo) & wmask) <= diff, negated when `negate` (integer leaves: the reference's unsigned-difference form)
    uint32_t negate;
    uint32_t want_sum;
    uint64_t *bits;          // nullable
    AggPartial *partials;    // [grid] of this launch
};
constexpr int kEncChunkRows = 8192;      // dictionary scan: rows per workgroup step (4 waves x 4 loads x 64 lanes x 8 codes)
constexpr int kEncHistCodes = 8192;      // dictionaries up to this many entries count selected rows per code in LDS for the sum (beyond: dict[code] through the L2)
hipError_t launch_dict_scan(hipStream_t s, int grid, const EncScanArgs &a);
hipError_t launch_runend_scan(hipStream_t s, int grid_bits, int grid_runs, const EncScanArgs &a);
// constant / delta packs (raw == false: values computed from the row number, nothing read) and raw packs (values read from the pool)
hipError_t launch_flat_scan(hipStream_t s, int grid, bool raw, const EncScanArgs &a);

// bit-packed container payload -> materialized values: out[i] = T((stream >> i*log2) & mask) + minv
hipError_t launch_bitpack_decode(hipStream_t s, int width, const uint64_t *packed, uint64_t n_words, uint32_t log2, uint64_t minv,
                                 uint64_t n_values, void *dst);

// constant / delta containers: dst[i] = T(i) * delta + base
hipError_t launch_affine_fill(hipStream_t s, int width, uint64_t base, uint64_t delta, uint64_t n, void *dst);

// simple8 container payload -> values; scratch: n_words + 1 uint32, n_out: device counter (receives the decoded total)
hipError_t launch_s8b_decode(hipStream_t s, int width, const uint64_t *words, uint64_t n_words, uint64_t minv, uint64_t n, uint32_t *scratch,
                             unsigned long long *n_out, void *dst);
// ALP: dst[i] = T(enc[i]) * f * e (f, e as IEEE bits of T), then dst[pos[k]] = patches[k]
hipError_t launch_alp_decode(hipStream_t s, int width, const void *enc, uint64_t f_bits, uint64_t e_bits, uint64_t n, void *dst);
hipError_t launch_patch_scatter(hipStream_t s, int width, const void *patches, const uint32_t *pos, uint64_t n_patch, uint64_t n, void *dst);
// ALP-RD: bits = uint(left[i]) << shift | right[i]
hipError_t launch_alprd_merge(hipStream_t s, int width, const uint16_t *left, const void *right, uint32_t shift, uint64_t n, void *dst);
// string containers: StringPool rows from decoded (ofs, len[, code]) children; bounds check of a ptr vector
hipError_t launch_str_ptr(hipStream_t s, const uint32_t *ofs, const uint32_t *len, uint64_t n_entries, const uint32_t *code, uint64_t n,
                          uint64_t base, uint64_t *dst);
hipError_t launch_ptr_check(hipStream_t s, const uint64_t *ptr, uint64_t n, uint64_t pool_len, uint32_t *flag /* pre-zeroed */);
// synthetic bytes column (DESIGN.md "synthetic data", kind 5): row i of pack p holds key k = (p / packs_per_window) <<
// window_bits | (r(i) & (2^window_bits - 1)); key k's bytes are the little-endian words mix64(8k + j + keymix); the
// row's payload sits at (i % pack_rows) * key_len inside its pack's pool
hipError_t launch_generate_bytes(hipStream_t s, uint64_t seed, uint64_t first_row, uint64_t n_rows, uint64_t pack_rows, uint32_t key_len,
                                 uint32_t window_bits, uint32_t packs_per_window, uint64_t pool_stride, uint64_t *ptr, uint8_t *buf);
// dictionary / run-end containers over already decoded children
hipError_t launch_dict_gather(hipStream_t s, int width, const void *dict, uint64_t n_dict, const uint16_t *codes, uint64_t n, void *dst);
hipError_t launch_runend_expand(hipStream_t s, int width, const void *values, const uint32_t *ends, uint64_t n_runs, uint64_t n, void *dst);

// synthetic columns
hipError_t launch_generate(hipStream_t s, int width, bool to_f32, int kind, uint64_t seed, uint64_t first_row,
                           uint64_t n_rows, int64_t lo, uint64_t span, void *dst);

// literal sequential reducer (one lane)
// out-of-order time windows (knox_sort.hip): stable LSD radix sort of (window, row, value) triples, 1..4 passes of <= 8 bits
constexpr int kWindowSortItems = 16;  // round 4 sweep (2^28 shuffled rows, 1024 / 16384 / 2^20 windows): 6: 3.26 / 3.95 / 9.72 ms, 8: 2.99 / 3.83 / 11.8, 12: 2.95 / 3.59 / 9.34, 16: 2.88 / 3.50 / 9.13  // elements per thread and tile of the scatter kernel
struct WindowSortGeom {
    uint32_t nb;         // workgroups
    uint64_t per_block;  // elements per workgroup (a multiple of the tile)
};
WindowSortGeom window_sort_geom(uint64_t n, int n_cus);
void window_sort_passes(unsigned key_bits, uint32_t bits[4], int *n_passes);
// keys[n_rows] (0xffffffff = row not selected / outside every window); hist[nb*256] scanned, tot[256] = per-digit totals
// idx_out == nullptr: the sort carries (key, value) pairs, no row ids
hipError_t launch_window_sort_begin(hipStream_t s, const int64_t *time, uint64_t n_rows, const uint8_t *mask, int64_t t_from, int64_t step,
                                    uint32_t n_buckets, WindowSortGeom g, uint32_t bits0, uint32_t *keys, uint32_t *hist, uint32_t *tot);
hipError_t launch_window_sort_first(hipStream_t s, int width, const uint32_t *keys, const void *value_col, uint64_t n_rows, WindowSortGeom g,
                                    uint32_t bits0, const uint32_t *hist, const uint32_t *tot, uint32_t *keys_out, uint32_t *idx_out,
                                    void *val_out, const uint32_t *idx_in = nullptr);
// a selection in front of the sort: the selected rows' (window index, [row id], value) compacted in row order by a streaming pass
struct WindowCompactGeom {
    uint64_t unit_rows;   // rows per unit (one wave each), a multiple of 256
    uint32_t n_units;
};
WindowCompactGeom window_compact_geom(uint64_t n_rows, int n_cus);
hipError_t launch_mask_unit_offsets(hipStream_t s, const uint8_t *mask, uint64_t n_rows, WindowCompactGeom g, uint32_t *counts,
                                    unsigned long long *total_dev);
hipError_t launch_window_compact(hipStream_t s, int width, const int64_t *time, const void *val, const uint8_t *mask, uint64_t n_rows,
                                 WindowCompactGeom g, const uint32_t *unit_off, int64_t t_from, int64_t step, uint32_t n_buckets, uint32_t *keys_out,
                                 uint32_t *idx_out, void *val_out);
// out-of-order time without the sort: Value() of the order-free reducers per window by LDS atomics (knox_fold.hip)
struct WindowDirectGeom {
    uint32_t grid, n_slots, n_words;   // grid 0: no order-free form for (reducer, column type), or too many windows
    uint32_t cap, n_ranges;            // windows per table; ranges of windows (each read all rows)
    size_t lds, scratch;
};
WindowDirectGeom window_direct_geom(int width, int dtype, int func, uint32_t n_buckets, int n_cus, bool masked);
hipError_t launch_window_direct(hipStream_t s, int width, int dtype, int func, WindowDirectGeom g, const int64_t *time, const void *val, const uint8_t *mask,
                                uint64_t n_rows, int64_t t_from, int64_t step, uint32_t n_buckets, void *scratch, void *values, uint8_t *seen);
// ... float sums / means: exact, as 160-bit fixed-point numbers (five 64-bit adders per window)
struct WindowFsumGeom {
    uint32_t grid, base, nbins;   // grid 0: not usable; base: the lowest biased exponent held; nbins: 32-bit digits (<= 5)
    uint32_t cap, n_ranges;       // windows per table; ranges of windows (each read all rows)
    size_t lds, scratch;
};
hipError_t launch_float_exprange(hipStream_t s, int width, int dtype, const void *val, uint64_t n_rows, uint32_t *out /* [max, ~min], zeroed */);
WindowFsumGeom window_fsum_geom(uint32_t exp_max, uint32_t exp_min, uint32_t n_buckets, int n_cus, bool masked);
hipError_t launch_window_fsum(hipStream_t s, int width, int dtype, int func, WindowFsumGeom g, const int64_t *time, const void *val, const uint8_t *mask,
                              uint64_t n_rows, int64_t t_from, int64_t step, uint32_t n_buckets, void *scratch, void *values, uint8_t *seen);
struct WindowFvarGeom {          // var / std of a float column: adders for the sum and for the sum of squares
    uint32_t grid, base, nbins1, nbins2;
    size_t lds, scratch;
};
WindowFvarGeom window_fvar_geom(uint32_t exp_max, uint32_t exp_min, uint32_t n_buckets, int n_cus);
hipError_t launch_window_fvar(hipStream_t s, int width, int dtype, int func, WindowFvarGeom g, const int64_t *time, const void *val, const uint8_t *mask,
                              uint64_t n_rows, int64_t t_from, int64_t step, uint32_t n_buckets, void *scratch, void *values, uint8_t *seen);
WindowFsumGeom window_fstate_geom(uint32_t exp_max, uint32_t exp_min, uint32_t n_buckets, int n_cus);      // ... whole states of a float column
hipError_t launch_window_fstate(hipStream_t s, int width, int dtype, WindowFsumGeom g, const int64_t *time, const void *val, const uint8_t *mask, uint64_t n_rows,
                                int64_t t_from, int64_t step, uint32_t n_buckets, uint64_t row_base, void *scratch, AggStateDev *states_out);
// ... and the whole state (integer columns): one walk below 8 bytes, two for 8-byte values
struct WindowStateGeom {
    uint32_t grid, n_slots, n_words;   // grid 0: not an integer column, or the table does not fit LDS
    size_t lds, scratch;
};
WindowStateGeom window_state_geom(int width, int dtype, uint32_t n_buckets, int n_cus);
hipError_t launch_window_state(hipStream_t s, int width, int dtype, WindowStateGeom g, const int64_t *time, const void *val, const uint8_t *mask, uint64_t n_rows,
                               int64_t t_from, int64_t step, uint32_t n_buckets, uint64_t row_base, void *scratch, AggStateDev *states_out);
hipError_t launch_compact_keys_hist(hipStream_t s, const uint32_t *keys, uint64_t n, WindowSortGeom g, uint32_t bits0, uint32_t *hist, uint32_t *tot);
hipError_t launch_window_sort_pass(hipStream_t s, int width, const uint32_t *keys_in, const uint32_t *idx_in, const void *val_in, uint64_t n,
                                   WindowSortGeom g, uint32_t shift, uint32_t nbits, uint32_t *hist, uint32_t *tot, uint32_t *keys_out,
                                   uint32_t *idx_out, void *val_out);
hipError_t launch_key_bounds(hipStream_t s, const uint32_t *keys, uint64_t n, uint32_t n_buckets, uint64_t *start);
hipError_t launch_remap_rows(hipStream_t s, AggStateDev *states, uint32_t n_buckets, const uint32_t *idx, uint64_t row_base);
// per-window work items planned on the device: item_start[0..n_buckets] (last = item count), then the items
uint32_t window_plan_blocks(uint32_t n_buckets, uint32_t *per_block);
hipError_t launch_window_plan(hipStream_t s, const uint64_t *start, uint32_t n_buckets, uint32_t chunk_shift, uint32_t *block_sum,
                              uint32_t *item_start);
hipError_t launch_window_items(hipStream_t s, const uint64_t *start, const uint32_t *item_start, uint32_t n_buckets, uint32_t chunk_shift,
                               uint32_t n_items, BucketItem *items);
// Reducer.Value() per window (knox_finalize.h): values[k] of `width` bytes, seen[k]
hipError_t launch_window_values(hipStream_t s, int width, const AggStateDev *states, uint32_t n_buckets, int func, void *values, uint8_t *seen);
// compiled REGEXP program on the device (knox_regex.h): table | flags | ascii_class are one contiguous image
struct RegexProg {
    const uint16_t *direct;  // [n_states][128]: table[state][ascii_class[byte]] folded, or nullptr (image too large for LDS)
    const uint16_t *table;
    const uint8_t *flags;
    const uint8_t *ascii_class;
    const uint32_t *bound_lo;
    const uint8_t *bound_class;
    uint32_t n_bounds, n_classes, n_states;
    uint32_t lds_image_bytes;  // bytes of [direct] | table | flags | ascii the kernel stages
};
hipError_t launch_regexp_match(hipStream_t s, const RegexProg &pg, uint32_t image_bytes, const BytesView &v, uint8_t *bits, const uint8_t *mask);
hipError_t launch_bucket_sequential(hipStream_t s, int dtype, int func, const int64_t *time, const void *col, const uint8_t *mask,
                                    const uint64_t *start, uint32_t n_buckets, int wave_per_window, void *out_v, uint8_t *out_seen);
hipError_t launch_reduce_sequential(hipStream_t s, int dtype, int func, const void *col, uint64_t n_rows,
                                    const uint8_t *mask, uint64_t *out3 /* value bits, seen, row */);

}  // namespace knox
