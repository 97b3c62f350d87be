// Time windows beyond the plain ordered reduction (internal/reducer/bucket_native.go:104-159):
//
//  * out-of-order timestamps (bucket_native.go:123-141: Push finds or inserts the window's reducer by time, so rows may
//    arrive in any time order): the SELECTED rows are brought into window order by a stable LSD radix sort written for
//    exactly this shape -- the key range is known (window index < n_buckets, so 1..4 passes of <= 8 bits), rows that are
//    not selected or fall outside every window are dropped in the first pass, and the VALUE travels with the (window, row)
//    pair, so there is no random gather afterwards and the reduction reads a contiguous, window-ordered copy.  Within a
//    window the rows keep their stream order, which is the only order a reducer can observe.
//  * work-item planning for the per-window reduction on the device (no host round trip of the window bounds), and
//  * Reducer.Value() per window on the device (knox_finalize.h): 8 bytes per window cross PCIe instead of 144.
//
// Sort structure per pass (classic reduce-then-scan, no inter-workgroup waiting): every workgroup owns one contiguous
// range of the input; (1) digit histogram per workgroup, (2) scan over workgroups per digit, (3) scatter: the range is
// walked tile by tile, each tile is ranked stably (wave-level match-any over the digit bits, per-wave LDS counters),
// reordered through LDS and written out as contiguous per-digit runs.
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "knox_finalize.h"
#include "knox_kernels.h"
#include "knox_window_dev.h"

namespace knox {

namespace {
constexpr int kSortBlock = 256;
constexpr int kSortWaves = kSortBlock / 64;
constexpr int kRadix = 256;
// exclusive scan of one value per thread over the 256 threads of the workgroup; *total = sum.  `tmp` = 4 LDS words.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *tmp, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += t;
    }
    __syncthreads();  // tmp may still be read by the previous use
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) {
        const uint32_t t = tmp[w];
        if ((uint32_t)w < wave) base += t;
        all += t;
    }
    *total = all;
    return base + inc - v;
}
}  // namespace

// ---- pass 0, step 1: window keys + digit histogram ------------------------------------------------------------------
// key = window index of the row's timestamp; rows that are not selected or fall outside every window get kNoKey and
// leave the sort in the first scatter
__global__ void __launch_bounds__(kSortBlock) window_keys_hist_kernel(const int64_t *__restrict__ time, uint64_t n_rows, const uint8_t *__restrict__ mask,
                                                                      int64_t t_from, int64_t step, uint32_t n_buckets, uint64_t per_block,
                                                                      uint32_t dmask, uint32_t *__restrict__ keys, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[kRadix];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t span = (uint64_t)n_buckets * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    const uint64_t begin = (uint64_t)blockIdx.x * per_block;
    const uint64_t end = begin + per_block < n_rows ? begin + per_block : n_rows;
    // four independent (timestamp, selection byte) loads per thread and loop iteration: with one, 89 % of the wave cycles
    // were waits (3.7 TB/s, profiles/r04_windows.txt); the timestamp is loaded whether or not the row is selected -- its
    // 64-byte sector is fetched anyway -- so the loads do not wait for the selection bits
    constexpr int U = 4;
    for (uint64_t i0 = begin + threadIdx.x; i0 < end; i0 += (uint64_t)kSortBlock * U) {
        int64_t t[U];
        bool sel[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + (uint64_t)u * kSortBlock;
            const bool in = i < end;
            t[u] = in ? time[i] : 0;
            sel[u] = in && (mask == nullptr || ((mask[i >> 3] >> (i & 7)) & 1));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + (uint64_t)u * kSortBlock;
            if (i >= end) break;
            const uint32_t k = sel[u] ? window_index(t[u], t_from, (uint64_t)step, span, inv) : kNoKey;
            keys[i] = k;
            if (k != kNoKey) atomicAdd(&h[k & dmask], 1u);
        }
    }
    __syncthreads();
    hist[(uint64_t)blockIdx.x * kRadix + threadIdx.x] = h[threadIdx.x];
}

// ---- pass 0 behind a selection: compaction first ------------------------------------------------------------------------------
// With a quarter of the rows selected the first scatter ranks, tile by tile, 4096 rows to keep 1024: it reads the key and the value
// of EVERY row at its 3.5 TB/s (1.24 of the 2.5 ms of 2^28 shuffled rows -> 1024 windows).  A streaming pass does that part:
// the column is cut into units of a few thousand rows, one wave each; (A) selected rows per unit (the bitset alone: 1/8 B per
// row), an exclusive scan over the units, (B) the wave walks its unit and writes (window index, [row id], value) of the selected
// rows to their place, in row order (ballots give every lane the selected rows before its own).  Rows outside every window keep
// kNoKey and leave in the scatter, which now runs over the selected rows only.
// Measured (2^28 rows, 25 % selected, whole calls on one box, KNOX_WINDOW_COMPACT=0 / 1): 1024 windows 2.56 / 2.54 ms with the
// first form of (B), 2.3-2.5 ms with the vector form below; 16384 windows 2.95-3.08 / 2.6-2.8 ms.  (B) itself runs at 4.5-4.9 TB/s
// (16 B read per row whatever the selection -- at 25 % every 64-byte line of the values holds a selected row -- + 12 B written per
// selected one); what was tried to raise it and changed nothing: 2 .. 64 waves per SIMD-slot's worth of units, unit sizes that are
// not powers of two, nontemporal loads and stores (kept), a three-stage software pipeline.
__global__ void __launch_bounds__(kSortBlock) mask_unit_count_kernel(const uint8_t *__restrict__ mask, uint64_t n_rows, uint64_t unit_rows,
                                                                     uint32_t n_units, uint32_t *__restrict__ counts) {
    const uint32_t u = blockIdx.x * kSortWaves + (threadIdx.x >> 6), lane = lane_id();
    if (u >= n_units) return;
    const uint64_t r0 = (uint64_t)u * unit_rows, r1 = r0 + unit_rows < n_rows ? r0 + unit_rows : n_rows;
    const uint64_t b0 = r0 >> 3, b1 = (r1 + 7) >> 3;   // unit_rows is a multiple of 256: whole bytes, the last unit's last byte may be ragged
    uint32_t c = 0;
    for (uint64_t b = b0 + lane; b < b1; b += 64) {
        uint32_t x = mask[b];
        if ((b + 1) * 8 > n_rows) x &= 0xffu >> (8 - (uint32_t)(n_rows & 7));
        c += (uint32_t)__builtin_popcount(x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0) counts[u] = c;
}

// VEC: the clock is 16-byte and the value column 2-value aligned (device allocations are; a wrapped column may not be) -- a lane takes
// two NEIGHBOURING rows of each 128-row half of the chunk, so that every load instruction of the wave covers one contiguous kilobyte
// (a lane with four consecutive rows made each instruction touch every second 16 bytes of two kilobytes: four times the lines per
// instruction through the vector L1, 1.35 ms for 2^28 rows at 25 % selected), and the selected rows of a chunk are put in order in
// the wave's own piece of LDS first and leave it as whole runs -- one store instruction per array and 64 selected rows.
template <typename U, bool VEC>
__global__ void __launch_bounds__(kSortBlock) window_compact_kernel(const int64_t *__restrict__ time, const U *__restrict__ val, const uint8_t *__restrict__ mask,
                                                                    uint64_t n_rows, uint64_t unit_rows, uint32_t n_units, const uint32_t *__restrict__ unit_off,
                                                                    int64_t t_from, int64_t step, uint32_t n_buckets, uint32_t *__restrict__ keys_out,
                                                                    uint32_t *__restrict__ idx_out, U *__restrict__ val_out) {
    __shared__ uint32_t s_key[VEC ? kSortWaves : 1][256], s_idx[VEC ? kSortWaves : 1][256];
    __shared__ U s_val[VEC ? kSortWaves : 1][256];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: the chunk addresses stay in scalar registers)
    const uint32_t u = blockIdx.x * kSortWaves + wave, lane = lane_id();
    if (u >= n_units) return;
    const uint64_t r0 = (uint64_t)u * unit_rows, r1 = r0 + unit_rows < n_rows ? r0 + unit_rows : n_rows;
    uint64_t out = unit_off[u];
    const uint64_t span = (uint64_t)n_buckets * (uint64_t)step;
    const double inv = 1.0 / (double)step;
    uint64_t c0 = r0;
    if constexpr (VEC) {
        // 512 rows per round, every load of the round asked for before the first is used: four selection bytes, four 16-byte pieces of
        // the clock and four of the values per lane -- 8 KiB per wave and round trip to memory.  (With 256 rows per round and the value
        // loads behind the selection bits -- two dependent round trips for 4 KiB -- the waves waited 62 % of their cycles; in this form
        // 26 %, at the same 1.1 ms.)  Where a unit is thinly selected (under a sixteenth of its rows) the value loads do wait for the
        // bits, and lines without a selected row are not fetched.
        typedef long long t2 __attribute__((ext_vector_type(2)));
        const bool dense = (uint64_t)(unit_off[u + 1] - unit_off[u]) * 16 >= r1 - r0;
        const uint32_t sh = (2u * lane) & 7u;
        for (; c0 + 512 <= r1; c0 += 512) {
            uint32_t mb[4];
            t2 t[4];
            CompactPair<U> x[4];
#pragma unroll
            for (int h = 0; h < 4; h++) mb[h] = mask[(c0 + 128u * h + 2u * lane) >> 3];
#pragma unroll
            for (int h = 0; h < 4; h++) t[h] = __builtin_nontemporal_load(reinterpret_cast<const t2 *>(time + c0 + 128u * h + 2u * lane));
            if (dense) {
#pragma unroll
                for (int h = 0; h < 4; h++) x[h] = __builtin_nontemporal_load(reinterpret_cast<const CompactPair<U> *>(val + c0 + 128u * h + 2u * lane));
            } else {
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    x[h] = CompactPair<U>{0, 0};
                    if ((mb[h] >> sh) & 3u) x[h] = __builtin_nontemporal_load(reinterpret_cast<const CompactPair<U> *>(val + c0 + 128u * h + 2u * lane));
                }
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {     // the wave's piece of LDS holds 256 rows: two turns
                uint32_t total = 0;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int h = 2 * half + k;
                    const uint32_t m2 = (mb[h] >> sh) & 3u;
                    const uint64_t ba = __ballot(m2 & 1u), bb = __ballot(m2 & 2u);
                    uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(ba >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ba, 0u)) +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t)(bb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb, 0u));
                    total += (uint32_t)__popcll(ba) + (uint32_t)__popcll(bb);
                    const uint32_t row = (uint32_t)c0 + 128u * h + 2u * lane;
                    if (m2 & 1u) {
                        s_key[wave][pos] = window_index(t[h].x, t_from, (uint64_t)step, span, inv);
                        s_idx[wave][pos] = row;
                        s_val[wave][pos] = x[h].x;
                        pos++;
                    }
                    if (m2 & 2u) {
                        s_key[wave][pos] = window_index(t[h].y, t_from, (uint64_t)step, span, inv);
                        s_idx[wave][pos] = row + 1;
                        s_val[wave][pos] = x[h].y;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = lane; i < total; i += 64) {
                    __builtin_nontemporal_store(s_key[wave][i], keys_out + out + i);
                    if (idx_out != nullptr) __builtin_nontemporal_store(s_idx[wave][i], idx_out + out + i);
                    __builtin_nontemporal_store(s_val[wave][i], val_out + out + i);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                out += total;
            }
        }
    }
    // the general form: a lane takes four consecutive rows; the ragged end of the column, and columns without the alignment
    for (; c0 < r1; c0 += 256) {
        const uint64_t row = c0 + 4ull * lane;      // this lane's four rows
        uint32_t nib = 0;
        int64_t t[4] = {0, 0, 0, 0};
        U x[4] = {0, 0, 0, 0};
        if (row < r1) {
            nib = (mask[row >> 3] >> (row & 7)) & 0xfu;              // row is a multiple of 4: one nibble
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (row + j < r1) { t[j] = time[row + j]; x[j] = val[row + j]; }
                else nib &= ~(1u << j);
            }
        }
        uint32_t before = 0, total = 0;            // selected rows of the chunk in front of this lane's, and in all
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t b = __ballot((nib >> j) & 1u);
            before += __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            total += (uint32_t)__popcll(b);
        }
        uint64_t o = out + before;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((nib >> j) & 1u) {
                keys_out[o] = window_index(t[j], t_from, (uint64_t)step, span, inv);
                if (idx_out != nullptr) idx_out[o] = (uint32_t)(row + j);
                val_out[o] = x[j];
                o++;
            }
        }
        out += total;
    }
}

// units of the compaction: a few per CU and wave slot, whole multiples of 512 rows (the vector form's round)
WindowCompactGeom window_compact_geom(uint64_t n_rows, int n_cus) {
    WindowCompactGeom g;
    uint64_t units = (uint64_t)n_cus * 64;
    uint64_t per = (n_rows + units - 1) / units;
    per = (per + 511) / 512 * 512;
    if (per < 1024) per = 1024;
    g.unit_rows = per;
    g.n_units = (uint32_t)((n_rows + per - 1) / per);
    return g;
}

// counts[0 .. n_units] <- exclusive offsets, counts[n_units] = *total_dev = the selected rows
hipError_t launch_mask_unit_offsets(hipStream_t s, const uint8_t *mask, uint64_t n_rows, WindowCompactGeom g, uint32_t *counts,
                                    unsigned long long *total_dev) {
    hipLaunchKernelGGL(mask_unit_count_kernel, dim3((g.n_units + kSortWaves - 1) / kSortWaves), dim3(kSortBlock), 0, s, mask, n_rows, g.unit_rows,
                       g.n_units, counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_idx_scan(s, counts, g.n_units, total_dev);
}

hipError_t launch_window_compact(hipStream_t s, int width, const int64_t *time, const void *val, const uint8_t *mask, uint64_t n_rows,
                                 WindowCompactGeom g, const uint32_t *unit_off, int64_t t_from, int64_t step, uint32_t n_buckets, uint32_t *keys_out,
                                 uint32_t *idx_out, void *val_out) {
    const dim3 grid((g.n_units + kSortWaves - 1) / kSortWaves), block(kSortBlock);
    const char *form = getenv("KNOX_COMPACT_FORM");   // "rows": the general form for every chunk (tests)
    const bool vec = ((uintptr_t)time & 15) == 0 && ((uintptr_t)val % (2 * (size_t)width)) == 0 && !(form != nullptr && strcmp(form, "rows") == 0);
#define KNOX_COMPACT(U)                                                                                                                                    \
    if (vec)                                                                                                                                               \
        hipLaunchKernelGGL((window_compact_kernel<U, true>), grid, block, 0, s, time, (const U *)val, mask, n_rows, g.unit_rows, g.n_units, unit_off, t_from, \
                           step, n_buckets, keys_out, idx_out, (U *)val_out);                                                                              \
    else                                                                                                                                                   \
        hipLaunchKernelGGL((window_compact_kernel<U, false>), grid, block, 0, s, time, (const U *)val, mask, n_rows, g.unit_rows, g.n_units, unit_off, t_from, \
                           step, n_buckets, keys_out, idx_out, (U *)val_out)
    switch (width) {
    case 8: KNOX_COMPACT(uint64_t); break;
    case 4: KNOX_COMPACT(uint32_t); break;
    case 2: KNOX_COMPACT(uint16_t); break;
    case 1: KNOX_COMPACT(uint8_t); break;
    default: return hipErrorInvalidValue;
    }
#undef KNOX_COMPACT
    return hipGetLastError();
}

// ---- later passes, step 1: digit histogram of the keys as the previous pass left them --------------------------------
__global__ void __launch_bounds__(kSortBlock) key_hist_kernel(const uint32_t *__restrict__ keys, uint64_t n, uint64_t per_block, uint32_t shift,
                                                              uint32_t dmask, uint32_t *__restrict__ hist, int skip_nokey) {
    __shared__ uint32_t h[kRadix];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t begin = (uint64_t)blockIdx.x * per_block;
    const uint64_t end = begin + per_block < n ? begin + per_block : n;
    // (Round 4: lanes with equal digits found by match-any ballots, one add per group instead of one per lane, in both histogram kernels:
    //  2^28 shuffled rows -> 1024 windows 2.75 ms against 2.51 ms with the plain adds -- dropped.)
    constexpr int U = 4;    // independent loads per thread and round (one: 92 % of the wave cycles were waits, 2.8 TB/s)
    for (uint64_t i0 = begin + threadIdx.x; i0 < end; i0 += (uint64_t)kSortBlock * U) {
        uint32_t k[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + (uint64_t)u * kSortBlock;
            k[u] = i < end ? keys[i] : kNoKey;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool in = i0 + (uint64_t)u * kSortBlock < end;
            if (in && (!skip_nokey || k[u] != kNoKey)) atomicAdd(&h[(k[u] >> shift) & dmask], 1u);   // (compacted keys still hold the rows outside every window)
        }
    }
    __syncthreads();
    hist[(uint64_t)blockIdx.x * kRadix + threadIdx.x] = h[threadIdx.x];
}

// ---- step 2: one workgroup per digit: hist[b][d] -> number of elements with digit d in workgroups < b; tot[d] = all ----
__global__ void __launch_bounds__(kSortBlock) hist_scan_kernel(uint32_t *__restrict__ hist, uint32_t nb, uint32_t *__restrict__ tot) {
    __shared__ uint32_t tmp[kSortWaves];
    const uint32_t d = blockIdx.x;
    const uint32_t per = (nb + kSortBlock - 1) / kSortBlock;
    const uint32_t b0 = threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) sum += hist[(uint64_t)b * kRadix + d];
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, tmp, &total);
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t c = hist[(uint64_t)b * kRadix + d];
        hist[(uint64_t)b * kRadix + d] = run;
        run += c;
    }
    if (threadIdx.x == 0) tot[d] = total;
}

// ---- step 3: stable scatter ------------------------------------------------------------------------------------------
// U = the value's storage type (uint8_t .. uint64_t).  FIRST: the input is (keys[i], row i, value_col[i]) with kNoKey
// rows dropped; otherwise the (key, row, value) triples of the previous pass.  ITEMS elements per thread and tile; wave w
// of a tile owns the elements [w*64*ITEMS, (w+1)*64*ITEMS) of it, element (j, lane) at j*64 + lane (coalesced loads,
// element order == offset order, which is what makes the ranking stable).
template <typename U, bool FIRST, bool IDX, int ITEMS>
__global__ void __launch_bounds__(kSortBlock) sort_scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ idx_in,
                                                                  const U *__restrict__ val_in, uint64_t n, uint64_t per_block, uint32_t shift,
                                                                  uint32_t nbits, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ tot,
                                                                  uint32_t *__restrict__ keys_out, uint32_t *__restrict__ idx_out,
                                                                  U *__restrict__ val_out) {
    constexpr int TILE = kSortBlock * ITEMS;
    __shared__ uint32_t run_off[kRadix];             // next free output slot per digit for this workgroup
    __shared__ uint32_t cnt[kSortWaves][kRadix];     // per-wave digit counters of the tile -> offset of the wave inside the digit
    __shared__ uint32_t tile_start[kRadix];          // first sorted position of the digit inside the tile
    __shared__ uint32_t gbase[kRadix];               // output slot of that position
    __shared__ uint32_t tmp[kSortWaves + 1];
    __shared__ uint32_t s_key[TILE], s_idx[IDX ? TILE : 1];
    __shared__ U s_val[TILE];
    const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const uint32_t dmask = (1u << nbits) - 1;
    // IDX = false (idx_out == nullptr at the launch): the caller wants no row ids (Value() per window never shows them:
    // knox_reduce_buckets_values) -- the triples travel as (key, value) pairs, 4 of 16 bytes per element and pass less, and 16 KiB less
    // of LDS.  (Round 4: window indexes as 2-byte keys in memory when n_buckets <= 65535 were built as well and measured on the same box,
    // 2^28 shuffled rows, 25 % selected -> 1024 windows: 2.94 ms against 2.49 ms with 4-byte keys -- the 2-byte stores cost more than
    // the bytes they save; dropped.)
    constexpr bool with_idx = IDX;
    {
        tile_start[tid] = tot[tid];
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t d = 0; d < tid; d++) base += tile_start[d];
        run_off[tid] = base + hist[(uint64_t)blockIdx.x * kRadix + tid];
        __syncthreads();
    }
    const uint64_t begin = (uint64_t)blockIdx.x * per_block;
    const uint64_t end = begin + per_block < n ? begin + per_block : n;
    // Round 4, two restructurings of this loop built and measured at 1024 / 16384 / 2^20 windows of 2^28 shuffled rows (whole calls,
    // this form: 2.79-2.88 / 3.38-3.50 / 9.1-9.7 ms), both dropped:
    //   * the NEXT tile's elements loaded before the current tile's four phases (two register sets): 2.94 / 3.71 / 11.5 ms;
    //   * no reorder through LDS, every element stored from its registers straight to its output slot (7 KiB of LDS, four
    //     workgroups per CU instead of two): 3.63 / 4.90 / 12.1 ms -- the coalesced runs of the staged form are worth more than
    //     the occupancy.
    for (uint64_t tile0 = begin; tile0 < end; tile0 += TILE) {
#pragma unroll
        for (int w = 0; w < kSortWaves; w++) cnt[w][tid] = 0;
        __syncthreads();
        uint32_t key[ITEMS], rank[ITEMS], row[ITEMS] = {};
#ifndef KNOX_SORT_RMW_RANK
        uint32_t held[ITEMS], lead[ITEMS];
#endif
        U val[ITEMS];
        // phase 1: load + rank inside the wave's chunk
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const uint64_t e = tile0 + (uint64_t)wave * (64 * ITEMS) + (uint64_t)j * 64 + lane;
            key[j] = kNoKey;
            if (e < end) {
                key[j] = keys_in[e];
                if (FIRST) {
                    row[j] = (with_idx && idx_in != nullptr) ? idx_in[e] : (uint32_t)e;   // compacted input: the caller's row ids travel with it
                } else if (with_idx) {
                    row[j] = idx_in[e];
                }
                val[j] = val_in[e];
            }
        }
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const bool active = key[j] != kNoKey;  // later passes: always true below `end` (kNoKey never survives pass 0)
            const uint32_t d = (key[j] >> shift) & dmask;
            uint64_t peers = __ballot(active);
            for (uint32_t b = 0; b < nbits; b++) {
                const bool bit = (d >> b) & 1;
                const uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            const uint32_t group = (uint32_t)__popcll(peers);
#ifdef KNOX_SORT_RMW_RANK  // A/B builds: read, then write by the group's last lane -- 16 dependent LDS round trips per wave and tile
            uint32_t c = 0;
            if (active) c = cnt[wave][d];
            // all lanes of the wave have read before any writes: LDS operations of one wave retire in order
            if (active && below == group - 1) cnt[wave][d] = c + group;
            rank[j] = c + below;
#else
            // the group's FIRST lane adds the group's size and takes the old count back (one returning LDS add per digit present);
            // nothing here waits for the previous element's answer -- LDS operations of a wave execute in order, so the sixteen
            // adds queue up behind each other and the sixteen answers are fetched afterwards (lead[j]: the lane that holds it)
            uint32_t old = 0;
            if (active && below == 0) old = atomicAdd(&cnt[wave][d], group);
            rank[j] = below;
            held[j] = old;
            lead[j] = active ? (uint32_t)__builtin_ctzll(peers) : lane;
#endif
        }
#ifndef KNOX_SORT_RMW_RANK
#pragma unroll
        for (int j = 0; j < ITEMS; j++) rank[j] += (uint32_t)__shfl((int)held[j], (int)lead[j], 64);
#endif
        __syncthreads();
        // phase 2: digit d (= tid): waves' offsets inside the digit, the digit's start inside the tile, its output slot
        {
            uint32_t total = 0, c[kSortWaves];
#pragma unroll
            for (int w = 0; w < kSortWaves; w++) {
                c[w] = total;
                total += cnt[w][tid];
            }
            uint32_t n_tile;
            const uint32_t st = block_exclusive_scan(total, tmp, &n_tile);
            // one LDS read per element in phase 3 (the wave's first slot of the digit inside the tile) and one in phase 4 (output slot
            // minus tile position) instead of two each: the scatter is bound by its LDS operations, not by HBM
#pragma unroll
            for (int w = 0; w < kSortWaves; w++) cnt[w][tid] = st + c[w];
            gbase[tid] = run_off[tid] - st;
            run_off[tid] += total;
            if (tid == 0) tmp[kSortWaves] = n_tile;
        }
        __syncthreads();
        // phase 3: into sorted position inside the tile
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            if (key[j] != kNoKey) {
                const uint32_t d = (key[j] >> shift) & dmask;
                const uint32_t pos = cnt[wave][d] + rank[j];
                s_key[pos] = key[j];
                if (with_idx) s_idx[pos] = row[j];
                s_val[pos] = val[j];
            }
        }
        __syncthreads();
        // phase 4: contiguous runs per digit
        const uint32_t n_tile = tmp[kSortWaves];
        for (uint32_t i = tid; i < n_tile; i += kSortBlock) {
            const uint32_t k = s_key[i];
            const uint32_t d = (k >> shift) & dmask;
            const uint32_t dst = gbase[d] + i;  // gbase: the digit's output slot minus its first position inside the tile (wraps mod 2^32)
            keys_out[dst] = k;
            if (with_idx) idx_out[dst] = s_idx[i];
            val_out[dst] = s_val[i];
        }
        __syncthreads();
    }
}

namespace {
template <typename U, bool IDX>
hipError_t scatter_launch2(hipStream_t s, bool first, uint32_t nb, const uint32_t *keys_in, const uint32_t *idx_in, const void *val_in, uint64_t n,
                           uint64_t per_block, uint32_t shift, uint32_t nbits, const uint32_t *hist, const uint32_t *tot, uint32_t *keys_out,
                           uint32_t *idx_out, void *val_out) {
    constexpr int ITEMS = kWindowSortItems;
    if (first)
        hipLaunchKernelGGL((sort_scatter_kernel<U, true, IDX, ITEMS>), dim3(nb), dim3(kSortBlock), 0, s, keys_in, idx_in, (const U *)val_in, n, per_block,
                           shift, nbits, hist, tot, keys_out, idx_out, (U *)val_out);
    else
        hipLaunchKernelGGL((sort_scatter_kernel<U, false, IDX, ITEMS>), dim3(nb), dim3(kSortBlock), 0, s, keys_in, idx_in, (const U *)val_in, n,
                           per_block, shift, nbits, hist, tot, keys_out, idx_out, (U *)val_out);
    return hipGetLastError();
}

template <typename U>
hipError_t scatter_launch(hipStream_t s, bool first, uint32_t nb, const uint32_t *keys_in, const uint32_t *idx_in, const void *val_in, uint64_t n,
                          uint64_t per_block, uint32_t shift, uint32_t nbits, const uint32_t *hist, const uint32_t *tot, uint32_t *keys_out,
                          uint32_t *idx_out, void *val_out) {
    return idx_out != nullptr ? scatter_launch2<U, true>(s, first, nb, keys_in, idx_in, val_in, n, per_block, shift, nbits, hist, tot, keys_out, idx_out, val_out)
                              : scatter_launch2<U, false>(s, first, nb, keys_in, idx_in, val_in, n, per_block, shift, nbits, hist, tot, keys_out, idx_out, val_out);
}

hipError_t scatter_any(hipStream_t s, int width, bool first, uint32_t nb, const uint32_t *keys_in, const uint32_t *idx_in, const void *val_in,
                       uint64_t n, uint64_t per_block, uint32_t shift, uint32_t nbits, const uint32_t *hist, const uint32_t *tot,
                       uint32_t *keys_out, uint32_t *idx_out, void *val_out) {
    switch (width) {
    case 8: return scatter_launch<uint64_t>(s, first, nb, keys_in, idx_in, val_in, n, per_block, shift, nbits, hist, tot, keys_out, idx_out, val_out);
    case 4: return scatter_launch<uint32_t>(s, first, nb, keys_in, idx_in, val_in, n, per_block, shift, nbits, hist, tot, keys_out, idx_out, val_out);
    case 2: return scatter_launch<uint16_t>(s, first, nb, keys_in, idx_in, val_in, n, per_block, shift, nbits, hist, tot, keys_out, idx_out, val_out);
    case 1: return scatter_launch<uint8_t>(s, first, nb, keys_in, idx_in, val_in, n, per_block, shift, nbits, hist, tot, keys_out, idx_out, val_out);
    default: return hipErrorInvalidValue;
    }
}
}  // namespace

WindowSortGeom window_sort_geom(uint64_t n, int n_cus) {
    WindowSortGeom g;
    const uint64_t tile = (uint64_t)kSortBlock * kWindowSortItems;
    uint64_t nb = (n + tile - 1) / tile;
    const uint64_t cap = (uint64_t)n_cus * 16;  // measured (tools/bench_windows.py, 2^28 shuffled rows): 2 -> 5.0 ms, 4 -> 4.0, 8 -> 3.55, 16 .. 32 -> 3.5, 128 -> 3.8
    if (nb > cap) nb = cap;
    if (nb == 0) nb = 1;
    g.nb = (uint32_t)nb;
    g.per_block = ((n + nb - 1) / nb + tile - 1) / tile * tile;
    return g;
}

void window_sort_passes(unsigned key_bits, uint32_t bits[4], int *n_passes) {
    const int p = key_bits <= 8 ? 1 : key_bits <= 16 ? 2 : key_bits <= 24 ? 3 : 4;
    unsigned left = key_bits;
    for (int i = 0; i < p; i++) {
        bits[i] = (left + (p - i) - 1) / (p - i);  // split evenly: 15 bits -> 8 + 7
        left -= bits[i];
    }
    *n_passes = p;
}

// keys[i] for every row + the first pass's histogram, scanned: tot[d] = selected rows with low digit d (their sum = the
// rows inside windows, read by the host to size the sort buffers)
hipError_t launch_window_sort_begin(hipStream_t s, const int64_t *time, uint64_t n_rows, const uint8_t *mask, int64_t t_from, int64_t step,
                                    uint32_t n_buckets, WindowSortGeom g, uint32_t bits0, uint32_t *keys, uint32_t *hist, uint32_t *tot) {
    hipLaunchKernelGGL(window_keys_hist_kernel, dim3(g.nb), dim3(kSortBlock), 0, s, time, n_rows, mask, t_from, step, n_buckets, g.per_block,
                       (1u << bits0) - 1, keys, hist);
    hipLaunchKernelGGL(hist_scan_kernel, dim3(kRadix), dim3(kSortBlock), 0, s, hist, g.nb, tot);
    return hipGetLastError();
}

// pass 0: (keys, row id, value_col) -> buffer A, dropping kNoKey rows
hipError_t launch_window_sort_first(hipStream_t s, int width, const uint32_t *keys, const void *value_col, uint64_t n_rows, WindowSortGeom g,
                                    uint32_t bits0, const uint32_t *hist, const uint32_t *tot, uint32_t *keys_out, uint32_t *idx_out,
                                    void *val_out, const uint32_t *idx_in) {
    return scatter_any(s, width, true, g.nb, keys, idx_in, value_col, n_rows, g.per_block, 0, bits0, hist, tot, keys_out, idx_out, val_out);
}

// pass p >= 1 over the n selected triples
hipError_t launch_window_sort_pass(hipStream_t s, int width, const uint32_t *keys_in, const uint32_t *idx_in, const void *val_in, uint64_t n,
                                   WindowSortGeom g, uint32_t shift, uint32_t nbits, uint32_t *hist, uint32_t *tot, uint32_t *keys_out,
                                   uint32_t *idx_out, void *val_out) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(key_hist_kernel, dim3(g.nb), dim3(kSortBlock), 0, s, keys_in, n, g.per_block, shift, (1u << nbits) - 1, hist, 0);
    hipLaunchKernelGGL(hist_scan_kernel, dim3(kRadix), dim3(kSortBlock), 0, s, hist, g.nb, tot);
    return scatter_any(s, width, false, g.nb, keys_in, idx_in, val_in, n, g.per_block, shift, nbits, hist, tot, keys_out, idx_out, val_out);
}

// the first digit's histogram of COMPACTED keys (kNoKey entries skipped), scanned: what launch_window_sort_begin leaves for uncompacted input
hipError_t launch_compact_keys_hist(hipStream_t s, const uint32_t *keys, uint64_t n, WindowSortGeom g, uint32_t bits0, uint32_t *hist, uint32_t *tot) {
    hipLaunchKernelGGL(key_hist_kernel, dim3(g.nb), dim3(kSortBlock), 0, s, keys, n, g.per_block, 0u, (1u << bits0) - 1, hist, 1);
    hipLaunchKernelGGL(hist_scan_kernel, dim3(kRadix), dim3(kSortBlock), 0, s, hist, g.nb, tot);
    return hipGetLastError();
}

// start[k] = first sorted position whose key is >= k, k = 0..n_buckets (start[n_buckets] = rows inside windows)
__global__ void __launch_bounds__(kSortBlock) key_bounds_kernel(const uint32_t *__restrict__ keys, uint64_t n, uint32_t n_buckets, uint64_t *__restrict__ start) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_buckets) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    start[k] = lo;
}

hipError_t launch_key_bounds(hipStream_t s, const uint32_t *keys, uint64_t n, uint32_t n_buckets, uint64_t *start) {
    hipLaunchKernelGGL(key_bounds_kernel, dim3((n_buckets + 1 + kSortBlock - 1) / kSortBlock), dim3(kSortBlock), 0, s, keys, n, n_buckets, start);
    return hipGetLastError();
}

// the states were reduced over sorted positions: map their row ids back to the caller's rows
__global__ void __launch_bounds__(kSortBlock) remap_rows_kernel(AggStateDev *states, uint32_t n_buckets, const uint32_t *__restrict__ idx, uint64_t row_base) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_buckets) return;
    AggStateDev &s = states[k];
    if (s.min_row >= 0) s.min_row = (int64_t)(idx[s.min_row] + row_base);
    if (s.max_row >= 0) s.max_row = (int64_t)(idx[s.max_row] + row_base);
    if (s.first_row >= 0) s.first_row = (int64_t)(idx[s.first_row] + row_base);
    if (s.last_row >= 0) s.last_row = (int64_t)(idx[s.last_row] + row_base);
}

hipError_t launch_remap_rows(hipStream_t s, AggStateDev *states, uint32_t n_buckets, const uint32_t *idx, uint64_t row_base) {
    hipLaunchKernelGGL(remap_rows_kernel, dim3((n_buckets + kSortBlock - 1) / kSortBlock), dim3(kSortBlock), 0, s, states, n_buckets, idx, row_base);
    return hipGetLastError();
}

// ---- work items of the per-window reduction, planned on the device ---------------------------------------------------
// window k's row range [start[k], start[k+1]) is cut into items of at most 2^chunk_shift rows.
// item_start[k] = number of items of the windows before k (k = 0..n_buckets: the last entry is the item count).
__global__ void __launch_bounds__(kSortBlock) item_count_kernel(const uint64_t *__restrict__ start, uint32_t n_buckets, uint32_t chunk_shift,
                                                                uint32_t per_block, uint32_t *__restrict__ block_sum) {
    __shared__ uint32_t tmp[kSortWaves];
    const uint64_t chunk = 1ull << chunk_shift;
    const uint32_t k0 = blockIdx.x * per_block;
    uint32_t sum = 0;
    for (uint32_t k = k0 + threadIdx.x; k < k0 + per_block && k < n_buckets; k += kSortBlock) sum += (uint32_t)((start[k + 1] - start[k] + chunk - 1) >> chunk_shift);
    uint32_t total;
    block_exclusive_scan(sum, tmp, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kSortBlock) item_block_scan_kernel(uint32_t *__restrict__ block_sum, uint32_t n_blocks) {
    __shared__ uint32_t tmp[kSortWaves];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += kSortBlock) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < n_blocks ? block_sum[b] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, tmp, &total);
        if (b < n_blocks) block_sum[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) block_sum[n_blocks] = carry;  // the item count
}

__global__ void __launch_bounds__(kSortBlock) item_start_kernel(const uint64_t *__restrict__ start, uint32_t n_buckets, uint32_t chunk_shift,
                                                                uint32_t per_block, const uint32_t *__restrict__ block_sum, uint32_t n_blocks,
                                                                uint32_t *__restrict__ item_start) {
    __shared__ uint32_t tmp[kSortWaves];
    const uint64_t chunk = 1ull << chunk_shift;
    const uint32_t k0 = blockIdx.x * per_block;
    uint32_t carry = block_sum[blockIdx.x];
    for (uint32_t kb = k0; kb < k0 + per_block && kb < n_buckets; kb += kSortBlock) {
        const uint32_t k = kb + threadIdx.x;
        const uint32_t v = (k < k0 + per_block && k < n_buckets) ? (uint32_t)((start[k + 1] - start[k] + chunk - 1) >> chunk_shift) : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, tmp, &total);
        if (k < k0 + per_block && k < n_buckets) item_start[k] = carry + ex;
        carry += total;
    }
    if (blockIdx.x == n_blocks - 1 && threadIdx.x == 0) item_start[n_buckets] = block_sum[n_blocks];
}

// one thread per item: its window by binary search over item_start, its row range from the window's start
__global__ void __launch_bounds__(kSortBlock) item_fill_kernel(const uint64_t *__restrict__ start, const uint32_t *__restrict__ item_start,
                                                               uint32_t n_buckets, uint32_t chunk_shift, BucketItem *__restrict__ items) {
    const uint32_t n_items = item_start[n_buckets];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    uint32_t lo = 0, hi = n_buckets;  // last k with item_start[k] <= i (windows without items share a value with their successor)
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (item_start[mid] <= i) lo = mid; else hi = mid;
    }
    // skip back over empty windows is not needed: item_start[lo] <= i < item_start[lo + 1] holds for the LAST such lo
    const uint64_t b = start[lo] + ((uint64_t)(i - item_start[lo]) << chunk_shift);
    const uint64_t e = b + (1ull << chunk_shift) < start[lo + 1] ? b + (1ull << chunk_shift) : start[lo + 1];
    BucketItem it;
    it.begin = b;
    it.end = e;
    it.bucket = lo;
    it._pad = 0;
    items[i] = it;
}

uint32_t window_plan_blocks(uint32_t n_buckets, uint32_t *per_block) {
    uint32_t per = 4096;
    uint32_t nb = (n_buckets + per - 1) / per;
    if (nb == 0) nb = 1;
    *per_block = per;
    return nb;
}

// item_start[0..n_buckets] from start[0..n_buckets]; block_sum = window_plan_blocks() + 1 scratch words
hipError_t launch_window_plan(hipStream_t s, const uint64_t *start, uint32_t n_buckets, uint32_t chunk_shift, uint32_t *block_sum,
                              uint32_t *item_start) {
    uint32_t per;
    const uint32_t nb = window_plan_blocks(n_buckets, &per);
    hipLaunchKernelGGL(item_count_kernel, dim3(nb), dim3(kSortBlock), 0, s, start, n_buckets, chunk_shift, per, block_sum);
    hipLaunchKernelGGL(item_block_scan_kernel, dim3(1), dim3(kSortBlock), 0, s, block_sum, nb);
    hipLaunchKernelGGL(item_start_kernel, dim3(nb), dim3(kSortBlock), 0, s, start, n_buckets, chunk_shift, per, block_sum, nb, item_start);
    return hipGetLastError();
}

hipError_t launch_window_items(hipStream_t s, const uint64_t *start, const uint32_t *item_start, uint32_t n_buckets, uint32_t chunk_shift,
                               uint32_t n_items, BucketItem *items) {
    if (n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(item_fill_kernel, dim3((n_items + kSortBlock - 1) / kSortBlock), dim3(kSortBlock), 0, s, start, item_start, n_buckets,
                       chunk_shift, items);
    return hipGetLastError();
}

// ---- Reducer.Value() per window on the device --------------------------------------------------------------------------
template <typename U>
__global__ void __launch_bounds__(kSortBlock) window_values_kernel(const AggStateDev *__restrict__ states, uint32_t n_buckets, int func,
                                                                   U *__restrict__ values, uint8_t *__restrict__ seen) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_buckets) return;
    uint64_t bits = 0;
    int sn = 0;
    int64_t row = -1;
    const int rc = knox_fin::agg_value_bits(states[k], func, &bits, &sn, &row);
    values[k] = (U)bits;
    seen[k] = (rc == 0 && sn) ? 1 : 0;
}

hipError_t launch_window_values(hipStream_t s, int width, const AggStateDev *states, uint32_t n_buckets, int func, void *values, uint8_t *seen) {
    if (n_buckets == 0) return hipSuccess;
    const dim3 g((n_buckets + kSortBlock - 1) / kSortBlock), b(kSortBlock);
    switch (width) {
    case 8: hipLaunchKernelGGL((window_values_kernel<uint64_t>), g, b, 0, s, states, n_buckets, func, (uint64_t *)values, seen); break;
    case 4: hipLaunchKernelGGL((window_values_kernel<uint32_t>), g, b, 0, s, states, n_buckets, func, (uint32_t *)values, seen); break;
    case 2: hipLaunchKernelGGL((window_values_kernel<uint16_t>), g, b, 0, s, states, n_buckets, func, (uint16_t *)values, seen); break;
    case 1: hipLaunchKernelGGL((window_values_kernel<uint8_t>), g, b, 0, s, states, n_buckets, func, (uint8_t *)values, seen); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace knox
