// knox_decode.hip -- container ingest on the device (bit-packed, simple8, ALP / ALP-RD, dictionary, run-end, constant /
// delta, string containers), the scan that runs on bit-packed words directly, and the synthetic-data generators.
#include "knox_dev.h"

namespace knox {

// ---------------------------------------------------------------------------------------------
// Bit-packed integer container decode (internal/encode/bitpack/decode.go:56-74 DecodeValue):
// value i = T((word[i*log2 >> 6] >> (i*log2 & 63) | word[+1] << (64 - shift)) & mask) + minv.
// One lane per value: the packed words are read through L1/L2 (neighbouring lanes share them),
// the materialized column is written with full-width coalesced stores.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) bitpack_decode_kernel(const uint64_t *packed, uint64_t n_words, uint32_t log2, uint64_t minv,
                                                                uint64_t n_values, void *dst) {
    const uint64_t mask = log2 >= 64 ? ~0ull : ((1ull << log2) - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        if (log2 > 0) {
            const uint64_t idx = i * log2, pos = idx >> 6;
            const uint32_t shift = (uint32_t)(idx & 63);
            uint64_t word = packed[pos] >> shift;
            if (64 - shift < log2 && pos + 1 < n_words) word |= packed[pos + 1] << (64 - shift);
            v = word & mask;
        }
        v += minv;
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

hipError_t launch_bitpack_decode(hipStream_t s, int width, const uint64_t *packed, uint64_t n_words, uint32_t log2, uint64_t minv,
                                 uint64_t n_values, void *dst) {
    if (n_values == 0) return hipSuccess;
    uint64_t g = (n_values + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    switch (width) {
    case 8: KNOX_LAUNCH((bitpack_decode_kernel<8>), (int)g, packed, n_words, log2, minv, n_values, dst);
    case 4: KNOX_LAUNCH((bitpack_decode_kernel<4>), (int)g, packed, n_words, log2, minv, n_values, dst);
    case 2: KNOX_LAUNCH((bitpack_decode_kernel<2>), (int)g, packed, n_words, log2, minv, n_values, dst);
    case 1: KNOX_LAUNCH((bitpack_decode_kernel<1>), (int)g, packed, n_words, log2, minv, n_values, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// Scan on bit-packed containers without materializing them (the reference compares on the packed
// form too: internal/encode/bitpack/cmp_*.go, int_bitpack.go Match*).  Value i of a pack is the
// log2-bit field at bit i*log2 of its word stream, plus For (bitpack/decode.go:56-74).
//
// Work shape: one wave per 4096 values; in step t lane l evaluates value t*64 + l, so
//   * the wave64 ballot of the hits IS the bitset word of rows [t*64, t*64+64) -- no bit shuffling;
//     lane t parks word t and after 64 steps the wave stores 64 words = 512 contiguous bytes;
//   * (l*log2) & 63 -- the field's shift -- is the same in every step (t*64*log2 is a multiple of 64
//     bits), only the word index advances by log2 per step; the lanes of a step read log2 + 1
//     consecutive words: one cache-line-sized access.
// Traffic: log2/8 + 1/8 bytes per row instead of 8 + 1/8.
// ---------------------------------------------------------------------------------------------
typedef uint64_t u64x2u __attribute__((ext_vector_type(2), aligned(8)));

template <bool SUM>
__global__ void __launch_bounds__(kBlock) packed_scan_kernel(const PackedScanArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_chunks = (uint64_t)a.n_packs * a.groups_per_pack;
    const uint64_t wave0 = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * kWavesPerBlock;
    const uint64_t n_words_bits = (a.n_rows + 63) >> 6;
    Acc acc;
    acc.init();
    uint64_t cnt = 0;
    for (uint64_t wc = wave0; wc < n_chunks; wc += n_waves) {
        const uint32_t pack = (uint32_t)(wc / a.groups_per_pack);
        const uint64_t base = (wc % a.groups_per_pack) * 4096;  // first value of the chunk inside its pack
        const PackedPack hp = a.packs[pack];
        const uint32_t b = hp.log2;
        const uint64_t mask = b >= 64 ? ~0ull : ((1ull << b) - 1);
        const uint32_t sh = (uint32_t)(lane * b) & 63u, wl = (uint32_t)(lane * b) >> 6;
        const uint64_t *wp = a.words + hp.word_off + (base >> 6) * b + wl;
        const uint64_t c0 = (hp.minv ^ a.key_xor) - a.lo;  // key(field + For) - lo = field + c0 when no carry crosses the sign flip:
                                                            // the flip is an xor of the top bit = an add of 2^63 mod 2^64, so it commutes
        // lane t ends up holding the ballot of step t
        uint32_t my_lo = 0, my_hi = 0;
        uint64_t sum = 0;  // of the FIELDS of the hits; For is added once per chunk: + For * hits
        const bool full = base + 4096 <= hp.n;  // only the last chunk of a short last pack needs the row bound
        // 8 steps per round: the eight 16-byte loads are issued before any of them is consumed
#pragma unroll 1
        for (uint32_t t0 = 0; t0 < 64; t0 += 8) {
            u64x2u w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                // (a short last pack: the groups behind its values are not there -- the words buffer ends 64 bytes behind the payload)
                const bool there = full || base + (uint64_t)(t0 + k) * 64 < hp.n;
                w[k] = there ? __builtin_nontemporal_load((const u64x2u *)(wp + (uint64_t)(t0 + k) * b)) : u64x2u{0ull, 0ull};
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t t = t0 + k;
                uint64_t field = w[k].x >> sh;
                if (sh != 0) field |= w[k].y << (64 - sh);
                field &= mask;
                bool hit = (((field + c0) & a.wmask) <= a.diff) != (a.negate != 0);
                if (!full) hit = hit && (base + t * 64 + (uint32_t)lane < hp.n);
                const uint64_t word = __ballot(hit);
                // lane t parks the ballot of step t (v_writelane would do it in two instructions, but it takes
                // the lane select from m0 / an inline constant only: measured slower and fragile as inline asm)
                my_lo = lane == (int)t ? (uint32_t)word : my_lo;
                my_hi = lane == (int)t ? (uint32_t)(word >> 32) : my_hi;
                if constexpr (SUM) sum += hit ? field : 0ull;
            }
        }
        const uint64_t myword = ((uint64_t)my_hi << 32) | my_lo;
        const uint64_t wi = (((uint64_t)pack * a.pack_rows + base) >> 6) + (uint64_t)lane;
        if (a.bits != nullptr && wi < n_words_bits) __builtin_nontemporal_store(myword, a.bits + wi);
        const uint64_t hits = (uint64_t)__builtin_popcountll(myword);  // lane t: hits of step t
        cnt += hits;
        if constexpr (SUM) acc.isum += sum + hp.minv * hits;  // wrapping: sum over lanes = sum of fields + For * all hits
    }
    acc.count = cnt;
    block_reduce_store(acc, a.partials + blockIdx.x);
}

// Fields of at most 32 bits, the chunk staged in LDS (round 4).  The kernel above issues one 16-byte global load per lane and STEP
// (64 values): 2^30 values are 16.8 M wave-level loads of which every 64 lanes want the same few cache lines -- the texture
// addresser, not HBM and not the ALUs, bound it at 1.2 T rows/s whatever the width (0.83 ms per 2^30 rows at 9 and at 20 bits).  Here
// a wave copies its chunk (4096 values = 512 b bytes) into its own LDS slice with coalesced 16-byte loads and the steps read two
// dwords from there (ds_read2_b32).  The predicate in field space: key - lo = field + c0 (mod 2^w) walks up by one with the field,
// so the fields that hit are ONE cyclic interval [x0, x0 + diff] mod 2^w cut to [0, 2^b): as a 32-bit test (field - s) <= d -- when
// the interval wraps, its two pieces [s, 2^b) and [0, e] are one interval mod 2^32 again, because no field lies in between.  Per
// value: funnel shift, mask, subtract, compare -- the compare's lane mask IS the bitset word -- and two v_writelane to park the word
// in lane t.
template <bool SUM>
__global__ void __launch_bounds__(kBlock) packed_scan_lds_kernel(const PackedScanArgs a, uint32_t wave_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t packed_lds[];
    const int lane = threadIdx.x & 63;
    uint32_t *stage = (uint32_t *)(packed_lds + (size_t)(threadIdx.x >> 6) * wave_bytes);
    const uint64_t n_chunks = (uint64_t)a.n_packs * a.groups_per_pack;
    const uint64_t wave0 = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * kWavesPerBlock;
    const uint64_t n_words_bits = (a.n_rows + 63) >> 6;
    Acc acc;
    acc.init();
    uint64_t cnt = 0;
    // a chunk's shape (wave uniform) and its words on their way: the NEXT chunk's 16-byte vectors are loaded into registers (at
    // most 16 per lane: 512 * 32 bytes / 64 lanes) before the current chunk's steps run from LDS -- without that the global
    // latency of every chunk is exposed, and the LDS slices leave room for two to five workgroups per CU only
    struct Chunk {
        PackedPack hp;
        uint64_t base;
        uint32_t pack, n_vec;
        bool empty, full;
    };
    constexpr int kMaxVec = 16;
    u32x4 pre[kMaxVec];
    auto shape = [&](uint64_t wc) {
        Chunk c;
        c.pack = (uint32_t)(wc / a.groups_per_pack);
        c.base = (wc % a.groups_per_pack) * 4096;  // first value of the chunk inside its pack
        c.hp = a.packs[c.pack];
        c.empty = c.base >= c.hp.n;                // a chunk behind a short last pack's values
        c.full = c.base + 4096 <= c.hp.n;
        const uint32_t groups = c.empty ? 0u : c.full ? 64u : (uint32_t)((c.hp.n - c.base + 63) >> 6);
        c.n_vec = (groups * c.hp.log2 * 8u + 15u) >> 4;     // the chunk's valid words as 16-byte vectors
        return c;
    };
    auto fetch = [&](const Chunk &c) {
        const u32x4 *src = (const u32x4 *)(a.words + c.hp.word_off + (c.base >> 6) * c.hp.log2) + lane;
#pragma unroll
        for (int k = 0; k < kMaxVec; k++)
            if ((uint32_t)(k * 64 + lane) < c.n_vec) pre[k] = __builtin_nontemporal_load(src + k * 64);
    };
    if (wave0 >= n_chunks) {
        block_reduce_store(acc, a.partials + blockIdx.x);
        return;
    }
    Chunk cur = shape(wave0);
    fetch(cur);
    for (uint64_t wc = wave0; wc < n_chunks; wc += n_waves) {
        const PackedPack hp = cur.hp;
        const uint64_t base = cur.base;
        const uint32_t pack = cur.pack;
        const bool full = cur.full, empty = cur.empty;
#pragma unroll
        for (int k = 0; k < kMaxVec; k++)
            if ((uint32_t)(k * 64 + lane) < cur.n_vec) ((u32x4 *)stage)[k * 64 + lane] = pre[k];
        if (wc + n_waves < n_chunks) {   // wave uniform
            cur = shape(wc + n_waves);
            fetch(cur);
        }
        const uint64_t wi = (((uint64_t)pack * a.pack_rows + base) >> 6) + (uint64_t)lane;
        if (empty) {
            if (a.bits != nullptr && wi < n_words_bits) __builtin_nontemporal_store(0ull, a.bits + wi);
            continue;
        }
        const uint32_t b = hp.log2;                              // <= 32 (the host picks this kernel)
        const uint64_t c0 = (hp.minv ^ a.key_xor) - a.lo;
        const uint32_t m1 = b == 32u ? 0xffffffffu : (1u << b) - 1u;              // the largest field
        const uint64_t wm = a.wmask, x0 = (0ull - c0) & wm;
        bool none = false;
        uint32_t s32 = 0, d32 = 0xffffffffu;                                       // diff == wm: every field
        if (a.diff < wm) {
            const uint64_t end = (x0 + a.diff) & wm;
            if (end >= x0) {                                                       // one piece
                none = x0 > (uint64_t)m1;
                s32 = (uint32_t)x0;
                d32 = (uint32_t)((end > (uint64_t)m1 ? (uint64_t)m1 : end) - x0);
            } else if (x0 > (uint64_t)m1) {                                        // wrapped, only the piece [0, end] reaches the fields
                s32 = 0;
                d32 = end > (uint64_t)m1 ? m1 : (uint32_t)end;
            } else {                                                               // [x0, m1] and [0, min(end, m1)], end < x0
                s32 = (uint32_t)x0;
                d32 = (end > (uint64_t)m1 ? m1 : (uint32_t)end) - s32;             // mod 2^32
            }
        }
        const bool flip = a.negate != 0;
        const uint32_t bitoff = (uint32_t)lane * b, sh32 = bitoff & 31u;
        const uint32_t *mine = stage + (bitoff >> 5);                             // group t: 2 b dwords further on
        uint32_t w_lo = 0, w_hi = 0;
        uint64_t sum = 0;  // of the FIELDS of the hits; For is added once per chunk: + For * hits
#pragma unroll 1   // (unrolled, all 128 LDS reads are hoisted: 172 VGPRs, two waves per SIMD)
        for (int t0 = 0; t0 < 64; t0 += 8) {
            uint32_t d0[8], d1[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                d0[k] = mine[(uint32_t)(t0 + k) * 2u * b];
                d1[k] = mine[(uint32_t)(t0 + k) * 2u * b + 1u];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t field = __builtin_amdgcn_alignbit(d1[k], d0[k], sh32) & m1;
                bool hit = (((field - s32) <= d32) && !none) != flip;
                if (!full) hit = hit && (base + (uint32_t)(t0 + k) * 64 + (uint32_t)lane < hp.n);
                const uint64_t word = __ballot(hit);
                // lane t parks the ballot of step t (this compiler has no __builtin_amdgcn_writelane; m0 carries the lane select: two
                // SGPR operands would exceed the constant bus)
                asm("s_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
                    : "+v"(w_lo), "+v"(w_hi)
                    : "s"((uint32_t)word), "s"(t0 + k), "s"((uint32_t)(word >> 32))
                    : "m0");
                if constexpr (SUM) sum += hit ? (uint64_t)field : 0ull;
            }
        }
        const uint64_t myword = ((uint64_t)w_hi << 32) | w_lo;
        if (a.bits != nullptr && wi < n_words_bits) __builtin_nontemporal_store(myword, a.bits + wi);
        const uint64_t hits = (uint64_t)__builtin_popcountll(myword);  // lane t: hits of step t
        cnt += hits;
        if constexpr (SUM) acc.isum += sum + hp.minv * hits;  // wrapping: sum over lanes = sum of fields + For * all hits
    }
    acc.count = cnt;
    block_reduce_store(acc, a.partials + blockIdx.x);
}

// max_log2 <= 32 and every pack's words 16-byte aligned: the LDS-staged form; else the general one
hipError_t launch_packed_scan_lds(hipStream_t s, int grid, const PackedScanArgs &a, uint32_t max_log2) {
    if (grid <= 0) return hipSuccess;
    const uint32_t wave_bytes = 512u * (max_log2 ? max_log2 : 1u) + 16u;   // + the dword a field's second read may touch behind the chunk
    const size_t lds = (size_t)kWavesPerBlock * wave_bytes;
    if (lds > 48 * 1024) {  // 32-bit fields: 64 KiB + 64 B per workgroup
        static const bool raised = [] {
            (void)hipFuncSetAttribute((const void *)packed_scan_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
            (void)hipFuncSetAttribute((const void *)packed_scan_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
            return true;
        }();
        (void)raised;
    }
    if (a.want_sum) hipLaunchKernelGGL(packed_scan_lds_kernel<true>, dim3(grid), dim3(kBlock), lds, s, a, wave_bytes);
    else hipLaunchKernelGGL(packed_scan_lds_kernel<false>, dim3(grid), dim3(kBlock), lds, s, a, wave_bytes);
    return hipGetLastError();
}

hipError_t launch_packed_scan(hipStream_t s, int grid, const PackedScanArgs &a) {
    if (grid <= 0) return hipSuccess;
    if (a.want_sum) KNOX_LAUNCH(packed_scan_kernel<true>, grid, a);
    KNOX_LAUNCH(packed_scan_kernel<false>, grid, a);
}

// ---------------------------------------------------------------------------------------------
// Scan on DICTIONARY containers without materializing them (round 4).  The reference matches on the codes too
// (DictionaryContainer.Match*, internal/encode/int_dict.go:181-380: the value is looked up in the sorted dictionary, the
// compare runs on the uint16 codes).  Here the predicate is evaluated on every dictionary entry of the pack -- a hit bitmap
// of <= 2^16 bits in LDS, built when a workgroup enters a pack; no assumption about the dictionary's order -- and the codes
// stream through it: one LDS bit lookup per row, 2 B + 1/8 B of HBM traffic per row for any value width.
// Work shape: a workgroup owns a contiguous range of 8192-row chunks (so it changes packs rarely); per chunk a wave takes
// 2048 rows = 4 loads of 16 B per lane (8 codes); a lane's 8 result bits are OR-ed with its quad's (DPP) into one bitset
// word per 4 lanes, and lane 4q + j stores load j's word of quad q: 64 consecutive words = 256 B per wave and chunk.
// Sum (wrapping, of the selected VALUES): dict[code] from LDS when the pack's dictionary fits 32 KiB, else through the L2.
// ---------------------------------------------------------------------------------------------
template <int W> __device__ __forceinline__ uint64_t enc_load_val(const void *vals, uint64_t i) {
    if constexpr (W == 8) return ((const uint64_t *)vals)[i];
    else if constexpr (W == 4) return ((const uint32_t *)vals)[i];
    else if constexpr (W == 2) return ((const uint16_t *)vals)[i];
    else return ((const uint8_t *)vals)[i];
}

template <int W, bool SUM>
__global__ void __launch_bounds__(kBlock) dict_scan_kernel(const EncScanArgs a, uint64_t chunks_per_wg) {
    __shared__ uint32_t hitmap[2048];                                   // 2^16 bits
    // the sum of the selected VALUES = sum over codes of (selected rows with that code) x dict[code]: the rows only bump a
    // per-code counter in LDS (ds_add, no return value: a quarter of the rows at 25 % selectivity), the multiplication
    // happens once per dictionary entry when the workgroup leaves the pack.  (First version: dict[code] fetched from an
    // LDS copy per selected row -- 64-bit gathers, 2.9 x the time of the count-only scan.)  Dictionaries beyond
    // kEncHistCodes entries (the reference builds them up to 2^15, hashprobe.MAX_DICT_LIMIT) read dict[code] through the L2.
    __shared__ uint32_t chist[SUM ? kEncHistCodes : 1];
    // the hit codes of a SORTED dictionary under a range predicate are one interval [first, last]: the per-code bitmap lookup
    // (one LDS instruction per row: the count-only scan was LDS-issue bound at 63 % of the HBM peak) becomes a compare.  Found,
    // not assumed: first / last / number of hit entries are collected while the bitmap is built; an unsorted dictionary, a NOT
    // EQUAL (two intervals) or anything else that is not one interval keeps the bitmap.
    __shared__ uint32_t s_first, s_last, s_hits;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t chunks_pp = a.pack_rows / kEncChunkRows;
    const uint64_t n_chunks = (uint64_t)a.n_list * chunks_pp;
    const uint64_t c0 = (uint64_t)blockIdx.x * chunks_per_wg;
    const uint64_t c1 = c0 + chunks_per_wg < n_chunks ? c0 + chunks_per_wg : n_chunks;
    Acc acc;
    acc.init();
    uint64_t cnt = 0, sum = 0;
    uint32_t cur_pack = 0xffffffffu;
    EncPack hp = {0, 0, 0, 0, 0, 0, 0, 0};
    bool use_hist = false, contig = false;
    uint32_t c_lo = 0, c_diff = 0, q_lo = 0, q_diff = 0, q_m1 = 0;
    auto flush_hist = [&]() {   // workgroup uniform; callers synchronise before and after
        if constexpr (SUM) {
            if (use_hist)
                for (uint32_t e = threadIdx.x; e < hp.len; e += kBlock) {
                    const uint32_t k = chist[e];
                    if (k) sum += (uint64_t)k * enc_load_val<W>(a.vals, hp.off + e);
                }
        }
    };
    for (uint64_t c = c0; c < c1; c++) {
        const uint32_t pack = a.list[c / chunks_pp];
        if (pack != cur_pack) {  // workgroup uniform
            __syncthreads();     // everybody is done with the previous pack's tables
            if (cur_pack != 0xffffffffu) flush_hist();
            __syncthreads();
            cur_pack = pack;
            hp = a.packs[pack];
            for (int i = threadIdx.x; i < 2048; i += kBlock) hitmap[i] = 0;
            if (threadIdx.x == 0) { s_first = 0xffffffffu; s_last = 0; s_hits = 0; }
            use_hist = SUM && hp.len <= (uint32_t)kEncHistCodes;
            if constexpr (SUM) {
                if (use_hist)
                    for (uint32_t e = threadIdx.x; e < hp.len; e += kBlock) chist[e] = 0;
            }
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < hp.len; e += kBlock) {
                const uint64_t v = enc_load_val<W>(a.vals, hp.off + e);
                const bool hit = (((v - a.lo) & a.wmask) <= a.diff) != (a.negate != 0);
                if (hit) {
                    atomicOr(&hitmap[e >> 5], 1u << (e & 31));
                    atomicMin(&s_first, e);
                    atomicMax(&s_last, e);
                    atomicAdd(&s_hits, 1u);
                }
            }
            __syncthreads();
            const uint32_t nh = s_hits;
            contig = nh == 0 || s_last - s_first + 1 == nh;
            c_lo = nh ? s_first : 1u;           // no hit at all: the empty interval [1, 0]
            c_diff = nh ? s_last - s_first : 0xffffffffu;
            if (nh == 0) c_diff = 0, c_lo = 0x10000u;  // (code - 0x10000) wraps above every 16-bit code: never <= 0
            // the same interval for narrow_load: the low end in both halves of a dword, the width, "no hit at all" as a constant
            q_lo = (c_lo & 0xffffu) * 0x00010001u;
            q_diff = nh ? s_last - s_first : 0u;
            q_m1 = nh ? 0xffu : 0u;
        }
        const uint64_t in_pack = (c % chunks_pp) * kEncChunkRows + (uint64_t)wave * 2048;  // this wave's first row inside the pack
        if (in_pack >= hp.rows) continue;                                                   // a short last pack (wave uniform)
        const uint64_t row0 = (uint64_t)pack * a.pack_rows + in_pack;
        const u32x4 *src = (const u32x4 *)(a.codes + hp.code_off + in_pack) + lane;
        u32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = __builtin_nontemporal_load(src + u * 64);
        uint32_t mine = 0;
        // one interval of hit codes and no pack end inside the wave's 2048 rows (wave uniform): the codes go through the 2-byte scans'
        // predicate (knox_dev.h narrow_load: v_pk_sub_u16 + an SDWA compare per half word + add-with-carry, 2.5 operations per row
        // where the general form below takes ~8 -- extract, subtract, compare, row bound, code bound, shift, or); a code beyond the
        // dictionary cannot lie inside the interval, whose ends are dictionary entries
        const bool quick = contig && in_pack + 2048 <= hp.rows;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t w[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
            const uint64_t lane_row = in_pack + (uint64_t)u * 512 + (uint64_t)lane * 8;
            uint32_t m = 0;
            if (quick) {
                uint32_t unused = 0;
                m = narrow_load<2, false>(r[u], q_lo, q_diff, q_m1, 0u, unused);
                if constexpr (SUM) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if ((m >> j) & 1u) {
                            const uint32_t code = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                            if (use_hist) atomicAdd(&chist[code], 1u);
                            else sum += enc_load_val<W>(a.vals, hp.off + code);
                        }
                    }
                }
            } else
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t code = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                uint32_t hit;
                if (contig) hit = (code - c_lo) <= c_diff ? 1u : 0u;          // workgroup uniform: hoisted out of the unrolled loops
                else hit = (hitmap[code >> 5] >> (code & 31)) & 1u;
                hit = (lane_row + j < hp.rows && code < hp.len) ? hit : 0u;   // rows past a short pack's end; a code outside the dictionary never matches
                m |= hit << j;
                if constexpr (SUM) {
                    if (hit) {
                        if (use_hist) atomicAdd(&chist[code], 1u);
                        else sum += enc_load_val<W>(a.vals, hp.off + code);
                    }
                }
            }
            const uint32_t word = pack_word<8>(m, lane);   // the quad's 32 rows, in all four of its lanes
            mine = (lane & 3) == u ? word : mine;
        }
        cnt += (uint64_t)__builtin_popcount(mine);
        if (a.bits != nullptr) {
            const uint64_t wi = (row0 >> 5) + (uint64_t)(lane & 3) * 16 + (uint64_t)(lane >> 2);  // load (lane & 3), quad (lane >> 2)
            if (wi * 32 < ((a.n_rows + 63) & ~63ull)) __builtin_nontemporal_store(mine, (uint32_t *)a.bits + wi);
        }
    }
    __syncthreads();
    if (cur_pack != 0xffffffffu) flush_hist();
    acc.count = cnt;
    acc.isum = sum;
    block_reduce_store(acc, a.partials + blockIdx.x);
}

hipError_t launch_dict_scan(hipStream_t s, int grid, const EncScanArgs &a) {
    if (grid <= 0) return hipSuccess;
    const uint64_t n_chunks = (uint64_t)a.n_list * (a.pack_rows / kEncChunkRows);
    const uint64_t per = (n_chunks + (uint64_t)grid - 1) / (uint64_t)grid;
#define KNOX_DICT(W)                                                                       \
    if (a.want_sum) KNOX_LAUNCH((dict_scan_kernel<W, true>), grid, a, per);                \
    KNOX_LAUNCH((dict_scan_kernel<W, false>), grid, a, per)
    switch (a.width) {
    case 8: KNOX_DICT(8);
    case 4: KNOX_DICT(4);
    case 2: KNOX_DICT(2);
    case 1: KNOX_DICT(1);
    default: return hipErrorInvalidValue;
    }
#undef KNOX_DICT
}

// ---------------------------------------------------------------------------------------------
// Scan on RUN-END containers without materializing them (round 4).  The reference compares the run VALUES and translates
// the matching runs into row ranges (RunEndContainer.Match* -> applyMatch, internal/encode/int_runend.go:224-330).  Here:
//   runend_bits_kernel   one lane per 64-bit bitset word: lower_bound of the word's first row in the pack's ends, then the
//                        runs that overlap the word are walked; a run's bit range is set when its value matches.  Reads
//                        (W + 4) B per run (through the L2: neighbouring words share runs), writes 1/8 B per row.
//   runend_sum_kernel    one lane per run: count += length, sum += value * length (wrapping) for matching runs.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) runend_bits_kernel(const EncScanArgs a) {
    const uint64_t words_pp = a.pack_rows >> 6;
    const uint64_t n_words = (uint64_t)a.n_list * words_pp;
    for (uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; li < n_words; li += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t pack = a.list[li / words_pp];
        const uint64_t wi = (uint64_t)pack * words_pp + li % words_pp;   // the word inside the column's bitset
        const EncPack hp = a.packs[pack];
        const uint64_t r0 = (li % words_pp) << 6;       // first row of the word inside its pack
        if (r0 >= hp.rows) {
            if (wi * 64 < ((a.n_rows + 63) & ~63ull)) a.bits[wi] = 0;
            continue;
        }
        const uint32_t *ends = a.ends + hp.code_off;   // a run-end pack's first end (its values: hp.off)
        uint32_t lo = 0, hi = hp.len;                    // first run whose inclusive end is >= r0
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if ((uint64_t)ends[mid] < r0) lo = mid + 1; else hi = mid;
        }
        uint64_t word = 0;
        uint64_t row = r0;
        const uint64_t stop = r0 + 64 < hp.rows ? r0 + 64 : hp.rows;
        for (uint32_t r = lo; r < hp.len && row < stop; r++) {
            const uint64_t end = (uint64_t)ends[r] + 1 < stop ? (uint64_t)ends[r] + 1 : stop;   // exclusive
            const uint64_t v = enc_load_val<W>(a.vals, hp.off + r);
            const bool hit = (((v - a.lo) & a.wmask) <= a.diff) != (a.negate != 0);
            if (hit && end > row) {
                const uint32_t b0 = (uint32_t)(row - r0), n = (uint32_t)(end - row);
                word |= (n >= 64 ? ~0ull : ((1ull << n) - 1)) << b0;
            }
            row = end;
        }
        a.bits[wi] = word;
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) runend_sum_kernel(const EncScanArgs a) {
    Acc acc;
    acc.init();
    uint64_t cnt = 0, sum = 0;
    // one lane per run, packs walked by the workgroups (a pack's runs are few: the container exists for long runs)
    for (uint32_t li = blockIdx.x; li < a.n_list; li += gridDim.x) {
        const EncPack hp = a.packs[a.list[li]];
        const uint32_t *ends = a.ends + hp.code_off;   // a run-end pack's first end (its values: hp.off)
        for (uint32_t r = threadIdx.x; r < hp.len; r += kBlock) {
            const uint64_t first = r == 0 ? 0 : (uint64_t)ends[r - 1] + 1;
            uint64_t last1 = (uint64_t)ends[r] + 1;
            if (last1 > hp.rows) last1 = hp.rows;
            const uint64_t v = enc_load_val<W>(a.vals, hp.off + r);
            const bool hit = (((v - a.lo) & a.wmask) <= a.diff) != (a.negate != 0);
            if (hit && last1 > first) {
                cnt += last1 - first;
                sum += v * (last1 - first);
            }
        }
    }
    acc.count = cnt;
    acc.isum = sum;
    block_reduce_store(acc, a.partials + blockIdx.x);
}

hipError_t launch_runend_scan(hipStream_t s, int grid_bits, int grid_runs, const EncScanArgs &a) {
#define KNOX_RE(W)                                                                                              \
    if (a.bits != nullptr && grid_bits > 0) hipLaunchKernelGGL((runend_bits_kernel<W>), dim3(grid_bits), dim3(kBlock), 0, s, a); \
    hipLaunchKernelGGL((runend_sum_kernel<W>), dim3(grid_runs), dim3(kBlock), 0, s, a);                        \
    break
    switch (a.width) {
    case 8: KNOX_RE(8);
    case 4: KNOX_RE(4);
    case 2: KNOX_RE(2);
    case 1: KNOX_RE(1);
    default: return hipErrorInvalidValue;
    }
#undef KNOX_RE
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The packs of an encoded column that are neither dictionary nor run-end (round 4): constant [1] and delta [2] containers are
// scanned from their header alone -- row i is T(i * delta + base), wrapping in T exactly as ConstContainer / DeltaContainer.Get
// compute it (internal/encode/int_const.go, int_delta.go:109-111), so the scan reads NOTHING and writes the bitset -- and every
// other container kind (raw, bit-packed, simple8, nested ones) was decoded into the value pool at upload and is compared there.
// One lane per row, 64 rows per wave step: the wave64 ballot of the hits IS the bitset word.
// ---------------------------------------------------------------------------------------------
template <int W, bool RAW>
__global__ void __launch_bounds__(kBlock) flat_scan_kernel(const EncScanArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t words_pp = a.pack_rows >> 6;
    const uint64_t n_words = (uint64_t)a.n_list * words_pp;
    const uint64_t wave0 = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * kWavesPerBlock;
    Acc acc;
    acc.init();
    uint64_t cnt = 0, sum = 0;
    for (uint64_t li = wave0; li < n_words; li += n_waves) {
        const uint32_t pack = a.list[li / words_pp];
        const EncPack hp = a.packs[pack];
        const uint64_t r0 = (li % words_pp) << 6;
        const uint64_t wi = (uint64_t)pack * words_pp + (li % words_pp);
        const uint64_t row = r0 + (uint64_t)lane;
        bool hit = false;
        uint64_t v = 0;
        if (row < hp.rows) {
            if constexpr (RAW) v = enc_load_val<W>(a.vals, hp.off + row);
            else v = (row * hp.delta + hp.base) & a.wmask;
            hit = (((v - a.lo) & a.wmask) <= a.diff) != (a.negate != 0);
        }
        const uint64_t word = __ballot(hit);
        if (a.bits != nullptr && lane == 0 && wi * 64 < ((a.n_rows + 63) & ~63ull)) a.bits[wi] = word;
        if (lane == 0) cnt += (uint64_t)__builtin_popcountll(word);
        sum += hit ? v : 0ull;
    }
    acc.count = cnt;
    acc.isum = sum;
    block_reduce_store(acc, a.partials + blockIdx.x);
}

hipError_t launch_flat_scan(hipStream_t s, int grid, bool raw, const EncScanArgs &a) {
    if (grid <= 0) return hipSuccess;
#define KNOX_FLAT(W)                                                       \
    if (raw) KNOX_LAUNCH((flat_scan_kernel<W, true>), grid, a);            \
    KNOX_LAUNCH((flat_scan_kernel<W, false>), grid, a)
    switch (a.width) {
    case 8: KNOX_FLAT(8);
    case 4: KNOX_FLAT(4);
    case 2: KNOX_FLAT(2);
    case 1: KNOX_FLAT(1);
    default: return hipErrorInvalidValue;
    }
#undef KNOX_FLAT
}

// ---------------------------------------------------------------------------------------------
// Simple8 containers (internal/encode/int_s8b.go:87-115, s8b/generic/decode.go:15-80, unpack_for.go):
// every 64-bit word carries a 4-bit selector (bits 60..63) and n values of b bits, LSB first;
// selectors 0 / 1 are runs of 128 zeros / ones.  value = field + minv.  A word's position in the
// output is the prefix sum of the counts before it: count -> exclusive scan -> unpack, one lane per word.
// ---------------------------------------------------------------------------------------------
__device__ __constant__ uint8_t kS8bN[16] = {128, 128, 60, 30, 20, 15, 12, 10, 8, 7, 6, 5, 4, 3, 2, 1};
__device__ __constant__ uint8_t kS8bBits[16] = {0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15, 20, 30, 60};

__global__ void __launch_bounds__(kBlock) s8b_count_kernel(const uint64_t *words, uint64_t n_words, uint32_t *counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) counts[i] = kS8bN[words[i] >> 60];
}

template <int W>
__global__ void __launch_bounds__(kBlock) s8b_unpack_kernel(const uint64_t *words, uint64_t n_words, const uint32_t *offsets, uint64_t minv,
                                                            uint64_t n, void *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const uint64_t v = words[i];
    const uint32_t sel = (uint32_t)(v >> 60);
    const uint32_t cnt = kS8bN[sel], b = kS8bBits[sel];
    const uint64_t mask = b == 60 ? 0xFFFFFFFFFFFFFFFull : ((1ull << b) - 1);
    const uint64_t pos = offsets[i];
    for (uint32_t j = 0; j < cnt && pos + j < n; j++) {
        const uint64_t field = sel < 2 ? (uint64_t)sel : ((v >> (j * b)) & mask);
        const uint64_t x = field + minv;
        if constexpr (W == 8) ((uint64_t *)dst)[pos + j] = x;
        else if constexpr (W == 4) ((uint32_t *)dst)[pos + j] = (uint32_t)x;
        else if constexpr (W == 2) ((uint16_t *)dst)[pos + j] = (uint16_t)x;
        else ((uint8_t *)dst)[pos + j] = (uint8_t)x;
    }
}

// scratch: n_words + 1 uint32 (counts -> offsets, total at [n_words]); n_out: device counter
hipError_t launch_s8b_decode(hipStream_t s, int width, const uint64_t *words, uint64_t n_words, uint64_t minv, uint64_t n, uint32_t *scratch,
                             unsigned long long *n_out, void *dst) {
    if (n_words == 0 || n == 0) return hipSuccess;
    const int g = (int)((n_words + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(s8b_count_kernel, dim3(g), dim3(kBlock), 0, s, words, n_words, scratch);
    { const hipError_t e = launch_idx_scan(s, scratch, n_words, n_out); if (e != hipSuccess) return e; }
    switch (width) {
    case 8: KNOX_LAUNCH(s8b_unpack_kernel<8>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    case 4: KNOX_LAUNCH(s8b_unpack_kernel<4>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    case 2: KNOX_LAUNCH(s8b_unpack_kernel<2>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    case 1: KNOX_LAUNCH(s8b_unpack_kernel<1>, g, words, n_words, (const uint32_t *)scratch, minv, n, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// ALP float containers (internal/encode/float_alp.go:109-165, alp/decoder.go:40-130): value i =
// T(enc[i]) * F10[factor] * IF10[exponent] -- two separate multiplications in T's arithmetic, left
// to right (this file is compiled with -ffp-contract=off) -- then the exceptions are patched in.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) alp_decode_kernel(const void *enc, uint64_t f_bits, uint64_t e_bits, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if constexpr (W == 8) {
            const double f = __longlong_as_double((long long)f_bits), e = __longlong_as_double((long long)e_bits);
            double v = (double)((const int64_t *)enc)[i];
            v = v * f;
            v = v * e;
            ((double *)dst)[i] = v;
        } else {
            const float f = __uint_as_float((uint32_t)f_bits), e = __uint_as_float((uint32_t)e_bits);
            float v = (float)((const int32_t *)enc)[i];
            v = v * f;
            v = v * e;
            ((float *)dst)[i] = v;
        }
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) patch_scatter_kernel(const void *patches, const uint32_t *pos, uint64_t n_patch, uint64_t n, void *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_patch) return;
    const uint64_t p = pos[i];
    if (p >= n) return;
    if constexpr (W == 8) ((uint64_t *)dst)[p] = ((const uint64_t *)patches)[i];
    else ((uint32_t *)dst)[p] = ((const uint32_t *)patches)[i];
}

// ALP-RD (float_alprd.go:80-110, alp/rd.go:200-225): IEEE bits = uint(left) << shift | right
template <int W>
__global__ void __launch_bounds__(kBlock) alprd_merge_kernel(const uint16_t *left, const void *right, uint32_t shift, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if constexpr (W == 8) ((uint64_t *)dst)[i] = ((uint64_t)left[i] << shift) | ((const uint64_t *)right)[i];
        else ((uint32_t *)dst)[i] = ((uint32_t)left[i] << shift) | ((const uint32_t *)right)[i];
    }
}

hipError_t launch_alprd_merge(hipStream_t s, int width, const uint16_t *left, const void *right, uint32_t shift, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (width == 8) KNOX_LAUNCH(alprd_merge_kernel<8>, (int)g, left, right, shift, n, dst);
    if (width == 4) KNOX_LAUNCH(alprd_merge_kernel<4>, (int)g, left, right, shift, n, dst);
    return hipErrorInvalidValue;
}

hipError_t launch_alp_decode(hipStream_t s, int width, const void *enc, uint64_t f_bits, uint64_t e_bits, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (width == 8) KNOX_LAUNCH(alp_decode_kernel<8>, (int)g, enc, f_bits, e_bits, n, dst);
    if (width == 4) KNOX_LAUNCH(alp_decode_kernel<4>, (int)g, enc, f_bits, e_bits, n, dst);
    return hipErrorInvalidValue;
}

hipError_t launch_patch_scatter(hipStream_t s, int width, const void *patches, const uint32_t *pos, uint64_t n_patch, uint64_t n, void *dst) {
    if (n_patch == 0) return hipSuccess;
    const int g = (int)((n_patch + kBlock - 1) / kBlock);
    if (width == 8) KNOX_LAUNCH(patch_scatter_kernel<8>, g, patches, pos, n_patch, n, dst);
    if (width == 4) KNOX_LAUNCH(patch_scatter_kernel<4>, g, patches, pos, n_patch, n, dst);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// String containers -> StringPool rows (ptr = off << 32 | len into the column's payload buffer):
// compact (string_compact.go:64-105) row i = buf[ofs[i] : ofs[i] + len[i]], dictionary
// (string_dict.go:70-120) row i = dict[ofs[code[i]] : + len[code[i]]]; `base` = where this container's
// payload starts inside the column's buffer.  ptr_check flags any row that leaves the buffer.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) str_ptr_kernel(const uint32_t *ofs, const uint32_t *len, uint64_t n_entries, const uint32_t *code,
                                                         uint64_t n, uint64_t base, uint64_t *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = code ? (uint64_t)code[i] : i;
        if (k >= n_entries) k = n_entries - 1;  // corrupt code: stay inside the entry table (ptr_check sees the rest)
        uint64_t o = base + ofs[k];
        if (o > 0xffffffffull) o = 0xffffffffull;  // corrupt offset: keep it out of range instead of wrapping (ptr_check flags it)
        dst[i] = (o << 32) | (uint64_t)len[k];
    }
}

__global__ void __launch_bounds__(kBlock) ptr_check_kernel(const uint64_t *ptr, uint64_t n, uint64_t pool_len, uint32_t *flag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = ptr[i];
        if ((p >> 32) + (p & 0xffffffffull) > pool_len) atomicOr(flag, 1u);
    }
}

hipError_t launch_str_ptr(hipStream_t s, const uint32_t *ofs, const uint32_t *len, uint64_t n_entries, const uint32_t *code, uint64_t n,
                          uint64_t base, uint64_t *dst) {
    if (n == 0) return hipSuccess;
    if (n_entries == 0) return hipErrorInvalidValue;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    KNOX_LAUNCH(str_ptr_kernel, (int)g, ofs, len, n_entries, code, n, base, dst);
}

hipError_t launch_ptr_check(hipStream_t s, const uint64_t *ptr, uint64_t n, uint64_t pool_len, uint32_t *flag) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    KNOX_LAUNCH(ptr_check_kernel, (int)g, ptr, n, pool_len, flag);
}

// ---------------------------------------------------------------------------------------------
// Dictionary and run-end containers (internal/encode/int_dict.go:76-103, int_runend.go:84-120 and the
// float twins float_dict.go:75-102, float_runend.go:80-110), children already decoded on the device:
//   dictionary  value i = dict[codes[i]]           (codes: uint16)
//   run-end     value i = values[k], k = first run with ends[k] >= i   (ends: uint32, inclusive)
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) dict_gather_kernel(const void *dict, uint64_t n_dict, const uint16_t *codes, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c = codes[i];
        if (c >= n_dict) c = n_dict - 1;  // corrupt input: stay inside the dictionary
        const auto v = load_elem<W>(dict, c);
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) runend_expand_kernel(const void *values, const uint32_t *ends, uint64_t n_runs, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n_runs;  // first k with ends[k] >= i (sort.Search)
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)ends[mid] >= i) hi = mid; else lo = mid + 1;
        }
        if (lo >= n_runs) lo = n_runs - 1;
        const auto v = load_elem<W>(values, lo);
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

hipError_t launch_dict_gather(hipStream_t s, int width, const void *dict, uint64_t n_dict, const uint16_t *codes, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    if (n_dict == 0) return hipErrorInvalidValue;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    switch (width) {
    case 8: KNOX_LAUNCH(dict_gather_kernel<8>, (int)g, dict, n_dict, codes, n, dst);
    case 4: KNOX_LAUNCH(dict_gather_kernel<4>, (int)g, dict, n_dict, codes, n, dst);
    case 2: KNOX_LAUNCH(dict_gather_kernel<2>, (int)g, dict, n_dict, codes, n, dst);
    case 1: KNOX_LAUNCH(dict_gather_kernel<1>, (int)g, dict, n_dict, codes, n, dst);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_runend_expand(hipStream_t s, int width, const void *values, const uint32_t *ends, uint64_t n_runs, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    if (n_runs == 0) return hipErrorInvalidValue;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    switch (width) {
    case 8: KNOX_LAUNCH(runend_expand_kernel<8>, (int)g, values, ends, n_runs, n, dst);
    case 4: KNOX_LAUNCH(runend_expand_kernel<4>, (int)g, values, ends, n_runs, n, dst);
    case 2: KNOX_LAUNCH(runend_expand_kernel<2>, (int)g, values, ends, n_runs, n, dst);
    case 1: KNOX_LAUNCH(runend_expand_kernel<1>, (int)g, values, ends, n_runs, n, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// Constant and delta containers (internal/encode/int_const.go:67-88, int_delta.go:70-95, float_const.go:67-86):
// value i = T(i) * delta + base, wrapping in T (delta = 0: constant; floats: base = the IEEE bits).
// Nothing but three varints crosses PCIe; the column is written at HBM rate.
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(kBlock) affine_fill_kernel(uint64_t base, uint64_t delta, uint64_t n, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = i * delta + base;  // low W bytes = T(i) * delta + base in T's arithmetic
        if constexpr (W == 8) ((uint64_t *)dst)[i] = v;
        else if constexpr (W == 4) ((uint32_t *)dst)[i] = (uint32_t)v;
        else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)v;
        else ((uint8_t *)dst)[i] = (uint8_t)v;
    }
}

hipError_t launch_affine_fill(hipStream_t s, int width, uint64_t base, uint64_t delta, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    switch (width) {
    case 8: KNOX_LAUNCH(affine_fill_kernel<8>, (int)g, base, delta, n, dst);
    case 4: KNOX_LAUNCH(affine_fill_kernel<4>, (int)g, base, delta, n, dst);
    case 2: KNOX_LAUNCH(affine_fill_kernel<2>, (int)g, base, delta, n, dst);
    case 1: KNOX_LAUNCH(affine_fill_kernel<1>, (int)g, base, delta, n, dst);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic columns (DESIGN.md "synthetic data"): r(i) = mix64(i + mix64(seed))
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t gen_u64(int kind, uint64_t seedmix, uint64_t i, int64_t lo, uint64_t span) {
    const uint64_t r = mix64(i + seedmix);
    switch (kind) {
    case 0: return span ? (uint64_t)lo + __umul64hi(r, span) : r;
    case 1: return (uint64_t)__double_as_longlong((double)(r >> 44) * 0x1p-10);
    case 2: return (uint64_t)__double_as_longlong((double)(r >> 11) * 0x1p-53);
    case 3: return i + (uint64_t)lo;
    case 4: {
        if ((r & 0xfffff) == 0) {
            const uint64_t sel = (r >> 20) % 5;
            return sel == 0 ? 0x7ff8000000000000ull : sel == 1 ? 0x7ff0000000000000ull : sel == 2 ? 0xfff0000000000000ull
                 : sel == 3 ? 0x8000000000000000ull : 0ull;
        }
        const double d = ((double)(int64_t)(r >> 11) * 0x1p-53 - 0.5) * 0x1p21;
        return (uint64_t)__double_as_longlong(d);
    }
    default: return 0;
    }
}

template <int W>
__global__ void __launch_bounds__(kBlock) generate_kernel(int to_f32, int kind, uint64_t seedmix, uint64_t first_row,
                                                          uint64_t n_rows, int64_t lo, uint64_t span, void *dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t u = gen_u64(kind, seedmix, first_row + i, lo, span);
        if constexpr (W == 8) ((uint64_t *)dst)[i] = u;
        else if constexpr (W == 4) {
            if (to_f32) {
                const float f = (kind == 1 || kind == 2 || kind == 4) ? (float)__longlong_as_double((long long)u) : (float)(int64_t)u;
                ((float *)dst)[i] = f;
            } else ((uint32_t *)dst)[i] = (uint32_t)u;
        } else if constexpr (W == 2) ((uint16_t *)dst)[i] = (uint16_t)u;
        else ((uint8_t *)dst)[i] = (uint8_t)u;
    }
}

hipError_t launch_generate(hipStream_t s, int width, bool to_f32, int kind, uint64_t seed, uint64_t first_row, uint64_t n_rows,
                           int64_t lo, uint64_t span, void *dst) {
    if (n_rows == 0) return hipSuccess;
    uint64_t g = (n_rows + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    const uint64_t sm = mix64(seed);
    switch (width) {
    case 8: KNOX_LAUNCH((generate_kernel<8>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    case 4: KNOX_LAUNCH((generate_kernel<4>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    case 2: KNOX_LAUNCH((generate_kernel<2>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    case 1: KNOX_LAUNCH((generate_kernel<1>), (int)g, to_f32 ? 1 : 0, kind, sm, first_row, n_rows, lo, span, dst);
    default: return hipErrorInvalidValue;
    }
}

// synthetic bytes column (knox_kernels.h launch_generate_bytes; DESIGN.md "synthetic data" kind 5): one lane per row
__global__ void __launch_bounds__(kBlock) generate_bytes_kernel(uint64_t seedmix, uint64_t keymix, uint64_t first_row, uint64_t n_rows,
                                                                uint64_t pack_rows, uint32_t key_len, uint32_t window_bits,
                                                                uint32_t packs_per_window, uint64_t pool_stride, uint64_t *ptr, uint8_t *buf) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = first_row + i;
        const uint64_t pack = row / pack_rows, in = row - pack * pack_rows;
        const uint64_t k = ((pack / packs_per_window) << window_bits) | (mix64(row + seedmix) & ((1ull << window_bits) - 1ull));
        uint8_t *dst = buf + (i / pack_rows) * pool_stride + in * key_len;
        for (uint32_t j = 0; j * 8u < key_len; j++) {
            const uint64_t w = mix64(8ull * k + j + keymix);
            const uint32_t nb = key_len - j * 8u < 8u ? key_len - j * 8u : 8u;
            if ((key_len & 7u) == 0u) { ((uint64_t *)dst)[j] = w; continue; }  // pools are 64-byte aligned: whole words
            for (uint32_t b = 0; b < nb; b++) dst[j * 8u + b] = (uint8_t)(w >> (8u * b));
        }
        ptr[i] = ((in * key_len) << 32) | key_len;
    }
}

hipError_t launch_generate_bytes(hipStream_t s, uint64_t seed, uint64_t first_row, uint64_t n_rows, uint64_t pack_rows, uint32_t key_len,
                                 uint32_t window_bits, uint32_t packs_per_window, uint64_t pool_stride, uint64_t *ptr, uint8_t *buf) {
    if (n_rows == 0) return hipSuccess;
    if (pack_rows == 0 || first_row % pack_rows || packs_per_window == 0 || window_bits > 40) return hipErrorInvalidValue;
    uint64_t g = (n_rows + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    KNOX_LAUNCH(generate_bytes_kernel, (int)g, mix64(seed), mix64(seed ^ 0x6b657973ull), first_row, n_rows, pack_rows, key_len, window_bits,
                packs_per_window, pool_stride, ptr, buf);
}


}  // namespace knox
