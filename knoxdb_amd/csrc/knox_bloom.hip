// knox_bloom.hip -- bloom filters per pack: probe (bloom.go:136-150), build for numeric and bytes columns
// (bloom.go:162-169, pack/stats/filter.go:296-360), fixed-width and full XXH3-64 on the device.
#include "knox_bytes_dev.h"
#define KNOX_HD __host__ __device__
#define KNOX_XXH3_CONST __constant__
#include "knox_xxh3.h"

namespace knox {

// ---------------------------------------------------------------------------------------------
// bloom filter: internal/filter/bloom/bloom.go:136-150 (probe), :162-169 (add), xxh3.go:22-33
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rol64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xxh3_u64(uint64_t val) {
    const uint64_t input64 = (val >> 32) + (val << 32);
    uint64_t h = input64 ^ (0x1cad21f72c81017cull ^ 0xdb979083e96dd4deull);
    h ^= rol64(h, 49) ^ rol64(h, 24);
    h *= 0x9fb21c651e98df25ull;
    h ^= (h >> 35) + 8;
    h *= 0x9fb21c651e98df25ull;
    h ^= (h >> 28);
    return h;
}

__global__ void __launch_bounds__(kBlock) bloom_probe_kernel(const uint8_t *filters, const uint64_t *offsets, uint64_t n_packs,
                                                             uint64_t h, uint8_t *maybe_out) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_packs) return;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) {  // no filter for this pack: cannot exclude it
        maybe_out[p] = 1;
        return;
    }
    const uint8_t *buf = filters + o0;
    const uint8_t *bits = buf + 1;
    const uint32_t k = buf[0];
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    uint32_t h0 = (uint32_t)h;
    const uint32_t h1 = (uint32_t)(h >> 32);
    uint8_t ok = 1;
    for (uint32_t i = 0; i < k; i++) {
        if ((bits[(h0 & mask) >> 3] & (1u << (h0 & 7))) == 0) { ok = 0; break; }
        h0 += h1;
    }
    maybe_out[p] = ok;
}

hipError_t launch_bloom_probe(hipStream_t s, const uint8_t *filters, const uint64_t *offsets, uint64_t n_packs, uint64_t h,
                              uint8_t *maybe_out) {
    if (n_packs == 0) return hipSuccess;
    KNOX_LAUNCH(bloom_probe_kernel, (int)((n_packs + kBlock - 1) / kBlock), filters, offsets, n_packs, h, maybe_out);
}

// one workgroup per pack; filters pre-zeroed by the host; k byte = 4 (bloom.go:48-60)
// fixed-width value hashes (internal/hash/xxh3.go:35-58, hash.go:67-92): u32 = the 4..8 byte XXH3 branch,
// u16 / u8 = the 1..3 byte branch; floats hash their IEEE bits like the same-width integer
__device__ __forceinline__ uint64_t xxh3_rrmxmx(uint64_t h, uint64_t len) {
    h ^= rol64(h, 49) ^ rol64(h, 24);
    h *= 0x9fb21c651e98df25ull;
    h ^= (h >> 35) + len;
    h *= 0x9fb21c651e98df25ull;
    return h ^ (h >> 28);
}
__device__ __forceinline__ uint64_t xxh3_avalanche_small(uint64_t x) {
    x ^= x >> 33;
    x *= 14029467366897019727ull;
    x ^= x >> 29;
    x *= 1609587929392839161ull;
    return x ^ (x >> 32);
}
template <int W> __device__ __forceinline__ uint64_t xxh3_value(uint64_t v) {
    if constexpr (W == 8) return xxh3_u64(v);
    else if constexpr (W == 4) return xxh3_rrmxmx((v + (v << 32)) ^ (0x1cad21f72c81017cull ^ 0xdb979083e96dd4deull), 4);
    else if constexpr (W == 2) return xxh3_avalanche_small((((v * ((1ull << 24) + 1)) >> 8) + (2ull << 8)) ^ (0xbe4ba423ull ^ 0x396cfeb8ull));
    else return xxh3_avalanche_small(((v * ((1ull << 24) + (1ull << 16) + 1)) + (1ull << 8)) ^ (0xbe4ba423ull ^ 0x396cfeb8ull));
}

constexpr int kBloomSplit = 256;
template <int W>
__global__ void __launch_bounds__(kBlock) bloom_build_kernel(const void *col, uint64_t n_rows, uint64_t pack_rows, uint64_t n_packs,
                                                             uint8_t *filters, const uint64_t *offsets) {
    // kBloomSplit workgroups per pack (the bits are set with atomics, so they may share a filter), and XCD-aware: workgroups
    // are dealt round-robin over the 8 XCDs, so workgroup b works on pack 8*(b / (8*S)) + b % 8 -- each XCD sets the bits of
    // ONE pack's filter at a time, which then stays in that XCD's L2 (2 MiB of 4 MiB) instead of every atomic going to
    // HBM (one workgroup per pack, 2^30 rows: 0.22 s)
    const uint64_t p = (blockIdx.x / (8 * kBloomSplit)) * 8 + (blockIdx.x & 7), part = (blockIdx.x >> 3) % kBloomSplit;
    if (p >= n_packs) return;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    uint8_t *buf = filters + o0;
    if (threadIdx.x == 0 && part == 0) buf[0] = 4;
    const uint64_t bits_off = o0 + 1;  // byte offset of bit array inside `filters`
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    uint64_t r0 = p * pack_rows, r1 = r0 + pack_rows;
    if (r1 > n_rows) r1 = n_rows;
    for (uint64_t r = r0 + part * blockDim.x + threadIdx.x; r < r1; r += (uint64_t)blockDim.x * kBloomSplit) {
        const uint64_t h = xxh3_value<W>((uint64_t)load_elem<W>(col, r));
        uint32_t h0 = (uint32_t)h;
        const uint32_t h1 = (uint32_t)(h >> 32);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t byte_addr = bits_off + ((h0 & mask) >> 3);
            const uint32_t bit = 1u << (h0 & 7);
            // byte-granular OR through the aligned 32-bit word that contains the byte
            uint32_t *word = (uint32_t *)(filters + (byte_addr & ~3ull));
            atomicOr(word, bit << (8 * (byte_addr & 3)));
            h0 += h1;
        }
    }
}

// bytes column (StringPool): hash.Hash = XXH3-64 of every row's bytes (pack/stats/filter.go:352-354), one lane per row
__global__ void __launch_bounds__(kBlock) bloom_build_bytes_kernel(BytesView v, uint64_t n_packs, uint8_t *filters, const uint64_t *offsets) {
    const uint64_t p = (blockIdx.x / (8 * kBloomSplit)) * 8 + (blockIdx.x & 7), part = (blockIdx.x >> 3) % kBloomSplit;
    if (p >= n_packs) return;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    if (threadIdx.x == 0 && part == 0) filters[o0] = 4;
    const uint64_t bits_off = o0 + 1;
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    uint64_t r0 = p * v.pack_rows, r1 = r0 + v.pack_rows;
    if (r1 > v.n_rows) r1 = v.n_rows;
    const uint8_t *pool = v.pool_base ? v.buf + v.pool_base[p] : v.buf;
    for (uint64_t r = r0 + part * blockDim.x + threadIdx.x; r < r1; r += (uint64_t)blockDim.x * kBloomSplit) {
        const uint64_t pr = v.ptr[r];
        const uint64_t h = knox_xxh3::hash(pool + (uint32_t)(pr >> 32), (uint64_t)(uint32_t)pr);
        uint32_t h0 = (uint32_t)h;
        const uint32_t h1 = (uint32_t)(h >> 32);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t byte_addr = bits_off + ((h0 & mask) >> 3);
            uint32_t *word = (uint32_t *)(filters + (byte_addr & ~3ull));
            atomicOr(word, (1u << (h0 & 7)) << (8 * (byte_addr & 3)));
            h0 += h1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Partitioned build (the product path for packs of >= 4096 rows and filters of <= 2 MiB): no global atomics per row.
// The atomic kernels above are bound by device-scope atomicOr into a 2 MiB filter (4 per row: 2^30 rows 0.16 s =
// 54 GB/s of column data).  Here every probe location is binned by FILTER SEGMENT (2^16 bits = 8 KiB), each segment's
// bits are OR-ed into an LDS bitmap and stored once, coalesced.  Per batch of packs:
//   bloom_count_kernel  one workgroup per tile of 4096 rows: hash every row, LDS histogram of the 4 locations
//                       (h0 + i*h1) & mask by segment (loc >> 16), added to the pack's 256 segment counts (<= 256 global
//                       atomics per tile).  8 B read per row
//   bloom_scan_kernel   one wave per pack: exclusive prefix of the 256 counts = where each segment's entries start
//   bloom_bin_kernel    the same tiles again: counting sort of the tile's 16384 locations by segment in LDS, one range
//                       reserved per (tile, segment) behind the pack's segment cursor, 16-bit in-segment offsets written as
//                       contiguous runs.  8 B read + 8 B written per row; the entries of one segment end up contiguous,
//                       whatever the data looks like (exact counts: a column of one repeated value works like any other)
//   bloom_fill_kernel   one workgroup per (pack, 8 consecutive segments = 64 KiB of filter): stream each segment's entries,
//                       ds_or them into the LDS bitmap, write the 64 KiB out (the bit array starts at byte offset 1 of the
//                       filter: aligned dwords via v_alignbyte, head / tail bytes one by one).  8 B read + 2 B written per row
// Same bytes as the reference's addUnroll4 (bloom.go:162-169): OR is order independent.  tests/test_gpu_misc.py and
// tests/test_gpu_fullsize.py compare the filters byte for byte with the oracle's.
// ---------------------------------------------------------------------------------------------
constexpr int kBinTileRows = 4096;                     // rows per tile: 16 per thread
constexpr int kBinRowsPerThread = kBinTileRows / kBlock;
constexpr int kBinSlots = kBinTileRows * 4;            // locations per tile
constexpr int kSegShift = 16;                          // 2^16 bits per segment: in-segment offsets fit 16 bits
constexpr int kFillSegs = 8;                           // segments per fill workgroup (64 KiB of LDS)

struct HashNumeric {
    static constexpr bool kIsBytes = false;
    const void *col;
    template <int W> __device__ __forceinline__ uint64_t at(uint64_t row, uint64_t) const { return xxh3_value<W>((uint64_t)load_elem<W>(col, row)); }
};
struct HashBytes {
    static constexpr bool kIsBytes = true;
    const uint64_t *ptr;
    const uint8_t *buf;
    const uint64_t *pool_base;
    template <int W> __device__ __forceinline__ uint64_t at(uint64_t row, uint64_t pack) const {
        const uint8_t *pool = pool_base ? buf + pool_base[pack] : buf;
        const uint64_t pr = ptr[row];
        return knox_xxh3::hash(pool + (uint32_t)(pr >> 32), (uint64_t)(uint32_t)pr);
    }
};

// the tile's row hashes into registers, branch free: a row beyond the pack's end re-hashes the pack's last row (its
// locations are dropped by the `valid` bit at the LDS operations) -- predicating the hash itself makes the compiler keep
// the whole array in flux across 16 branches (4000 register moves per tile in the first version of these kernels)
template <int W, typename H>
__device__ __forceinline__ uint32_t bloom_tile_hashes(const H &hasher, uint64_t p, uint64_t pack_rows, uint64_t n_rows, uint32_t tile,
                                                      uint64_t (&h)[kBinRowsPerThread]) {
    uint64_t r1 = (p + 1) * pack_rows;
    if (r1 > n_rows) r1 = n_rows;
    const uint64_t r0 = p * pack_rows + (uint64_t)tile * kBinTileRows + threadIdx.x;
    const uint64_t left = r1 > r0 ? r1 - r0 : 0;              // rows of this thread: i * kBlock < left
    uint32_t valid = 0;
    if constexpr (H::kIsBytes) {
        // bytes rows: knox_xxh3::hash is a CALL per row (it branches on the length eight ways), and every call waits for its own
        // loads -- sixteen dependent round trips to memory per thread and tile (the bytes build: 71 % of the wave cycles waiting,
        // profiles/r04_bloom_build.txt).  When no row of the wave's tile is longer than 32 bytes (hashes, addresses, UUIDs, names,
        // tags: the branches of XXH3 that need at most four 8-byte words) the rows' words are loaded eight rows at a time before any
        // arithmetic; the pools are allocated with 256 bytes of slack (knox_abi.cpp), so the first word of a short row may be read whole.
        uint64_t pr[kBinRowsPerThread];
        bool fast = true, mid = true;   // every row <= 32 bytes / every row 33 ... 64 bytes
#pragma unroll
        for (int i = 0; i < kBinRowsPerThread; i++) {
            const bool ok = (uint64_t)i * kBlock < left;
            pr[i] = hasher.ptr[ok ? r0 + (uint64_t)i * kBlock : r1 - 1];
            valid |= ok ? 1u << i : 0u;
            fast = fast && (uint32_t)pr[i] <= 32u;
            mid = mid && ((uint32_t)pr[i] - 33u) <= 31u;
        }
        if (__all(mid)) {  // wave uniform: URLs, paths, long identifiers -- eight words per row, four rows loaded ahead
            const uint8_t *pool = hasher.pool_base ? hasher.buf + hasher.pool_base[p] : hasher.buf;
            static_assert(kBinRowsPerThread % 4 == 0, "quarters of four rows");
#pragma unroll
            for (int i0 = 0; i0 < kBinRowsPerThread; i0 += 4) {
                uint64_t w[4][8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint8_t *in = pool + (uint32_t)(pr[i0 + i] >> 32);
                    const uint32_t len = (uint32_t)pr[i0 + i];
                    w[i][0] = knox_xxh3::rd64(in);
                    w[i][1] = knox_xxh3::rd64(in + 8);
                    w[i][2] = knox_xxh3::rd64(in + 16);
                    w[i][3] = knox_xxh3::rd64(in + 24);
                    w[i][4] = knox_xxh3::rd64(in + len - 32);
                    w[i][5] = knox_xxh3::rd64(in + len - 24);
                    w[i][6] = knox_xxh3::rd64(in + len - 16);
                    w[i][7] = knox_xxh3::rd64(in + len - 8);
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
                    h[i0 + i] = knox_xxh3::hash_33_64_words(w[i][0], w[i][1], w[i][2], w[i][3], w[i][4], w[i][5], w[i][6], w[i][7], (uint32_t)pr[i0 + i]);
            }
            return valid;
        }
        if (__all(fast)) {  // wave uniform
            const uint8_t *pool = hasher.pool_base ? hasher.buf + hasher.pool_base[p] : hasher.buf;
            static_assert(kBinRowsPerThread % 8 == 0, "halves of eight rows");
#pragma unroll
            for (int i0 = 0; i0 < kBinRowsPerThread; i0 += 8) {
                uint64_t w[8][4];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint8_t *in = pool + (uint32_t)(pr[i0 + i] >> 32);
                    const uint32_t len = (uint32_t)pr[i0 + i];
                    w[i][0] = knox_xxh3::rd64(in);
                    w[i][3] = knox_xxh3::rd64(in + (len > 8u ? len - 8u : 0u));
                    w[i][1] = 0;
                    w[i][2] = 0;
                    if (len > 16u) {
                        w[i][1] = knox_xxh3::rd64(in + 8);
                        w[i][2] = knox_xxh3::rd64(in + len - 16);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) h[i0 + i] = knox_xxh3::hash_le32_words(w[i][0], w[i][1], w[i][2], w[i][3], (uint32_t)pr[i0 + i]);
            }
            return valid;
        }
        valid = 0;
    }
#pragma unroll
    for (int i = 0; i < kBinRowsPerThread; i++) {
        const bool ok = (uint64_t)i * kBlock < left;
        h[i] = hasher.template at<W>(ok ? r0 + (uint64_t)i * kBlock : r1 - 1, p);
        valid |= ok ? 1u << i : 0u;
    }
    return valid;
}

template <int W, typename H>
__global__ void __launch_bounds__(kBlock) bloom_count_kernel(H hasher, uint64_t n_rows, uint64_t pack_rows, uint64_t p0, uint32_t tiles_pp,
                                                             const uint64_t *offsets, uint32_t *seg_counts) {
    __shared__ uint32_t hist[256];
    const uint32_t tile = blockIdx.x % tiles_pp;
    const uint64_t pl = blockIdx.x / tiles_pp, p = p0 + pl;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    hist[threadIdx.x] = 0;
    __syncthreads();
    uint64_t h[kBinRowsPerThread];
    const uint32_t valid = bloom_tile_hashes<W>(hasher, p, pack_rows, n_rows, tile, h);
#pragma unroll
    for (int i = 0; i < kBinRowsPerThread; i++) {
        if (valid & (1u << i)) {
            uint32_t h0 = (uint32_t)h[i];
            const uint32_t h1 = (uint32_t)(h[i] >> 32);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&hist[(h0 & mask) >> kSegShift], 1u);
                h0 += h1;
            }
        }
    }
    __syncthreads();
    const uint32_t c = hist[threadIdx.x];
    if (c) atomicAdd(&seg_counts[pl * 256 + threadIdx.x], c);
}

// one wave per pack: seg_base[pl][0..256] = exclusive prefix of the counts; the counts become the bin kernel's cursors (= the bases)
__global__ void __launch_bounds__(kBlock) bloom_scan_kernel(uint64_t np, uint32_t *seg_counts, uint32_t *seg_base) {
    const int lane = threadIdx.x & 63;
    const uint64_t pl = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (pl >= np) return;
    uint32_t carry = 0;
    for (int b = 0; b < 4; b++) {
        const uint32_t c = seg_counts[pl * 256 + b * 64 + lane];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const uint32_t excl = carry + incl - c;
        seg_base[pl * 257 + b * 64 + lane] = excl;
        seg_counts[pl * 256 + b * 64 + lane] = excl;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) seg_base[pl * 257 + 256] = carry;
}

template <int W, typename H>
__global__ void __launch_bounds__(kBlock) bloom_bin_kernel(H hasher, uint64_t n_rows, uint64_t pack_rows, uint64_t p0, uint32_t tiles_pp,
                                                           const uint64_t *offsets, uint32_t *seg_cursor, uint16_t *entries, uint64_t entries_pp) {
    __shared__ uint32_t hist[256];    // counts, then the bin's LDS cursor
    __shared__ uint32_t lstart[256];  // the bin's start inside the tile's sorted run
    __shared__ uint32_t gpos[256];    // where the tile's run of this bin goes inside the pack's entries
    __shared__ uint32_t wsum[kBlock / 64];
    __shared__ uint16_t sorted[kBinSlots];
    __shared__ uint8_t segof[kBinSlots];
    const uint32_t tile = blockIdx.x % tiles_pp;
    const uint64_t pl = blockIdx.x / tiles_pp, p = p0 + pl;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    const uint32_t mask = (uint32_t)((o1 - o0 - 1) * 8 - 1);
    hist[threadIdx.x] = 0;
    __syncthreads();
    uint64_t h[kBinRowsPerThread];
    const uint32_t valid = bloom_tile_hashes<W>(hasher, p, pack_rows, n_rows, tile, h);
#pragma unroll
    for (int i = 0; i < kBinRowsPerThread; i++) {
        if (valid & (1u << i)) {
            uint32_t h0 = (uint32_t)h[i];
            const uint32_t h1 = (uint32_t)(h[i] >> 32);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&hist[(h0 & mask) >> kSegShift], 1u);
                h0 += h1;
            }
        }
    }
    __syncthreads();
    // exclusive scan of the 256 bins (thread t owns bin t) + the bin's range behind the pack's cursor
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = hist[threadIdx.x];
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    const uint32_t excl = base + incl - c;
    lstart[threadIdx.x] = excl;
    gpos[threadIdx.x] = c ? atomicAdd(&seg_cursor[pl * 256 + threadIdx.x], c) : 0u;
    __syncthreads();
    hist[threadIdx.x] = excl;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kBinRowsPerThread; i++) {
        if (valid & (1u << i)) {
            uint32_t h0 = (uint32_t)h[i];
            const uint32_t h1 = (uint32_t)(h[i] >> 32);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t loc = h0 & mask;
                const uint32_t pos = atomicAdd(&hist[loc >> kSegShift], 1u);
                sorted[pos] = (uint16_t)loc;
                segof[pos] = (uint8_t)(loc >> kSegShift);
                h0 += h1;
            }
        }
    }
    __syncthreads();
    // consecutive lanes write consecutive entries of a bin's run: a few destination runs per wave store
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint16_t *dst = entries + pl * entries_pp;
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
        const uint32_t sg = segof[i];
        dst[gpos[sg] + (i - lstart[sg])] = sorted[i];
    }
}

__global__ void __launch_bounds__(kBlock) bloom_fill_kernel(uint64_t p0, uint32_t groups_pp, uint8_t *filters, const uint64_t *offsets,
                                                            const uint16_t *entries, uint64_t entries_pp, const uint32_t *seg_base) {
    __shared__ uint32_t bm[kFillSegs * 2048 + 4];
    const uint32_t g = blockIdx.x % groups_pp;
    const uint64_t pl = blockIdx.x / groups_pp, p = p0 + pl;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    const uint64_t fbytes = o1 - o0 - 1;                      // m / 8
    const uint32_t n_seg = (uint32_t)((fbytes * 8 + (1u << kSegShift) - 1) >> kSegShift);
    const uint32_t seg0 = g * kFillSegs;
    if (seg0 >= n_seg) return;
    if (g == 0 && threadIdx.x == 0) filters[o0] = 4;          // k (bloom.go:48-60)
    for (int i = threadIdx.x; i < kFillSegs * 2048 + 4; i += kBlock) bm[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint16_t *pe = entries + pl * entries_pp;
    const uint32_t *pb = seg_base + pl * 257;
    for (uint32_t sl = wave; sl < kFillSegs && seg0 + sl < n_seg; sl += kBlock / 64) {
        uint32_t *seg_bm = bm + sl * 2048;
        const uint32_t e0 = pb[seg0 + sl], e1 = pb[seg0 + sl + 1];
        uint32_t i = e0 + lane;
        for (; i + 7 * 64 < e1; i += 8 * 64) {               // 8 independent loads in flight per lane
            uint32_t off[8];
#pragma unroll
            for (int u = 0; u < 8; u++) off[u] = pe[i + u * 64];
#pragma unroll
            for (int u = 0; u < 8; u++) atomicOr(&seg_bm[off[u] >> 5], 1u << (off[u] & 31));
        }
        for (; i < e1; i += 64) {
            const uint32_t off = pe[i];
            atomicOr(&seg_bm[off >> 5], 1u << (off & 31));
        }
    }
    __syncthreads();
    // write out: the bit array starts at filters + o0 + 1 (any byte alignment)
    uint64_t nbytes = fbytes - (uint64_t)seg0 * 8192;
    if (nbytes > (uint64_t)kFillSegs * 8192) nbytes = (uint64_t)kFillSegs * 8192;
    uint8_t *dst = filters + o0 + 1 + (uint64_t)seg0 * 8192;
    const uint8_t *bmb = (const uint8_t *)bm;
    uint32_t head = (uint32_t)((4 - ((uintptr_t)dst & 3)) & 3);
    if (head > nbytes) head = (uint32_t)nbytes;
    const uint32_t nd = (uint32_t)((nbytes - head) >> 2);     // aligned dwords
    const uint32_t tail0 = head + nd * 4;
    if (threadIdx.x < head) dst[threadIdx.x] = bmb[threadIdx.x];
    if (threadIdx.x >= 8 && threadIdx.x - 8 < nbytes - tail0) dst[tail0 + threadIdx.x - 8] = bmb[tail0 + threadIdx.x - 8];
    uint32_t *dw = (uint32_t *)(dst + head);
    for (uint32_t j = threadIdx.x; j < nd; j += kBlock) {
        const uint32_t b = head + 4 * j;                      // byte offset inside the bitmap
        const uint32_t w0 = bm[b >> 2], w1 = bm[(b >> 2) + 1];
        dw[j] = __builtin_amdgcn_alignbyte(w1, w0, b & 3);
    }
}

// ---------------------------------------------------------------------------------------------
// Round 4: the build reads the column ONCE.  The count / scan passes above exist to give every (pack, segment) its exact
// place in the entries; what they buy is exactness under any distribution, what they cost is a second read (and, for bytes
// columns, a second XXH3) of every row.  Hashed locations are uniform, so a segment's entry count is known in advance up
// to sqrt noise: every (pack, segment) gets a FIXED region of
//     cap = mean + mean / 8 + 256 entries,      mean = 4 * pack_rows / segments   (2^20-row packs: 16384 -> 18688; 16 sigma)
// and tiles reserve ranges in it with one atomicAdd per (tile, segment) on the segment's cursor -- no count pass, no scan.
//   bloom_bin1_kernel   per tile of 4096 rows: hash (once), ONE LDS pass (atomicAdd with return: the entry's rank inside its
//                       (tile, segment) run -- the count and the position pass of bloom_bin_kernel in one), scan of the 256
//                       run lengths, one global atomicAdd per non-empty run, reorder through LDS (one 32-bit word per
//                       entry: segment | offset, half the LDS operations of the two sub-dword arrays), contiguous 16-bit
//                       runs out.  A run that does not fit its region any more sets the batch's overflow flag (and is cut).
//   bloom_fill2_kernel  one workgroup per (pack, 8 segments): the regions start 16-byte aligned, so a lane takes eight
//                       entries per load (the first version's 2-byte loads: 2.7 TB/s); ds_or; one coalesced store.
// Heavy hitters (a value in a large share of a pack's rows while the filter is sized for many distinct values) overflow
// their four segments.  OR is idempotent, but the regions are not sets: the host reads the flag after the batch and runs
// the exact count / scan / bin / fill passes for that batch (and the rest of the column) -- same bytes either way
// (tests/test_gpu_misc.py builds such a column).
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t bloom_seg_count(uint64_t fbytes) { return (uint32_t)((fbytes * 8 + (1u << kSegShift) - 1) >> kSegShift); }
__host__ __device__ inline uint32_t bloom_seg_cap(uint64_t pack_rows, uint32_t n_seg) {
    const uint64_t mean = (pack_rows * 4 + n_seg - 1) / n_seg;
    return (uint32_t)((mean + mean / 8 + 256 + 7) & ~7ull);  // a multiple of 8 entries: every region starts 16-byte aligned
}

template <int W, typename H>
__global__ void __launch_bounds__(kBlock) bloom_bin1_kernel(H hasher, uint64_t n_rows, uint64_t pack_rows, uint64_t p0, uint32_t tiles_pp,
                                                            const uint64_t *offsets, uint32_t *seg_cursor, uint16_t *entries, uint64_t entries_pp,
                                                            uint32_t *overflow_flag) {
    __shared__ uint32_t hist[256];    // the tile's run length per segment (filled by the ranking pass)
    __shared__ uint32_t lstart[256];  // the run's start inside the tile's sorted order
    __shared__ uint32_t gpos[256];    // where the run goes inside its segment's region
    __shared__ uint32_t wsum[kBlock / 64];
    __shared__ uint16_t sorted[kBinSlots];  // in-segment offsets, runs of one segment contiguous (32 KiB: four workgroups per CU)
    const uint32_t tile = blockIdx.x % tiles_pp;
    const uint64_t pl = blockIdx.x / tiles_pp, p = p0 + pl;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    const uint64_t fbytes = o1 - o0 - 1;
    const uint32_t mask = (uint32_t)(fbytes * 8 - 1);
    const uint32_t cap = bloom_seg_cap(pack_rows, bloom_seg_count(fbytes));
    hist[threadIdx.x] = 0;
    __syncthreads();
    uint64_t h[kBinRowsPerThread];
    const uint32_t valid = bloom_tile_hashes<W>(hasher, p, pack_rows, n_rows, tile, h);
    // ranks of this thread's 64 locations inside their (tile, segment) runs, two per register
    uint32_t rank2[kBinRowsPerThread * 2];
#pragma unroll
    for (int i = 0; i < kBinRowsPerThread; i++) {
        uint32_t h0 = (uint32_t)h[i];
        const uint32_t h1 = (uint32_t)(h[i] >> 32);
        uint32_t r[4] = {0, 0, 0, 0};
        if (valid & (1u << i)) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[k] = atomicAdd(&hist[(h0 & mask) >> kSegShift], 1u);
                h0 += h1;
            }
        }
        rank2[2 * i] = r[0] | (r[1] << 16);      // a run holds at most 16384 entries
        rank2[2 * i + 1] = r[2] | (r[3] << 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = hist[threadIdx.x];
    // the run's place in its segment's region: ONE device-scope atomic per non-empty run, issued now and consumed only after
    // the LDS reorder below -- its round trip to the memory side (microseconds) is hidden behind that phase
    uint32_t g = 0;
    if (c) g = atomicAdd(&seg_cursor[pl * 256 + threadIdx.x], c);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    lstart[threadIdx.x] = base + incl - c;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kBinRowsPerThread; i++) {
        if (valid & (1u << i)) {
            uint32_t h0 = (uint32_t)h[i];
            const uint32_t h1 = (uint32_t)(h[i] >> 32);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t loc = h0 & mask;
                const uint32_t rk = (rank2[2 * i + (k >> 1)] >> ((k & 1) * 16)) & 0xffffu;
                sorted[lstart[loc >> kSegShift] + rk] = (uint16_t)loc;
                h0 += h1;
            }
        }
    }
    if (c && g + c > cap) atomicOr(overflow_flag, 1u);
    gpos[threadIdx.x] = g;
    __syncthreads();
    // write-out by runs: wave w takes the segments w, w + 4, ...: lane l first fetches the run (length, start in the tile, place in
    // the region) of segment w + 4 l, then the wave walks its 64 runs with those three values broadcast from the owning lane
    // (v_readlane: no LDS round trip per run for the bookkeeping), four runs per step so that four entry reads are in flight;
    // the lanes copy a run's entries side by side (a run is ~64 entries = one 128-byte store per wave)
    uint16_t *dst = entries + pl * entries_pp;
    const uint32_t my_sg = (uint32_t)wave + 4u * (uint32_t)lane;
    const uint32_t my_n = hist[my_sg], my_ls = lstart[my_sg], my_at = gpos[my_sg];
#pragma unroll 1
    for (int l0 = 0; l0 < 64; l0 += 4) {
        uint32_t n[4], ls[4], at0[4];
        uint16_t e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            n[k] = (uint32_t)__builtin_amdgcn_readlane((int)my_n, l0 + k);
            ls[k] = (uint32_t)__builtin_amdgcn_readlane((int)my_ls, l0 + k);
            at0[k] = (uint32_t)__builtin_amdgcn_readlane((int)my_at, l0 + k);
            e[k] = (uint32_t)lane < n[k] ? sorted[ls[k] + lane] : (uint16_t)0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t sg = (uint32_t)wave + 4u * (uint32_t)(l0 + k);
            uint16_t *d = dst + (uint64_t)sg * cap;
            if ((uint32_t)lane < n[k] && at0[k] + lane < cap) d[at0[k] + lane] = e[k];
            for (uint32_t i = 64 + lane; i < n[k]; i += 64)   // runs longer than a wave (skewed tiles)
                if (at0[k] + i < cap) d[at0[k] + i] = sorted[ls[k] + i];
        }
    }
}

__global__ void __launch_bounds__(kBlock) bloom_fill2_kernel(uint64_t pack_rows, uint64_t p0, uint32_t groups_pp, uint8_t *filters, const uint64_t *offsets,
                                                             const uint16_t *entries, uint64_t entries_pp, const uint32_t *seg_cursor) {
    __shared__ uint32_t bm[kFillSegs * 2048 + 4];
    const uint32_t g = blockIdx.x % groups_pp;
    const uint64_t pl = blockIdx.x / groups_pp, p = p0 + pl;
    const uint64_t o0 = offsets[p], o1 = offsets[p + 1];
    if (o1 <= o0 + 1) return;
    const uint64_t fbytes = o1 - o0 - 1;
    const uint32_t n_seg = bloom_seg_count(fbytes);
    const uint32_t cap = bloom_seg_cap(pack_rows, n_seg);
    const uint32_t seg0 = g * kFillSegs;
    if (seg0 >= n_seg) return;
    if (g == 0 && threadIdx.x == 0) filters[o0] = 4;          // k (bloom.go:48-60)
    for (int i = threadIdx.x; i < kFillSegs * 2048 + 4; i += kBlock) bm[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint16_t *pe = entries + pl * entries_pp;
    for (uint32_t sl = wave; sl < kFillSegs && seg0 + sl < n_seg; sl += kBlock / 64) {
        uint32_t *seg_bm = bm + sl * 2048;
        const uint32_t cur = seg_cursor[pl * 256 + seg0 + sl];
        const uint32_t cnt = cur < cap ? cur : cap;
        const u32x4 *src = (const u32x4 *)(pe + (uint64_t)(seg0 + sl) * cap);   // 16-byte aligned: cap and entries_pp are multiples of 8 entries
        const uint32_t n_vec = (cnt + 7) >> 3;
        constexpr int U = 8;                                  // 8 x 16 bytes in flight per lane
        for (uint32_t v0 = lane; v0 < n_vec; v0 += 64 * U) {
            u32x4 r[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t v = v0 + (uint32_t)u * 64;
                r[u] = v < n_vec ? __builtin_nontemporal_load(src + v) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t v = v0 + (uint32_t)u * 64;
                if (v >= n_vec) break;
                const uint32_t left = cnt - v * 8;            // entries of this vector that exist (>= 1)
                const uint32_t w[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if ((uint32_t)j < left) {
                        const uint32_t off = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                        atomicOr(&seg_bm[off >> 5], 1u << (off & 31));
                    }
                }
            }
        }
    }
    __syncthreads();
    // write out: the bit array starts at filters + o0 + 1 (any byte alignment)
    uint64_t nbytes = fbytes - (uint64_t)seg0 * 8192;
    if (nbytes > (uint64_t)kFillSegs * 8192) nbytes = (uint64_t)kFillSegs * 8192;
    uint8_t *dst = filters + o0 + 1 + (uint64_t)seg0 * 8192;
    const uint8_t *bmb = (const uint8_t *)bm;
    uint32_t head = (uint32_t)((4 - ((uintptr_t)dst & 3)) & 3);
    if (head > nbytes) head = (uint32_t)nbytes;
    const uint32_t nd = (uint32_t)((nbytes - head) >> 2);     // aligned dwords
    const uint32_t tail0 = head + nd * 4;
    if (threadIdx.x < head) dst[threadIdx.x] = bmb[threadIdx.x];
    if (threadIdx.x >= 8 && threadIdx.x - 8 < nbytes - tail0) dst[tail0 + threadIdx.x - 8] = bmb[tail0 + threadIdx.x - 8];
    uint32_t *dw = (uint32_t *)(dst + head);
    for (uint32_t j = threadIdx.x; j < nd; j += kBlock) {
        const uint32_t b = head + 4 * j;                      // byte offset inside the bitmap
        const uint32_t w0 = bm[b >> 2], w1 = bm[(b >> 2) + 1];
        dw[j] = __builtin_amdgcn_alignbyte(w1, w0, b & 3);
    }
}

// scratch of a one-read batch: fixed regions (n_seg x cap entries per pack, sized for the largest filter), the cursors, the flag
static uint64_t bloom_one_entries_pp(uint64_t pack_rows, uint64_t max_fbytes) {
    const uint32_t n_seg = bloom_seg_count(max_fbytes);
    return (uint64_t)n_seg * bloom_seg_cap(pack_rows, n_seg) + 1024;  // + slack: a pack with a smaller filter has fewer, larger regions (rounding)
}
size_t bloom_one_scratch_bytes(uint64_t packs, uint64_t pack_rows, uint64_t max_fbytes) {
    return (size_t)packs * (bloom_one_entries_pp(pack_rows, max_fbytes) * 2 + 256 * 4) + 512;
}

// packs [p0, p0 + np) through the one-read build.  *overflow_flag_dev (a device word inside scratch, zeroed here) is non-zero
// afterwards when some region overflowed: the caller then rebuilds these packs with launch_bloom_build_part.
hipError_t launch_bloom_build_one(hipStream_t s, int width, const void *col, const BytesView *v, uint64_t n_rows, uint64_t pack_rows, uint64_t p0,
                                  uint64_t np, uint64_t max_fbytes, uint8_t *filters, const uint64_t *offsets, void *scratch, uint32_t **overflow_flag_dev) {
    if (np == 0) return hipSuccess;
    const uint32_t tiles_pp = bloom_part_tiles(pack_rows);
    const uint64_t epp = bloom_one_entries_pp(pack_rows, max_fbytes);
    uint16_t *entries = (uint16_t *)scratch;
    uint32_t *seg_cursor = (uint32_t *)(entries + np * epp);
    uint32_t *flag = seg_cursor + np * 256;
    *overflow_flag_dev = flag;
    hipError_t e = launch_zero_bytes(s, seg_cursor, np * 256 * 4 + 64);
    if (e != hipSuccess) return e;
    const dim3 grid1((unsigned)(np * tiles_pp)), block(kBlock);
    if (width == 0) {
        HashBytes hb{v->ptr, v->buf, v->pool_base};
        hipLaunchKernelGGL((bloom_bin1_kernel<8, HashBytes>), grid1, block, 0, s, hb, n_rows, pack_rows, p0, tiles_pp, offsets, seg_cursor, entries, epp, flag);
    } else {
        HashNumeric hn{col};
        switch (width) {
        case 8: hipLaunchKernelGGL((bloom_bin1_kernel<8, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, seg_cursor, entries, epp, flag); break;
        case 4: hipLaunchKernelGGL((bloom_bin1_kernel<4, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, seg_cursor, entries, epp, flag); break;
        case 2: hipLaunchKernelGGL((bloom_bin1_kernel<2, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, seg_cursor, entries, epp, flag); break;
        case 1: hipLaunchKernelGGL((bloom_bin1_kernel<1, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, seg_cursor, entries, epp, flag); break;
        default: return hipErrorInvalidValue;
        }
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t n_seg = bloom_seg_count(max_fbytes);
    const uint32_t groups_pp = (n_seg + kFillSegs - 1) / kFillSegs;
    hipLaunchKernelGGL(bloom_fill2_kernel, dim3((unsigned)(np * groups_pp)), block, 0, s, pack_rows, p0, groups_pp, filters, offsets, entries, epp, seg_cursor);
    return hipGetLastError();
}

uint32_t bloom_part_tiles(uint64_t pack_rows) { return (uint32_t)((pack_rows + kBinTileRows - 1) / kBinTileRows); }
static uint64_t bloom_entries_pp(uint64_t pack_rows) { return (pack_rows * 4 + 127) & ~(uint64_t)127; }
// scratch of a batch: the 16-bit entries (8 B per row), the segment counts / cursors and the segment bases of every pack
size_t bloom_part_scratch_bytes(uint64_t packs, uint64_t pack_rows) { return (size_t)packs * (bloom_entries_pp(pack_rows) * 2 + 256 * 4 + 260 * 4) + 256; }

// packs [p0, p0 + np) of the column through the partitioned build; scratch = bloom_part_scratch_bytes(np, pack_rows) bytes, max_fbytes
// = the largest filter (bit array bytes) among them.  width 0 = bytes column (v), else numeric (col)
hipError_t launch_bloom_build_part(hipStream_t s, int width, const void *col, const BytesView *v, uint64_t n_rows, uint64_t pack_rows, uint64_t p0,
                                   uint64_t np, uint64_t max_fbytes, uint8_t *filters, const uint64_t *offsets, void *scratch) {
    if (np == 0) return hipSuccess;
    const uint32_t tiles_pp = bloom_part_tiles(pack_rows);
    const uint64_t epp = bloom_entries_pp(pack_rows);
    uint16_t *entries = (uint16_t *)scratch;
    uint32_t *seg_counts = (uint32_t *)(entries + np * epp);
    uint32_t *seg_base = seg_counts + np * 256;
    hipError_t e = launch_zero_bytes(s, seg_counts, np * 256 * 4);
    if (e != hipSuccess) return e;
    const dim3 grid1((unsigned)(np * tiles_pp)), block(kBlock);
#define KNOX_BLOOM_PASS(KERN, ...)                                                                                                       \
    do {                                                                                                                                 \
        if (width == 0) {                                                                                                                \
            HashBytes hb{v->ptr, v->buf, v->pool_base};                                                                                  \
            hipLaunchKernelGGL((KERN<8, HashBytes>), grid1, block, 0, s, hb, n_rows, pack_rows, p0, tiles_pp, offsets, __VA_ARGS__);   \
        } else {                                                                                                                         \
            HashNumeric hn{col};                                                                                                         \
            switch (width) {                                                                                                             \
            case 8: hipLaunchKernelGGL((KERN<8, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL((KERN<4, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, __VA_ARGS__); break; \
            case 2: hipLaunchKernelGGL((KERN<2, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, __VA_ARGS__); break; \
            case 1: hipLaunchKernelGGL((KERN<1, HashNumeric>), grid1, block, 0, s, hn, n_rows, pack_rows, p0, tiles_pp, offsets, __VA_ARGS__); break; \
            default: return hipErrorInvalidValue;                                                                                        \
            }                                                                                                                            \
        }                                                                                                                                \
        e = hipGetLastError();                                                                                                           \
        if (e != hipSuccess) return e;                                                                                                   \
    } while (0)
    KNOX_BLOOM_PASS(bloom_count_kernel, seg_counts);
    hipLaunchKernelGGL(bloom_scan_kernel, dim3((unsigned)((np + kBlock / 64 - 1) / (kBlock / 64))), block, 0, s, np, seg_counts, seg_base);
    KNOX_BLOOM_PASS(bloom_bin_kernel, seg_counts, entries, epp);
#undef KNOX_BLOOM_PASS
    const uint32_t n_seg = (uint32_t)((max_fbytes * 8 + (1u << kSegShift) - 1) >> kSegShift);
    const uint32_t groups_pp = (n_seg + kFillSegs - 1) / kFillSegs;
    hipLaunchKernelGGL(bloom_fill_kernel, dim3((unsigned)(np * groups_pp)), block, 0, s, p0, groups_pp, filters, offsets, entries, epp, seg_base);
    return hipGetLastError();
}

hipError_t launch_bloom_build_bytes(hipStream_t s, const BytesView &v, uint8_t *filters, const uint64_t *offsets) {
    const uint64_t n_packs = (v.n_rows + v.pack_rows - 1) / v.pack_rows;
    if (n_packs == 0) return hipSuccess;
    KNOX_LAUNCH(bloom_build_bytes_kernel, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), v, n_packs, filters, offsets);
}

hipError_t launch_bloom_build(hipStream_t s, int width, const void *col, uint64_t n_rows, uint64_t pack_rows, uint8_t *filters,
                              const uint64_t *offsets) {
    const uint64_t n_packs = (n_rows + pack_rows - 1) / pack_rows;
    if (n_packs == 0) return hipSuccess;
    switch (width) {
    case 8: KNOX_LAUNCH(bloom_build_kernel<8>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    case 4: KNOX_LAUNCH(bloom_build_kernel<4>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    case 2: KNOX_LAUNCH(bloom_build_kernel<2>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    case 1: KNOX_LAUNCH(bloom_build_kernel<1>, (int)(((n_packs + 7) / 8) * 8 * kBloomSplit), col, n_rows, pack_rows, n_packs, filters, offsets);
    default: return hipErrorInvalidValue;
    }
}


}  // namespace knox
