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
