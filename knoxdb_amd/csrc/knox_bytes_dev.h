// knox_bytes_dev.h -- device helpers for bytes (StringPool) columns: equality / order of a row against a value and the
// walk over a BytesView in groups of 64 rows.  Shared by knox_bytes.hip and knox_bloom.hip.
#pragma once
#include "knox_dev.h"

namespace knox {

__device__ __forceinline__ bool bytes_equal(const uint8_t *p, const uint8_t *v, uint32_t n) {
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) {  // 16-byte pieces: half the load instructions of the row, the value side is wave-uniform
        u32x4 x, y;
        __builtin_memcpy(&x, p + i, 16);
        __builtin_memcpy(&y, v + i, 16);
        if (((x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w)) != 0) return false;
    }
    for (; i + 8 <= n; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, p + i, 8);
        __builtin_memcpy(&y, v + i, 8);
        if (x != y) return false;
    }
    for (; i < n; i++)
        if (p[i] != v[i]) return false;
    return true;
}

// bytes.Compare(p[:n], v[:m]): -1 / 0 / +1, byte-lexicographic, a proper prefix sorts first
__device__ __forceinline__ int bytes_cmp3(const uint8_t *p, uint32_t n, const uint8_t *v, uint32_t m) {
    const uint32_t c = n < m ? n : m;
    uint32_t i = 0;
    for (; i + 8 <= c; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, p + i, 8);
        __builtin_memcpy(&y, v + i, 8);
        if (x != y) return __builtin_bswap64(x) < __builtin_bswap64(y) ? -1 : 1;  // first differing byte decides
    }
    for (; i < c; i++)
        if (p[i] != v[i]) return p[i] < v[i] ? -1 : 1;
    return n == m ? 0 : (n < m ? -1 : 1);
}

// ---- walking a bytes column (BytesView) in groups of 64 rows = one wave ------------------------------------
// Without a live list group g covers rows [64g, 64g + 64).  With one, only the groups of the listed packs exist:
// group g = group (g % gpp) of pack live[g / gpp], gpp = pack_rows / 64.  A group never straddles two packs
// (pack_rows is a multiple of 64), so the pool base is wave-uniform.
__device__ __forceinline__ uint64_t bv_ngroups(const BytesView &v) {
    if (v.live != nullptr) return (uint64_t)v.n_live * (v.pack_rows >> 6);
    return (v.n_rows + 63) >> 6;
}
__device__ __forceinline__ uint64_t bv_group_row0(const BytesView &v, uint64_t g) {
    if (v.live == nullptr) return g << 6;
    uint64_t li, in;
    if (v.pack_shift != 255u) {
        li = g >> (v.pack_shift - 6u);
        in = g & ((1ull << (v.pack_shift - 6u)) - 1ull);
    } else {
        const uint64_t gpp = v.pack_rows >> 6;
        li = g / gpp;
        in = g - li * gpp;
    }
    return (uint64_t)v.live[li] * v.pack_rows + (in << 6);
}
__device__ __forceinline__ const uint8_t *bv_pool(const BytesView &v, uint64_t row0) {
    if (v.pool_base == nullptr) return v.buf;
    const uint64_t pack = v.pack_shift != 255u ? row0 >> v.pack_shift : row0 / v.pack_rows;
    return v.buf + v.pool_base[pack];
}
// the wave's 64 result bits of the group starting at row0 -> bitset bytes: overwrite (mask == null; MatchEqual's
// unmasked form overwrites every byte incl. a partial last one) or bits.Set for mask-selected rows only
__device__ __forceinline__ void bv_store_group(uint64_t ballot, uint64_t row0, uint64_t n_rows, int lane, uint8_t *bits, const uint8_t *mask,
                                               bool or_only) {
    if (mask == nullptr && !or_only && row0 + 64 <= n_rows) {
        if (lane == 0) *(uint64_t *)(bits + (row0 >> 3)) = ballot;  // row0 is a multiple of 64: one aligned 8-byte store
        return;
    }
    if (lane < 8) {
        const uint64_t r0 = row0 + 8 * (uint64_t)lane;
        if (r0 < n_rows) {
            uint8_t byte = (uint8_t)(ballot >> (8 * lane));
            const uint64_t bi = r0 >> 3;
            if (mask != nullptr) byte &= mask[bi];
            if (mask != nullptr || or_only) { if (byte) bits[bi] |= byte; }
            else bits[bi] = byte;
        }
    }
}

}  // namespace knox
