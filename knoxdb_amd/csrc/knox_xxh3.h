// knox_xxh3.h -- XXH3-64 (seed 0, default secret) of a byte string: hash.Hash of the reference
// (internal/hash/hash.go:26 = github.com/zeebo/xxh3 v1.1.0 Hash, go.mod:18, not vendored), written from
// the published XXH3 specification (xxHash v0.8 doc/xxhash_spec.md).  One source for the host ABI
// (knox_hash_bytes: the probe side) and the device (bloom build over bytes columns).
// Include with KNOX_HD defined as `__host__ __device__` (HIP) or empty (host C++), and
// KNOX_XXH3_CONST as `__constant__` (HIP) or empty.
#pragma once
#include <stdint.h>

#ifndef KNOX_HD
#define KNOX_HD
#endif
#ifndef KNOX_XXH3_CONST
#define KNOX_XXH3_CONST
#endif

namespace knox_xxh3 {

static KNOX_XXH3_CONST const uint8_t kSecret[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};

constexpr uint32_t P32_1 = 0x9E3779B1u, P32_2 = 0x85EBCA77u, P32_3 = 0xC2B2AE3Du;
constexpr uint64_t P64_1 = 0x9E3779B185EBCA87ull, P64_2 = 0xC2B2AE3D27D4EB4Full, P64_3 = 0x165667B19E3779F9ull,
                   P64_4 = 0x85EBCA77C2B2AE63ull, P64_5 = 0x27D4EB2F165667C5ull, PMX1 = 0x165667919E3779F9ull,
                   PMX2 = 0x9FB21C651E98DF25ull;

KNOX_HD inline uint64_t rd64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
KNOX_HD inline uint32_t rd32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
KNOX_HD inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
KNOX_HD inline uint64_t bswap(uint64_t x) { return __builtin_bswap64(x); }
KNOX_HD inline uint64_t mul_fold(uint64_t a, uint64_t b) {
    // 64 x 64 -> 128, low ^ high, from four 32 x 32 partial products (no __int128 on the device)
    const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
    const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
    const uint64_t lo = (mid << 32) | (uint32_t)p00;
    const uint64_t hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
    return lo ^ hi;
}
KNOX_HD inline uint64_t avalanche64(uint64_t h) {
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}
KNOX_HD inline uint64_t avalanche3(uint64_t h) {
    h ^= h >> 37; h *= PMX1; h ^= h >> 32;
    return h;
}
KNOX_HD inline uint64_t rrmxmx(uint64_t h, uint64_t len) {
    h ^= rotl(h, 49) ^ rotl(h, 24);
    h *= PMX2;
    h ^= (h >> 35) + len;
    h *= PMX2;
    return h ^ (h >> 28);
}
KNOX_HD inline uint64_t mix16(const uint8_t *in, const uint8_t *sec) {
    return mul_fold(rd64(in) ^ rd64(sec), rd64(in + 8) ^ rd64(sec + 8));
}
KNOX_HD inline void accumulate_512(uint64_t *acc, const uint8_t *in, const uint8_t *sec) {
    for (int i = 0; i < 8; i++) {
        const uint64_t dv = rd64(in + 8 * i), dk = dv ^ rd64(sec + 8 * i);
        acc[i ^ 1] += dv;
        acc[i] += (dk & 0xFFFFFFFFull) * (dk >> 32);
    }
}

// XXH3_64bits of an input of 17 ... 32 bytes from its four words already in registers: a0 a1 = bytes [0, 16), b0 b1 = bytes
// [len - 16, len) -- the same arithmetic as the 17..128 branch of hash() below; lets a kernel issue the loads of many rows
// before any of the multiplies (knox_bloom.hip)
KNOX_HD inline uint64_t hash_17_32_words(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1, uint64_t len) {
    const uint8_t *s = kSecret;
    uint64_t acc = len * P64_1;
    acc += mul_fold(a0 ^ rd64(s), a1 ^ rd64(s + 8));
    acc += mul_fold(b0 ^ rd64(s + 16), b1 ^ rd64(s + 24));
    return avalanche3(acc);
}

// XXH3_64bits of an input of 33 ... 64 bytes from its eight words already in registers: a = bytes [0, 16), b = bytes [16, 32),
// c = bytes [len - 32, len - 16), d = bytes [len - 16, len) -- the 17..128 branch of hash() below for these lengths
KNOX_HD inline uint64_t hash_33_64_words(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1, uint64_t c0, uint64_t c1, uint64_t d0, uint64_t d1,
                                         uint64_t len) {
    const uint8_t *s = kSecret;
    uint64_t acc = len * P64_1;
    acc += mul_fold(b0 ^ rd64(s + 32), b1 ^ rd64(s + 40));   // mix16(in + 16, s + 32)
    acc += mul_fold(c0 ^ rd64(s + 48), c1 ^ rd64(s + 56));   // mix16(in + len - 32, s + 48)
    acc += mul_fold(a0 ^ rd64(s), a1 ^ rd64(s + 8));         // mix16(in, s)
    acc += mul_fold(d0 ^ rd64(s + 16), d1 ^ rd64(s + 24));   // mix16(in + len - 16, s + 16)
    return avalanche3(acc);
}

// XXH3_64bits of an input of at most 32 bytes from words already in registers (every branch of hash() below up to 32 bytes, the
// same arithmetic): w0 = the 8 bytes at in (read even when the input is shorter: the caller's buffer must be readable 7 bytes past
// the input's end), w3 = the input's last 8 bytes (len >= 9), w1 = bytes [8, 16) and w2 = bytes [len - 16, len - 8) (len >= 17).
KNOX_HD inline uint64_t hash_le32_words(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint32_t len) {
    const uint8_t *s = kSecret;
    if (len > 16) return hash_17_32_words(w0, w1, w2, w3, len);
    if (len > 8) {
        const uint64_t lo = w0 ^ (rd64(s + 24) ^ rd64(s + 32)), hi = w3 ^ (rd64(s + 40) ^ rd64(s + 48));
        return avalanche3(len + bswap(lo) + hi + mul_fold(lo, hi));
    }
    if (len >= 4) {
        const uint32_t i1 = (uint32_t)w0, i2 = (uint32_t)(w0 >> (8 * (len - 4)));
        return rrmxmx(((uint64_t)i2 + ((uint64_t)i1 << 32)) ^ (rd64(s + 8) ^ rd64(s + 16)), len);
    }
    if (len > 0) {
        const uint32_t c1 = (uint32_t)w0 & 0xffu, c2 = (uint32_t)(w0 >> (8 * (len >> 1))) & 0xffu, c3 = (uint32_t)(w0 >> (8 * (len - 1))) & 0xffu;
        const uint32_t comb = (c1 << 16) | (c2 << 24) | c3 | (len << 8);
        return avalanche64((uint64_t)comb ^ (uint64_t)(rd32(s) ^ rd32(s + 4)));
    }
    return avalanche64(rd64(s + 56) ^ rd64(s + 64));
}

// XXH3_64bits(in, len)
KNOX_HD inline uint64_t hash(const uint8_t *in, uint64_t len) {
    const uint8_t *s = kSecret;
    if (len == 0) return avalanche64(rd64(s + 56) ^ rd64(s + 64));
    if (len <= 3) {
        const uint8_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
        const uint32_t comb = ((uint32_t)c1 << 16) | ((uint32_t)c2 << 24) | (uint32_t)c3 | ((uint32_t)len << 8);
        return avalanche64((uint64_t)comb ^ (uint64_t)(rd32(s) ^ rd32(s + 4)));
    }
    if (len <= 8) {
        const uint32_t i1 = rd32(in), i2 = rd32(in + len - 4);
        return rrmxmx(((uint64_t)i2 + ((uint64_t)i1 << 32)) ^ (rd64(s + 8) ^ rd64(s + 16)), len);
    }
    if (len <= 16) {
        const uint64_t lo = rd64(in) ^ (rd64(s + 24) ^ rd64(s + 32)), hi = rd64(in + len - 8) ^ (rd64(s + 40) ^ rd64(s + 48));
        return avalanche3(len + bswap(lo) + hi + mul_fold(lo, hi));
    }
    if (len <= 128) {
        uint64_t acc = len * P64_1;
        if (len > 32) {
            if (len > 64) {
                if (len > 96) { acc += mix16(in + 48, s + 96); acc += mix16(in + len - 64, s + 112); }
                acc += mix16(in + 32, s + 64); acc += mix16(in + len - 48, s + 80);
            }
            acc += mix16(in + 16, s + 32); acc += mix16(in + len - 32, s + 48);
        }
        acc += mix16(in, s); acc += mix16(in + len - 16, s + 16);
        return avalanche3(acc);
    }
    if (len <= 240) {
        uint64_t acc = len * P64_1;
        const int rounds = (int)(len / 16);
        for (int i = 0; i < 8; i++) acc += mix16(in + 16 * i, s + 16 * i);
        acc = avalanche3(acc);
        for (int i = 8; i < rounds; i++) acc += mix16(in + 16 * i, s + 16 * (i - 8) + 3);
        acc += mix16(in + len - 16, s + 136 - 17);
        return avalanche3(acc);
    }
    uint64_t acc[8] = {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1};
    const uint64_t stripes_per_block = (192 - 64) / 8, block_len = 64 * stripes_per_block;
    const uint64_t nblocks = (len - 1) / block_len;
    for (uint64_t b = 0; b < nblocks; b++) {
        for (uint64_t t = 0; t < stripes_per_block; t++) accumulate_512(acc, in + b * block_len + t * 64, s + t * 8);
        for (int i = 0; i < 8; i++) {  // scramble
            uint64_t a = acc[i];
            a ^= a >> 47;
            a ^= rd64(s + 192 - 64 + 8 * i);
            a *= P32_1;
            acc[i] = a;
        }
    }
    const uint64_t nstripes = ((len - 1) - block_len * nblocks) / 64;
    for (uint64_t t = 0; t < nstripes; t++) accumulate_512(acc, in + nblocks * block_len + t * 64, s + t * 8);
    accumulate_512(acc, in + len - 64, s + 192 - 64 - 7);
    uint64_t r = len * P64_1;
    for (int i = 0; i < 4; i++) r += mul_fold(acc[2 * i] ^ rd64(s + 11 + 16 * i), acc[2 * i + 1] ^ rd64(s + 11 + 16 * i + 8));
    return avalanche3(r);
}

}  // namespace knox_xxh3
