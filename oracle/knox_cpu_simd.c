/*
 * knox_cpu_simd.c -- CPU BASELINE ONLY (test infrastructure, see knox_oracle.c header).
 *
 * An AVX2, pack-parallel port of the int64 Between + popcount + masked sum step, using the
 * strategy of the reference's AVX2 kernels (internal/cmp/avx2/int64_AVX2_amd64.s:1154-1330:
 * compare 4 x 64-bit lanes, collect the lane masks with movemask, one byte of bitset per 8
 * values, POPCNT for the count).  It is MORE favourable to the CPU than the reference's real
 * flow, which expands the bitset to a selection vector and reduces row by row
 * (reader.go:432, result.go:134-150): here compare and sum are fused like on the GPU.
 * Used by bench.py as the extra "all host cores" figure next to the literal scalar port.
 * An AVX-512 form follows the reference's AVX-512 kernels (internal/cmp/avx512/int64_AVX512_amd64.s:655-…:
 * the unsigned compare lands in a k-mask, 8 lanes = one bitset byte, no movemask / pack step).
 * Built without -march flags; the SIMD paths are target attributes selected at run time
 * (level 0 = the best the CPU has, 1 = scalar, 2 = AVX2, 3 = AVX-512).
 */
#define _GNU_SOURCE
#include <immintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KO_EXPORT __attribute__((visibility("default")))

typedef struct {
    const int64_t *src;
    int64_t n, pack_rows, a, b;
    uint8_t *bits;
    int64_t first_pack, last_pack;
    int64_t count, sum;
    int level;
} job_t;

static void scan_scalar(const int64_t *src, int64_t n, int64_t a, int64_t b, uint8_t *bits, int64_t *count, int64_t *sum) {
    const uint64_t diff = (uint64_t)b - (uint64_t)a;
    int64_t c = 0;
    uint64_t s = 0;
    for (int64_t i = 0; i < n; i += 8) {
        uint8_t byte = 0;
        for (int k = 0; k < 8 && i + k < n; k++) {
            const int hit = ((uint64_t)src[i + k] - (uint64_t)a) <= diff;
            byte |= (uint8_t)(hit << k);
            s += hit ? (uint64_t)src[i + k] : 0;
            c += hit;
        }
        bits[i >> 3] = byte;
    }
    *count += c;
    *sum = (int64_t)((uint64_t)*sum + s);
}

__attribute__((target("avx2,popcnt"))) static void scan_avx2(const int64_t *src, int64_t n, int64_t a, int64_t b, uint8_t *bits,
                                                            int64_t *count, int64_t *sum) {
    const uint64_t diff = (uint64_t)b - (uint64_t)a;
    const __m256i va = _mm256_set1_epi64x(a);
    const __m256i sign = _mm256_set1_epi64x((long long)0x8000000000000000ull);
    const __m256i vd = _mm256_set1_epi64x((long long)(diff ^ 0x8000000000000000ull));
    __m256i acc = _mm256_setzero_si256();
    int64_t c = 0, i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256i x0 = _mm256_loadu_si256((const __m256i *)(src + i));
        const __m256i x1 = _mm256_loadu_si256((const __m256i *)(src + i + 4));
        /* unsigned (x - a) <= diff  <=>  !((x - a) ^ sign  >s  diff ^ sign) */
        const __m256i g0 = _mm256_cmpgt_epi64(_mm256_xor_si256(_mm256_sub_epi64(x0, va), sign), vd);
        const __m256i g1 = _mm256_cmpgt_epi64(_mm256_xor_si256(_mm256_sub_epi64(x1, va), sign), vd);
        const unsigned m = (~((unsigned)_mm256_movemask_pd(_mm256_castsi256_pd(g0)) | ((unsigned)_mm256_movemask_pd(_mm256_castsi256_pd(g1)) << 4))) & 0xffu;
        bits[i >> 3] = (uint8_t)m;
        c += _mm_popcnt_u32(m);
        acc = _mm256_add_epi64(acc, _mm256_andnot_si256(g0, x0));
        acc = _mm256_add_epi64(acc, _mm256_andnot_si256(g1, x1));
    }
    uint64_t lanes[4];
    _mm256_storeu_si256((__m256i *)lanes, acc);
    uint64_t s = lanes[0] + lanes[1] + lanes[2] + lanes[3];
    *count += c;
    *sum = (int64_t)((uint64_t)*sum + s);
    if (i < n) scan_scalar(src + i, n - i, a, b, bits + (i >> 3), count, sum);
}

__attribute__((target("avx512f,avx512bw,avx512dq,popcnt"))) static void scan_avx512(const int64_t *src, int64_t n, int64_t a, int64_t b, uint8_t *bits,
                                                                                    int64_t *count, int64_t *sum) {
    const uint64_t diff = (uint64_t)b - (uint64_t)a;
    const __m512i va = _mm512_set1_epi64(a);
    const __m512i vd = _mm512_set1_epi64((long long)diff);
    __m512i acc0 = _mm512_setzero_si512(), acc1 = _mm512_setzero_si512();
    int64_t c = 0, i = 0;
    for (; i + 64 <= n; i += 64) {  /* 64 values = 8 bitset bytes = one 64-bit store and one POPCNT */
        uint64_t word = 0;
#pragma GCC unroll 8
        for (int k = 0; k < 8; k++) {
            const __m512i x = _mm512_loadu_si512((const void *)(src + i + 8 * k));
            const __mmask8 m = _mm512_cmple_epu64_mask(_mm512_sub_epi64(x, va), vd);
            word |= (uint64_t)m << (8 * k);
            if (k & 1) acc1 = _mm512_mask_add_epi64(acc1, m, acc1, x);
            else acc0 = _mm512_mask_add_epi64(acc0, m, acc0, x);
        }
        memcpy(bits + (i >> 3), &word, 8);
        c += (int64_t)_mm_popcnt_u64(word);
    }
    for (; i + 8 <= n; i += 8) {
        const __m512i x = _mm512_loadu_si512((const void *)(src + i));
        const __mmask8 m = _mm512_cmple_epu64_mask(_mm512_sub_epi64(x, va), vd);
        bits[i >> 3] = (uint8_t)m;
        c += _mm_popcnt_u32((unsigned)m);
        acc0 = _mm512_mask_add_epi64(acc0, m, acc0, x);
    }
    const uint64_t s = (uint64_t)_mm512_reduce_add_epi64(_mm512_add_epi64(acc0, acc1));
    *count += c;
    *sum = (int64_t)((uint64_t)*sum + s);
    if (i < n) scan_scalar(src + i, n - i, a, b, bits + (i >> 3), count, sum);
}

static int best_level(void) {
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("popcnt")) return 3;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt")) return 2;
    return 1;
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    const int best = best_level();
    const int level = j->level <= 0 || j->level > best ? best : j->level;
    for (int64_t p = j->first_pack; p < j->last_pack; p++) {
        const int64_t r0 = p * j->pack_rows;
        const int64_t len = (j->n - r0 < j->pack_rows) ? j->n - r0 : j->pack_rows;
        if (level == 3) scan_avx512(j->src + r0, len, j->a, j->b, j->bits + (r0 >> 3), &j->count, &j->sum);
        else if (level == 2) scan_avx2(j->src + r0, len, j->a, j->b, j->bits + (r0 >> 3), &j->count, &j->sum);
        else scan_scalar(j->src + r0, len, j->a, j->b, j->bits + (r0 >> 3), &j->count, &j->sum);
    }
    return NULL;
}

static int64_t simd_scan(const int64_t *src, int64_t n, int64_t pack_rows, int64_t a, int64_t b, uint8_t *bits, int threads, int level, int64_t *sum_out);

/* pack-parallel over `threads` host threads (pack_rows must be a multiple of 8); returns the count.  AVX2 (or scalar without it): the
 * round-1/2 baseline, kept under its name */
KO_EXPORT int64_t ko_simd_scan_i64_bw_sum(const int64_t *src, int64_t n, int64_t pack_rows, int64_t a, int64_t b, uint8_t *bits,
                                          int threads, int64_t *sum_out) {
    return simd_scan(src, n, pack_rows, a, b, bits, threads, 2, sum_out);
}
/* the same at an explicit level: 0 = best available, 1 = scalar, 2 = AVX2, 3 = AVX-512 (a level the CPU lacks falls to its best) */
KO_EXPORT int64_t ko_simd_scan_i64_bw_sum_level(const int64_t *src, int64_t n, int64_t pack_rows, int64_t a, int64_t b, uint8_t *bits,
                                                int threads, int level, int64_t *sum_out) {
    return simd_scan(src, n, pack_rows, a, b, bits, threads, level, sum_out);
}
KO_EXPORT int ko_simd_best_level(void) { return best_level(); }

static int64_t simd_scan(const int64_t *src, int64_t n, int64_t pack_rows, int64_t a, int64_t b, uint8_t *bits, int threads, int level,
                         int64_t *sum_out) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    const int64_t n_packs = (n + pack_rows - 1) / pack_rows;
    pthread_t tid[256];
    job_t jobs[256];
    if (threads == 1) { /* on the calling thread: what a cmp call on one goroutine is (tools/cmp_latency.py times small vectors) */
        jobs[0] = (job_t){src, n, pack_rows, a, b, bits, 0, n_packs, 0, 0, level};
        worker(&jobs[0]);
        *sum_out = jobs[0].sum;
        return jobs[0].count;
    }
    for (int t = 0; t < threads; t++) {
        jobs[t] = (job_t){src, n, pack_rows, a, b, bits, n_packs * t / threads, n_packs * (t + 1) / threads, 0, 0, level};
        pthread_create(&tid[t], NULL, worker, &jobs[t]);
    }
    int64_t count = 0;
    uint64_t sum = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(tid[t], NULL);
        count += jobs[t].count;
        sum += (uint64_t)jobs[t].sum;
    }
    *sum_out = (int64_t)sum;
    return count;
}

KO_EXPORT int ko_simd_has_avx2(void) { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt"); }
