// knox_bytes.hip -- bytes (StringPool) column matchers: EQ / NE streaming form, ordered modes, IN / NOT IN, REGEXP.
#include "knox_bytes_dev.h"
#include <stdlib.h>
#include <string.h>
#define KNOX_HD __host__ __device__
#include "knox_regex.h"

namespace knox {

// ---------------------------------------------------------------------------------------------
// bytes column EQ / NE: pkg/stringx/match.go:20-84, layout pool.go:45-58 (ptr = off<<32 | len)
// one lane = one string; wave ballot -> 8 result bytes written by lanes 0..7
// ---------------------------------------------------------------------------------------------

// mode = FilterMode: 1 eq, 2 ne (MatchEqual / MatchNotEqual stringx/match.go:20-84), 3 gt, 4 ge, 5 lt,
// 6 le, 9 range [val, val2] (MatchGreater.. / MatchBetween :86-251)
__global__ void __launch_bounds__(kBlock) bytes_match_kernel(int mode, BytesView v, const uint8_t *val, uint32_t vlen, const uint8_t *val2,
                                                             uint32_t vlen2, uint8_t *bits, const uint8_t *mask,
                                                             unsigned long long *count_out) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    uint64_t cnt = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row0 = bv_group_row0(v, g);
        const uint8_t *pool = bv_pool(v, row0);
        const uint64_t row = row0 + lane;
        bool m = false;
        if (row < v.n_rows) {
            const uint64_t p = v.ptr[row];
            const uint32_t ofs = (uint32_t)(p >> 32), len = (uint32_t)p;
            if (mode <= 2) {
                const bool eq = (len == vlen) && bytes_equal(pool + ofs, val, len);
                m = mode == 1 ? eq : !eq;
            } else {
                const int c = bytes_cmp3(pool + ofs, len, val, vlen);
                switch (mode) {
                case 3: m = c > 0; break;
                case 4: m = c >= 0; break;
                case 5: m = c < 0; break;
                case 6: m = c <= 0; break;
                default: m = c >= 0 && bytes_cmp3(pool + ofs, len, val2, vlen2) <= 0; break;
                }
            }
        }
        const uint64_t ballot = __ballot(m);
        bv_store_group(ballot, row0, v.n_rows, lane, bits, mask, false);
        if (lane == 0) cnt += __builtin_popcountll(ballot);
    }
    if (lane == 0 && cnt && count_out) atomicAdd(count_out, (unsigned long long)cnt);
}

// EQ / NE against a value of >= 16 bytes: the streaming form.  A wave takes kEqGroups groups of 64 rows per
// iteration: all row pointers first, then for every row whose length matches TWO independent 16-byte loads -- bytes
// [0, 16) and [min(16, vlen - 16), +16), which cover the whole value up to 32 bytes -- so 12 loads per lane are in
// flight instead of a pointer -> first piece -> second piece chain.  Values above 32 bytes finish in bytes_equal for
// the rows that still match (rare).  Per row: 8 B pointer + len B payload read, 1/8 B written.
// (Round 3 also tried the scan kernel's deferred, phase-aligned bursts here: a persistent kernel whose workgroups take chunks of
// 8 x 1024 rows off a device counter in order, park the 128-byte bitset lines in LDS and flush on the clock's phases -- 0.79-0.80
// of peak at 4 / 8 / 12 workgroups per CU against 0.843-0.845 for this one-workgroup-per-1024-rows form (one unit per dequeue: the
// single counter word saturates at ~80 atomics per us, 0.39).  The pointer -> payload chain wants the parallelism of many short
// workgroups more than it minds the trickle of bitset writes, which are 0.3 % of its traffic.)
// (The payload pieces stay ordinary cached loads: as streaming (nt) loads -- round 3, same bench -- the kernel fell from 0.848 to
// 0.78 of peak; the two pieces of a row and the rows of a 128-byte line share sectors, which only a cache can serve once.)
constexpr int kEqGroups = 4;
__global__ void __launch_bounds__(kBlock) bytes_eq_kernel(int negate, BytesView v, const uint8_t *val, uint32_t vlen, uint8_t *bits,
                                                          const uint8_t *mask, unsigned long long *count_out) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    const uint32_t off_b = vlen >= 32u ? 16u : vlen - 16u;
    u32x4 va, vb;
    __builtin_memcpy(&va, val, 16);
    __builtin_memcpy(&vb, val + off_b, 16);
    uint64_t cnt = 0;
    for (uint64_t g0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kEqGroups; g0 < ngroups; g0 += nwaves_total * kEqGroups) {
        uint64_t row0[kEqGroups], p[kEqGroups];
        const uint8_t *src[kEqGroups];
#pragma unroll
        for (int k = 0; k < kEqGroups; k++) {
            const uint64_t g = g0 + k;
            row0[k] = g < ngroups ? bv_group_row0(v, g) : v.n_rows;
            const uint64_t row = row0[k] + lane;
            p[k] = row < v.n_rows ? __builtin_nontemporal_load(v.ptr + row) : ~0ull;  // len 2^32-1 != vlen: never loads
        }
        u32x4 xa[kEqGroups], xb[kEqGroups];
#pragma unroll
        for (int k = 0; k < kEqGroups; k++) {
            src[k] = (row0[k] < v.n_rows ? bv_pool(v, row0[k]) : v.buf) + (uint32_t)(p[k] >> 32);
            xa[k] = va;
            xb[k] = vb;
            if ((uint32_t)p[k] == vlen) {
                __builtin_memcpy(&xa[k], src[k], 16);
                __builtin_memcpy(&xb[k], src[k] + off_b, 16);
            }
        }
#pragma unroll
        for (int k = 0; k < kEqGroups; k++) {
            if (g0 + k >= ngroups) break;
            bool eq = (uint32_t)p[k] == vlen &&
                      (((xa[k].x ^ va.x) | (xa[k].y ^ va.y) | (xa[k].z ^ va.z) | (xa[k].w ^ va.w) | (xb[k].x ^ vb.x) | (xb[k].y ^ vb.y) |
                        (xb[k].z ^ vb.z) | (xb[k].w ^ vb.w)) == 0);
            if (eq && vlen > 32u) eq = bytes_equal(src[k] + 32, val + 32, vlen - 32u);
            const bool m = (row0[k] + lane < v.n_rows) && (negate ? !eq : eq);
            const uint64_t ballot = __ballot(m);
            bv_store_group(ballot, row0[k], v.n_rows, lane, bits, mask, false);
            if (lane == 0) cnt += __builtin_popcountll(ballot);
        }
    }
    if (lane == 0 && cnt && count_out) atomicAdd(count_out, (unsigned long long)cnt);
}

// Workgroups of the ordered-mode and REGEXP kernels: beyond the cap they walk the column grid-stride.  Measured on 2^28 x
// 32-byte keys (tools/bench_bytes.py with KNOX_BYTES_CAP_LOG2 = 13 / 16 / 20, profiles/r03_bytes_grid.txt): unlike bytes EQ
// -- which gained 10 points from "one group per wave, launch order" -- the ordered modes LOSE with it (LT 5.58 -> 5.32,
// RANGE 4.91 -> 4.49 TB/s of 40.125 B/row accounting; a random key is decided by its first 8 bytes, so these kernels are
// pointer-chasing latency, not payload streaming) and are best at 2^16 workgroups (LT 5.75, RANGE 4.96); REGEXP stages its
// DFA in LDS per workgroup and wants few of them (2^13: 6.1 ms, 2^20: 8.7 ms); IN (binary search per row) does not care.
static uint64_t bytes_grid_cap(int dflt_log2) {
    static const int forced = [] {
        const char *e = getenv("KNOX_BYTES_CAP_LOG2");
        return e && atoi(e) >= 8 && atoi(e) <= 24 ? atoi(e) : 0;
    }();
    return (uint64_t)1 << (forced ? forced : dflt_log2);
}

// Ordered modes (LT / LE / GT / GE / RANGE), the streaming form (round 4).  bytes.Compare of two strings is decided by their zero-padded
// 8-byte big-endian PREFIXES whenever those differ (a proper prefix sorts first, and its padding is the smallest byte), and a random key
// is decided by its first bytes: a wave takes four groups of 64 rows per iteration -- the pointers, then the first 8 bytes of every row, in
// flight together -- and only a row whose prefix EQUALS the value's goes through the full compare.  (bytes_match_kernel walks one group
// at a time: pointer -> payload is a dependent chain, 5.5 TB/s for LT and 4.7 for RANGE where EQ streams at 6.4.)  A row's first 8 bytes
// are read whole even when the row is shorter (the pools end in 256 bytes of slack) and cut to its length.
__global__ void __launch_bounds__(kBlock) bytes_ord_kernel(int mode, BytesView v, const uint8_t *val, uint32_t vlen, const uint8_t *val2, uint32_t vlen2,
                                                           uint8_t *bits, const uint8_t *mask, unsigned long long *count_out) {
    constexpr int kG = 4;
    const int lane = threadIdx.x & 63;
    // the values' prefixes (the value buffer is 16 bytes longer than the values: knox_abi.cpp)
    auto prefix_of = [](const uint8_t *q, uint32_t n) {
        uint64_t raw = 0;
        __builtin_memcpy(&raw, q, 8);
        if (n < 8) raw &= n ? (1ull << (8 * n)) - 1 : 0ull;
        return __builtin_bswap64(raw);
    };
    const uint64_t pre1 = prefix_of(val, vlen), pre2 = mode > 6 ? prefix_of(val2, vlen2) : 0ull;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    uint64_t cnt = 0;
    for (uint64_t g0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kG; g0 < ngroups; g0 += nwaves_total * kG) {
        uint64_t row0[kG], p[kG], pre[kG];
        const uint8_t *x[kG];
#pragma unroll
        for (int k = 0; k < kG; k++) {
            const bool there = g0 + k < ngroups;                         // wave uniform
            row0[k] = there ? bv_group_row0(v, g0 + k) : 0;
            const uint64_t row = row0[k] + lane;
            p[k] = (there && row < v.n_rows) ? v.ptr[row] : ~0ull;      // len 0xffffffff: no row
        }
#pragma unroll
        for (int k = 0; k < kG; k++) {
            const uint8_t *pool = g0 + k < ngroups ? bv_pool(v, row0[k]) : v.buf;
            x[k] = pool + (uint32_t)(p[k] >> 32);
            uint64_t raw = 0;
            if (p[k] != ~0ull) __builtin_memcpy(&raw, x[k], 8);
            const uint32_t len = (uint32_t)p[k];
            if (len < 8) raw &= len ? (1ull << (8 * len)) - 1 : 0ull;
            pre[k] = __builtin_bswap64(raw);
        }
#pragma unroll
        for (int k = 0; k < kG; k++) {
            if (g0 + k >= ngroups) break;                                // wave uniform
            bool m = false;
            if (p[k] != ~0ull) {
                const uint32_t len = (uint32_t)p[k];
                int c = pre[k] < pre1 ? -1 : pre[k] > pre1 ? 1 : bytes_cmp3(x[k], len, val, vlen);
                switch (mode) {
                case 3: m = c > 0; break;
                case 4: m = c >= 0; break;
                case 5: m = c < 0; break;
                case 6: m = c <= 0; break;
                default:
                    if (c >= 0) {
                        const int c2 = pre[k] < pre2 ? -1 : pre[k] > pre2 ? 1 : bytes_cmp3(x[k], len, val2, vlen2);
                        m = c2 <= 0;
                    }
                    break;
                }
            }
            const uint64_t ballot = __ballot(m);
            bv_store_group(ballot, row0[k], v.n_rows, lane, bits, mask, false);
            if (lane == 0) cnt += __builtin_popcountll(ballot);
        }
    }
    if (lane == 0 && cnt && count_out) atomicAdd(count_out, (unsigned long long)cnt);
}

hipError_t launch_bytes_match(hipStream_t s, int mode, const BytesView &v, const uint8_t *val, uint32_t vlen, const uint8_t *val2, uint32_t vlen2,
                              uint8_t *bits, const uint8_t *mask, unsigned long long *count_out) {
    const uint64_t ngroups = v.live ? (uint64_t)v.n_live * (v.pack_rows >> 6) : (v.n_rows + 63) >> 6;
    if (ngroups == 0) return hipSuccess;
    if (mode <= 2 && vlen >= 16u) {
        uint64_t g = (ngroups + (uint64_t)kWavesPerBlock * kEqGroups - 1) / ((uint64_t)kWavesPerBlock * kEqGroups);
        // one wave iteration per wave: measured on 2^30 x 32-byte keys (tools/bench_cfg.py --cfg 4), workgroups that walk
        // the column grid-stride (2048 of them: 7.4-7.5 ms) are 15 % slower than one workgroup per 1024 rows in launch
        // order (6.50 ms = 6.6 TB/s = 83 % of the HBM peak): pointer -> payload is a dependent chain, and neighbours in
        // launch order keep the DRAM pages they touch close together
        if (g > ((uint64_t)1 << 20)) g = (uint64_t)1 << 20;
        KNOX_LAUNCH(bytes_eq_kernel, (int)g, mode == 2 ? 1 : 0, v, val, vlen, bits, mask, count_out);
    }
    const char *form = getenv("KNOX_BYTES_ORD_FORM");
    if (mode >= 3 && !(form && !strcmp(form, "general"))) {   // ordered modes: four groups per wave iteration, decided by the 8-byte prefixes
        uint64_t g = (ngroups + (uint64_t)kWavesPerBlock * 4 - 1) / ((uint64_t)kWavesPerBlock * 4);
        if (g > ((uint64_t)1 << 20)) g = (uint64_t)1 << 20;
        KNOX_LAUNCH(bytes_ord_kernel, (int)g, mode, v, val, vlen, val2, vlen2, bits, mask, count_out);
    }
    uint64_t g = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > bytes_grid_cap(16)) g = bytes_grid_cap(16);
    KNOX_LAUNCH(bytes_match_kernel, (int)g, mode, v, val, vlen, val2, vlen2, bits, mask, count_out);
}

// bytes IN / NOT IN (bytesInSetMatcher / bytesNotInSetMatcher.MatchVector, match_bytes.go:408-470):
// membership of the row's bytes in the set.  The reference consults a hash map and confirms with
// bytes.Equal, i.e. plain equality; here the set arrives sorted (bytes.Compare order) and unique in
// StringPool form and every lane binary-searches it.  Matching rows are OR-ed into bits (bits.Set).
__global__ void __launch_bounds__(kBlock) bytes_inset_kernel(int negate, BytesView v, const uint64_t *set_ptr, uint32_t n_set,
                                                             const uint8_t *set_buf, uint8_t *bits, const uint8_t *mask) {
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row0 = bv_group_row0(v, g);
        const uint8_t *pool = bv_pool(v, row0);
        const uint64_t row = row0 + lane;
        bool m = false;
        if (row < v.n_rows) {
            const uint64_t p = v.ptr[row];
            const uint8_t *x = pool + (uint32_t)(p >> 32);
            const uint32_t len = (uint32_t)p;
            uint32_t lo = 0, hi = n_set;
            bool found = false;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint64_t sp = set_ptr[mid];
                const int c = bytes_cmp3(set_buf + (uint32_t)(sp >> 32), (uint32_t)sp, x, len);
                if (c == 0) { found = true; break; }
                if (c < 0) lo = mid + 1; else hi = mid;
            }
            m = negate ? !found : found;
        }
        const uint64_t ballot = __ballot(m);
        bv_store_group(ballot, row0, v.n_rows, lane, bits, mask, true);
    }
}

// Round 4: the same membership test with the search in LDS.  The kernel above walks the sorted set with a full bytes.Compare per
// step -- ten dependent (set pointer, set bytes, row bytes) fetches through the L2 per row for 583 members: 35 G rows/s on 32-byte
// keys where EQ runs at 160 (tools/bench_bytes.py).  bytes.Compare order is the order of the 8-byte big-endian PREFIXES (zero
// padded) first: the sets' prefixes sit in LDS (<= 4096 members), a branch-free binary search finds the first member whose prefix
// is not below the row's, and only the members sharing the row's prefix (usually one or none) are compared in full.  Four groups of
// 64 rows per wave iteration: the pointers, then the first 8 bytes of every row, are in flight together.  A row's first 8 bytes
// are read even when the row is shorter (the pools end in 256 bytes of slack; knox_abi.cpp) and cut to its length.
constexpr int kInGroups = 4;
constexpr uint32_t kBytesInLdsKeys = 4096;

__global__ void __launch_bounds__(kBlock) bytes_inset_prefix_kernel(int negate, BytesView v, const uint64_t *set_ptr, uint32_t n_set,
                                                                    const uint8_t *set_buf, const uint64_t *set_prefix, uint8_t *bits,
                                                                    const uint8_t *mask) {
    __shared__ uint64_t spre[kBytesInLdsKeys];
    uint32_t pow2 = 1;
    while (pow2 < n_set) pow2 <<= 1;
    for (uint32_t i = threadIdx.x; i < pow2; i += kBlock) spre[i] = i < n_set ? set_prefix[i] : ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    for (uint64_t g0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kInGroups; g0 < ngroups; g0 += nwaves_total * kInGroups) {
        uint64_t row0[kInGroups], p[kInGroups], pre[kInGroups];
        const uint8_t *x[kInGroups];
#pragma unroll
        for (int k = 0; k < kInGroups; k++) {
            const bool there = g0 + k < ngroups;                         // wave uniform
            row0[k] = there ? bv_group_row0(v, g0 + k) : 0;
            const uint64_t row = row0[k] + lane;
            p[k] = (there && row < v.n_rows) ? v.ptr[row] : ~0ull;      // len 0xffffffff: no row
        }
#pragma unroll
        for (int k = 0; k < kInGroups; k++) {
            const uint8_t *pool = g0 + k < ngroups ? bv_pool(v, row0[k]) : v.buf;
            x[k] = pool + (uint32_t)(p[k] >> 32);
            uint64_t raw = 0;
            if (p[k] != ~0ull) __builtin_memcpy(&raw, x[k], 8);
            const uint32_t len = (uint32_t)p[k];
            if (len < 8) raw &= len ? (1ull << (8 * len)) - 1 : 0ull;    // the bytes behind a short row are not the row's
            pre[k] = __builtin_bswap64(raw);
        }
#pragma unroll
        for (int k = 0; k < kInGroups; k++) {
            if (g0 + k >= ngroups) break;                                // wave uniform
            bool m = false;
            if (p[k] != ~0ull) {
                const uint32_t len = (uint32_t)p[k];
                uint32_t pos = 0;                                        // first member whose prefix is >= the row's
                for (uint32_t step = pow2 >> 1; step >= 1; step >>= 1) pos += spre[pos + step - 1] < pre[k] ? step : 0u;
                bool found = false;
                while (pos < n_set && spre[pos] == pre[k]) {
                    const uint64_t sp = set_ptr[pos];
                    if ((uint32_t)sp == len && bytes_equal(x[k], set_buf + (uint32_t)(sp >> 32), len)) { found = true; break; }
                    pos++;
                }
                m = negate ? !found : found;
            }
            const uint64_t ballot = __ballot(m);
            bv_store_group(ballot, row0[k], v.n_rows, lane, bits, mask, true);
        }
    }
}

// set_prefix: the members' first 8 bytes as big-endian integers, zero padded (same order as the sorted set); null or more than
// 4096 members: the general kernel
hipError_t launch_bytes_inset(hipStream_t s, bool negate, const BytesView &v, const uint64_t *set_ptr, uint32_t n_set, const uint8_t *set_buf,
                              const uint64_t *set_prefix, uint8_t *bits, const uint8_t *mask) {
    const uint64_t ngroups = v.live ? (uint64_t)v.n_live * (v.pack_rows >> 6) : (v.n_rows + 63) >> 6;
    if (ngroups == 0) return hipSuccess;
    const char *form = getenv("KNOX_BYTES_IN_FORM");
    if (set_prefix != nullptr && n_set <= kBytesInLdsKeys && !(form && !strcmp(form, "general"))) {
        uint64_t g = (ngroups + (uint64_t)kWavesPerBlock * kInGroups - 1) / ((uint64_t)kWavesPerBlock * kInGroups);
        if (g > bytes_grid_cap(16)) g = bytes_grid_cap(16);
        KNOX_LAUNCH(bytes_inset_prefix_kernel, (int)g, negate ? 1 : 0, v, set_ptr, n_set, set_buf, set_prefix, bits, mask);
    }
    uint64_t g = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > ((uint64_t)1 << 20)) g = (uint64_t)1 << 20;  // latency bound (binary search per row): one group per wave, +14 % over 8192 workgroups
    KNOX_LAUNCH(bytes_inset_kernel, (int)g, negate ? 1 : 0, v, set_ptr, n_set, set_buf, bits, mask);
}

// ---------------------------------------------------------------------------------------------
// bytes REGEXP (operator/filter/match_bytes.go:587-609: re.Match(row) for every row).  The host
// compiles the pattern into a DFA over rune classes (knox_regex.h); one lane = one row: decode a
// rune the way Go does (utf8.DecodeRune, invalid byte = U+FFFD of width 1), map it to its class
// (table for ASCII, binary search above), step table[state][class]; state 0 = ACCEPT (absorbing),
// flags bit1 = dead (no accepting state reachable) -> the lane stops early either way.  Tables of
// up to 48 KiB are staged in LDS, larger ones are read through L2.
// ---------------------------------------------------------------------------------------------
template <bool LDS, bool DIRECT>
__global__ void __launch_bounds__(kBlock) regexp_match_kernel(RegexProg pg, BytesView v, uint8_t *bits, const uint8_t *mask) {
    extern __shared__ __align__(16) uint8_t rx_lds[];
    const uint16_t *table = pg.table, *direct = pg.direct;
    const uint8_t *flags = pg.flags, *ascii = pg.ascii_class;
    if constexpr (LDS) {
        // image layout (host): [direct u16 * n_states*128 (DIRECT only)][table u16 * n_states*n_classes][flags n_states][ascii 128]
        const uint32_t db = DIRECT ? pg.n_states * 256u : 0u;
        const uint32_t tb = pg.n_states * pg.n_classes * 2, total = db + tb + pg.n_states + 128;
        const uint8_t *image = DIRECT ? (const uint8_t *)pg.direct : (const uint8_t *)pg.table;
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) rx_lds[i] = image[i];
        __syncthreads();
        direct = (const uint16_t *)rx_lds;
        table = (const uint16_t *)(rx_lds + db);
        flags = rx_lds + db + tb;
        ascii = flags + pg.n_states;
    }
    // the class intervals of the runes above ASCII: staged too when they are few (they almost always are)
    const uint32_t *bound_lo = pg.bound_lo;
    const uint8_t *bound_class = pg.bound_class;
    if constexpr (LDS) {
        if (pg.n_bounds <= 96u) {
            const uint32_t base = (pg.lds_image_bytes + 3u) & ~3u;
            uint32_t *blo = (uint32_t *)(rx_lds + base);
            uint8_t *bcl = rx_lds + base + 4u * pg.n_bounds;
            for (uint32_t i = threadIdx.x; i < pg.n_bounds; i += blockDim.x) { blo[i] = pg.bound_lo[i]; bcl[i] = pg.bound_class[i]; }
            __syncthreads();
            bound_lo = blo;
            bound_class = bcl;
        }
    }
    const int lane = threadIdx.x & 63;
    const uint64_t nwaves_total = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t ngroups = bv_ngroups(v);
    const uint64_t n_rows = v.n_rows;
    for (uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves_total) {
        const uint64_t row0 = bv_group_row0(v, g);
        const uint8_t *pool = bv_pool(v, row0);
        const uint64_t row = row0 + lane;
        bool m = false;
        if (row < n_rows && (mask == nullptr || ((mask[row >> 3] >> (row & 7)) & 1))) {
            const uint64_t p = v.ptr[row];
            const uint8_t *s = pool + (uint32_t)(p >> 32);
            const uint32_t len = (uint32_t)p;
            uint32_t state = 1, pos = 0;
            bool decided = false;
            // the first 32 bytes of the row are fetched at once (two unaligned 16-byte loads; the pool is padded) and
            // the 4-byte decode window is cut out of registers: one HBM round trip per row instead of one per 5 bytes.
            // Longer rows continue in 8-byte pieces.
            uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            if (len != 0) {
                u32x4 a, b;
                __builtin_memcpy(&a, s, 16);
                __builtin_memcpy(&b, s + 16, 16);
                w0 = ((uint64_t)a.y << 32) | a.x; w1 = ((uint64_t)a.w << 32) | a.z;
                w2 = ((uint64_t)b.y << 32) | b.x; w3 = ((uint64_t)b.w << 32) | b.z;
                if constexpr (DIRECT) {
                    // ASCII prefix of the row, one preloaded dword at a time (static register indexing, no shifts): up to
                    // four table steps per dword; ACCEPT (0) and the dead states are absorbing, so they are looked at
                    // once per dword.  The walk is VALU-issue bound (wave64: 4 cycles per instruction).  A partial last
                    // dword still takes four lookups (bytes past the row masked to 7 bits) and keeps the right one.
                    const uint32_t w32[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (!decided && pos == 4u * k && pos < len) {
                            const uint32_t win = w32[k], rem = len - pos;
                            const uint32_t hm = rem >= 4u ? 0x80808080u : (0x00808080u >> (8u * (3u - rem)));
                            if ((win & hm) == 0u) {
                                const uint32_t s1 = direct[state * 128u + (win & 0x7fu)];
                                const uint32_t s2 = direct[s1 * 128u + ((win >> 8) & 0x7fu)];
                                const uint32_t s3 = direct[s2 * 128u + ((win >> 16) & 0x7fu)];
                                const uint32_t s4 = direct[s3 * 128u + ((win >> 24) & 0x7fu)];
                                state = rem >= 4u ? s4 : rem == 3u ? s3 : rem == 2u ? s2 : s1;
                                pos += rem < 4u ? rem : 4u;
                                if (state == 0u) { m = true; decided = true; }
                                else if (flags[state] & 2) decided = true;
                            }
                        }
                    }
                }
            }
            uint64_t piece = 0;
            uint32_t pbase = 0x80000000u;
            while (!decided && pos < len) {
                uint32_t win;
                if (pos <= 28u) {
                    const uint32_t idx = pos >> 3, sh = (pos & 7u) * 8u;
                    const uint64_t lo = idx == 0 ? w0 : idx == 1 ? w1 : idx == 2 ? w2 : w3;
                    const uint64_t hi = idx == 0 ? w1 : idx == 1 ? w2 : w3;  // idx 3: pos <= 28 keeps the window inside w3
                    win = (uint32_t)(lo >> sh);
                    if (sh > 32u) win |= (uint32_t)(hi << (64u - sh));
                } else {
                    if (pos - pbase > 4u) {
                        __builtin_memcpy(&piece, s + pos, 8);
                        pbase = pos;
                    }
                    win = (uint32_t)(piece >> (8 * (pos - pbase)));
                }
                if constexpr (DIRECT) {
                    // four ASCII bytes inside the row: four table steps back to back.  ACCEPT (state 0) and the dead
                    // states are absorbing, so they are only looked at once per word -- the walk is VALU-issue bound
                    // (wave64: 4 cycles per instruction), every instruction per rune counts
                    if (pos + 4u <= len && (win & 0x80808080u) == 0u) {
                        state = direct[state * 128u + (win & 0xffu)];
                        state = direct[state * 128u + ((win >> 8) & 0xffu)];
                        state = direct[state * 128u + ((win >> 16) & 0xffu)];
                        state = direct[state * 128u + (win >> 24)];
                        pos += 4u;
                        if (state == 0u) { m = true; decided = true; break; }
                        if (flags[state] & 2) { decided = true; break; }
                        continue;
                    }
                }
                uint32_t w;
                const uint32_t r = knox_regex::decode_rune_word(win, len - pos, w);
                pos += w;
                uint32_t e;
                if (DIRECT && r < 0x80) {
                    e = direct[state * 128u + r];  // plain target state (no dead bit in the direct table)
                    if (e != 0 && (flags[e] & 2)) e |= knox_regex::kDeadBit;
                } else {
                    uint32_t cls;
                    if (r < 0x80) cls = ascii[r];
                    else {
                        uint32_t lo = 0, hi = pg.n_bounds;
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (bound_lo[mid] <= r) lo = mid; else hi = mid;
                        }
                        cls = bound_class[lo];
                    }
                    e = table[state * pg.n_classes + cls];
                }
                if (e == 0) { m = true; decided = true; break; }
                if (e & knox_regex::kDeadBit) { decided = true; break; }
                state = e;
            }
            if (!decided) m = flags[state] & 1;
        }
        const uint64_t ballot = __ballot(m);  // unmasked rows never match: with a mask this is bits.Set(i) for the masked rows that do
        bv_store_group(ballot, row0, n_rows, lane, bits, nullptr, mask != nullptr);
    }
}

hipError_t launch_regexp_match(hipStream_t s, const RegexProg &pg, uint32_t image_bytes, const BytesView &v, uint8_t *bits, const uint8_t *mask) {
    const uint64_t ngroups = v.live ? (uint64_t)v.n_live * (v.pack_rows >> 6) : (v.n_rows + 63) >> 6;
    if (ngroups == 0) return hipSuccess;
    uint64_t g = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > bytes_grid_cap(13)) g = bytes_grid_cap(13);
    const uint32_t lds_bytes = ((image_bytes + 3u) & ~3u) + 96u * 5u + 16u;  // image + (<= 96) staged class intervals
    if (pg.direct != nullptr)  // host guarantees: the image with the direct table fits the LDS budget
        hipLaunchKernelGGL((regexp_match_kernel<true, true>), dim3((unsigned)g), dim3(kBlock), lds_bytes, s, pg, v, bits, mask);
    else if (image_bytes <= 48 * 1024)
        hipLaunchKernelGGL((regexp_match_kernel<true, false>), dim3((unsigned)g), dim3(kBlock), lds_bytes, s, pg, v, bits, mask);
    else
        hipLaunchKernelGGL((regexp_match_kernel<false, false>), dim3((unsigned)g), dim3(kBlock), 0, s, pg, v, bits, mask);
    return hipGetLastError();
}


}  // namespace knox
