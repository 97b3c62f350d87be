// knox_scan.hip -- the streaming scan kernel (predicate tree of <= 4 leaves -> bitset + per-tile counts + aggregates),
// the dominant kernel of the path.  Compiled once per column width (-DKNOX_SCAN_W=8|4|2|1 -> knox_scan_w8.o ...) so the
// four families of instantiations build in parallel; launch_scan (knox_agg.hip) dispatches on the width.
#include "knox_dev.h"
#include <atomic>
#include <cstdio>

#ifndef KNOX_SCAN_W
#error "compile with -DKNOX_SCAN_W=8|4|2|1"
#endif

namespace knox {

template <int W, int NL, unsigned AGG, bool SEP, bool ANYF, unsigned FLV>
__global__ void __launch_bounds__(kBlock) scan_kernel(const ScanArgs a) {
    using U = typename WT<W>::U;
    constexpr int K = WT<W>::K;
    constexpr int G = 32 / K;
    constexpr int RPL = 64 * K;                       // rows per wave-load
    constexpr int WAVE_ROWS = RPL * kUnroll;
    constexpr uint32_t QMASK = (K == 32) ? 0xffffffffu : ((1u << K) - 1u);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // The wave's bitset words, staged in LDS and written out as 16-byte-per-lane bursts of whole 128-byte lines.
    // Plain filter (+ wrapping sum / count) scans -- the instantiations that run at 2 workgroups per CU and do nothing but
    // stream -- DEFER their bursts: the words of as many consecutive tiles as 15 KiB per wave hold (120 tiles of an 8-byte
    // column at 2 iterations) stay in LDS and leave together, every workgroup at the same rounds.  The memory controllers
    // then see long read-only phases and short write-only ones instead of a write burst every 2.4 us per workgroup, and a
    // bitset that lies in the column's OWN placement class costs 4 % instead of 16 % (round 3, on identical
    // buffers: 0.727 -> 0.835 of peak at 120 tiles, 0.81 at 64, 0.77 at 32; another class: unchanged at 0.87).
    // The same for the aggregate-heavy instantiations that run at 2 workgroups per CU anyway (knox_abi.cpp run_fused): one leaf
    // on 8-, 4- (integers) and 1-byte columns, two leaves (BASELINE configs[2]: +1.9 % on identical buffers).  int32 with all
    // aggregates lost 1.7-2.4 % with unaligned parked bursts and gains 1.0-1.3 % with the aligned ones (tools/ab_kernels.py against
    // a build without / with it); float32 sums want 8 workgroups per CU and keep their tile-by-tile bursts.
    constexpr bool kHeavy = (AGG & ~A_ISUM) != 0;
    // Which instantiations defer, and how many words a wave parks (15 KiB at 2 workgroups per CU: 120 tiles of an 8-byte column
    // at 2 iterations; 12 KiB at 3: 1-byte plain scans, 2-byte aggregate-heavy ones, trees of two plain or of three / four
    // leaves), comes from the SAME table run_fused takes its workgroups per CU from: scan_shape (knox_kernels.h).
    constexpr ScanShape kShape = scan_shape(W, NL, kHeavy, FLV == FL_FLOAT, (AGG & A_FSUM) != 0);
    constexpr uint32_t kParkWords = kShape.park_words;
    constexpr bool kDefer = kParkWords != 0;
    constexpr uint32_t kTileWordsMax = kMaxIters * WAVE_ROWS / 32;
    constexpr uint32_t kStageWords = kTileWordsMax > kParkWords ? kTileWordsMax : kParkWords;
    __shared__ __attribute__((aligned(16))) uint32_t stage[kWavesPerBlock][kStageWords];

    Acc acc;
    acc.init();
    WaveMM wm;
    wm.init();
    FirstLast fl;
    fl.init();
    uint32_t cnt32 = 0;
    const uint32_t iters = a.iters;
    // tiles whose words are parked before one flush (1 = every tile, as the other instantiations do; the arena's probes ask
    // for that too: they classify memory by the undeferred effect), the parked tiles so far, the first one's first row
    const uint32_t tile_words = iters * (WAVE_ROWS / 32);
    const uint32_t defer_tiles = (kDefer && a.live_packs == nullptr && (a.flags & 1u) == 0) ? kStageWords / tile_words : 1u;
    uint32_t parked = 0;
    uint64_t parked_row0 = 0;
    // The flushes are ALIGNED IN TIME across the chip: a phase lasts 2^phase_shift ticks of the 100 MHz real-time clock, the
    // largest power of two below the time the stage takes to fill (a tile is iters * 4 KiB per wave and column whatever the
    // width: ~ gridDim.x * iters * 15 / 64 ticks per tile and leaf at 7 TB/s, 240 for 512 workgroups at 2 iterations; 120 such
    // tiles -> 2^14 ticks = 164 us).  (Writing the per-tile counts in the same phases instead of a 128-byte line every 32
    // rounds was tried: no difference.)
    // Measured on identical buffers (tools/ab_defer.py, phases of 2^11 ... 2^14 ticks): aligned flushes take a bitset in the
    // column's own placement class from 0.846 (unaligned, 120 tiles) to 0.862 of peak, another class stays at 0.88; shorter
    // phases lose (2^11: 0.844 / 0.872).  flags bit1 (KNOX_DEFER=2): unaligned, for the A/B.
    uint32_t phase_shift = 0;
    if (defer_tiles > 1u && (a.flags & 2u) == 0) {
        const uint32_t fill_ticks = defer_tiles * ((gridDim.x * (uint32_t)NL * iters * 15u) >> 6);
        phase_shift = fill_ticks >= 64u ? 31u - (uint32_t)__builtin_clz(fill_ticks) : 0u;
    }
    uint32_t last_phase = phase_shift ? (uint32_t)(__builtin_amdgcn_s_memrealtime() >> phase_shift) : 0u;

    U lo[NL], diff[NL];
    bool isf[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        lo[l] = (U)a.leaf[l].lo;
        diff[l] = (U)a.leaf[l].diff;
        isf[l] = a.leaf[l].is_float != 0;
    }
    const uint32_t tt = (uint32_t)a.tt;  // <= 16 table entries for <= 4 leaves
    // single leaf: q = in-range bits, then (q & m1) | (~q & m0) applies hit / negated / constant
    const uint32_t m1 = (tt & 2u) ? QMASK : 0u, m0 = (tt & 1u) ? QMASK : 0u;
    Swar<(W <= 2 ? W : 1)> sw;
    if constexpr (W <= 2 && NL == 1) sw.init((uint32_t)lo[0], (uint32_t)diff[0], (tt & 2u) != 0, (tt & 1u) != 0);
    // narrow_load's constants: the low end replicated the way the subtract wants it, the width clamped to the element range
    uint32_t nlo = 0, ndiff = 0;
    if constexpr (W <= 2 && NL == 1) {
        nlo = W == 2 ? (((uint32_t)lo[0] & 0xffffu) * 0x00010001u) : ((uint32_t)lo[0] & 0xffu);
        ndiff = (uint32_t)(diff[0] > (U)WT<W>::MASK ? (U)WT<W>::MASK : diff[0]);
    }
    const int qshift = K * (lane & (G - 1));
    // order-key range the selected values of a plain single-leaf range predicate can take (narrow integer types): the
    // leaf's [lo, lo + diff] in key space when it does not wrap, the whole type otherwise
    uint64_t mm_lo = 0, mm_hi = (uint64_t)WT<W>::MASK;
    if constexpr (W <= 2 && NL == 1 && FLV != FL_FLOAT) {
        if (tt == 2u) {
            const uint64_t klo = (uint64_t)((lo[0] ^ (FLV == FL_SIGNED ? WT<W>::SIGN : (U)0)) & WT<W>::MASK);
            if (klo + (uint64_t)diff[0] <= (uint64_t)WT<W>::MASK) { mm_lo = klo; mm_hi = klo + (uint64_t)diff[0]; }
        }
    }

    uint32_t lb = blockIdx.x;
#ifdef KNOX_DIAG
    // diagnostics (debug bit3): XCD-contiguous mapping -- workgroups are dealt round-robin over the 8
    // XCDs, so logical block (b%8)*(G/8) + b/8 gives every XCD one contiguous run of tiles per round
    if ((a.debug & 8u) && (gridDim.x & 7u) == 0) lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    // per-wave-tile popcounts, laid out [workgroup][wave][round] so that a wave's counts are contiguous
    uint32_t *tcbase = a.tile_counts ? a.tile_counts + ((uint64_t)lb * kWavesPerBlock + (uint64_t)wave) * a.tc_rounds : nullptr;
    uint32_t tcreg = 0, round = 0, cnt_prev = 0;
    // packs pruned by the caller's statistics are never touched: with a live-pack list the virtual tile index walks
    // only the surviving packs (their bitset ranges were zeroed by the host).  A wave owns iters*WAVE_ROWS consecutive
    // rows of its tile (one contiguous bitset burst): the first of them
    auto wave_row0 = [&](uint64_t vt) -> uint64_t {
        uint64_t tile = vt;
        if (a.live_packs != nullptr) tile = ((uint64_t)a.live_packs[vt >> a.tpp_shift] << a.tpp_shift) + (vt & ((1ull << a.tpp_shift) - 1));
        return (tile * kWavesPerBlock + (uint64_t)wave) * iters * WAVE_ROWS;
    };
    // the kUnroll 16-byte loads per column of one wave iteration
    auto load_iter = [&](u32x4 (&r)[NL][kUnroll], u32x4 (&ra)[kUnroll], uint64_t row0) {
        const uint64_t vec0 = row0 / K + lane;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const u32x4 *p = (const u32x4 *)a.leaf[l].col + vec0;
#pragma unroll
#ifdef KNOX_LOAD_POLICY
            for (int u = 0; u < kUnroll; u++) r[l][u] = ld16_policy(p + u * 64);
#else
            for (int u = 0; u < kUnroll; u++) r[l][u] = ld16<true>(p + u * 64);
#endif
        }
#ifdef KNOX_LOAD_POLICY
        static_assert(kUnroll == 4, "ld_fence takes 4 registers");
#pragma unroll
        for (int l = 0; l < NL; l++) ld_fence(r[l][0], r[l][1], r[l][2], r[l][3]);
#endif
        if constexpr (SEP && AGG != 0) {
            const u32x4 *p = (const u32x4 *)a.agg_col + vec0;
#pragma unroll
            for (int u = 0; u < kUnroll; u++) ra[u] = ld16<true>(p + u * 64);
        }
    };
    // the parked tiles' words -> the bitset: tile j of the batch lies j rounds behind the first (live_packs == nullptr:
    // a round is gridDim.x tiles further in the table); 16 bytes per lane, whole 128-byte lines, streaming (nt) stores
    auto flush = [&]() {
        const uint32_t nv = tile_words / 4;                     // 16-byte vectors per tile
        const uint32_t total = parked * nv;
        const uint64_t round_words = ((uint64_t)gridDim.x * kWavesPerBlock * iters * WAVE_ROWS) >> 5;
        uint32_t *base = a.bits + (parked_row0 >> 5);
        for (uint32_t v = lane; v < total; v += 64) {
            const uint32_t j = defer_tiles == 1u ? 0u : v / nv;
            const u32x4 x = ((const u32x4 *)stage[wave])[v];
            u32x4 *dst = (u32x4 *)(base + (uint64_t)j * round_words) + (v - j * nv);
#ifdef KNOX_DIAG
            // diagnostics build only (make diag): debug bit0 = words are assembled but never leave the CU
            if (a.debug & 2u) *dst = x;  // plain (L2 write-back) store
            else if (a.debug & 4u) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(x) : "memory");
            else if (!(a.debug & 1u) || x.x == 0x12345678u) __builtin_nontemporal_store(x, dst);
#else
            __builtin_nontemporal_store(x, dst);
#endif
        }
        parked = 0;
    };
    // one wave iteration over loaded registers: predicate, bitset words into the LDS stage, aggregates; the last
    // iteration of a tile writes the wave's burst
    auto body = [&](const u32x4 (&r)[NL][kUnroll], const u32x4 (&ra)[kUnroll], uint64_t wrow0, uint32_t it) {
        const uint64_t row0 = wrow0 + (uint64_t)it * WAVE_ROWS;  // wave-uniform
        uint32_t q[kUnroll];
        constexpr bool kSwar = W <= 2 && NL == 1 && !SEP && (AGG == 0 || AGG == A_ISUM);
        constexpr bool kSwarAgg = W <= 2 && NL == 1 && !SEP && !kSwar && FLV != FL_FLOAT;  // every other aggregate class of the narrow types
        if constexpr (kSwarAgg) {
            constexpr int WS = W <= 2 ? W : 1;
            constexpr unsigned FS = FLV == FL_FLOAT ? FL_UNSIGNED : FLV;
            // Every selected value lies inside the predicate's own range, so once the wave holds a value AT both ends of
            // that range (strict compares: first occurrence wins) no later row can change its min / max: the min / max
            // work is dropped for the rest of the wave's rows (wave-uniform branch; low-cardinality columns get there fast)
#ifdef KNOX_NARROW_SWAR  // A/B builds: the SWAR form
            if ((AGG & A_MINMAX) != 0 && wm.smin <= mm_lo && wm.smax >= mm_hi)
                agg_iter_swar<WS, (AGG & ~A_MINMAX), FS>(acc, wm, fl, sw, r[0], q, cnt32, row0, a.row_base, lane);
            else
                agg_iter_swar<WS, AGG, FS>(acc, wm, fl, sw, r[0], q, cnt32, row0, a.row_base, lane);
#else
            if ((AGG & A_MINMAX) != 0 && wm.smin <= mm_lo && wm.smax >= mm_hi)
                agg_iter_narrow<WS, (AGG & ~A_MINMAX), FS>(acc, wm, fl, nlo, ndiff, m1, m0, r[0], q, cnt32, row0, a.row_base, lane);
            else
                agg_iter_narrow<WS, AGG, FS>(acc, wm, fl, nlo, ndiff, m1, m0, r[0], q, cnt32, row0, a.row_base, lane);
#endif
        } else if constexpr (kSwar) {
            uint32_t sum32 = 0;  // <= 64 values x 255 resp. 32 x 65535 per iteration
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
#ifdef KNOX_NARROW_SWAR  // A/B builds: the SWAR form
                q[u] = swar_load<W, AGG == A_ISUM>(sw, r[0][u], sum32);
#else
                q[u] = narrow_load<W, AGG == A_ISUM>(r[0][u], nlo, ndiff, m1, m0, sum32);
#endif
                cnt32 += __builtin_popcount(q[u]);
            }
            if constexpr (AGG == A_ISUM) acc.isum += sum32;
        } else {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            uint32_t qq = 0;
            if constexpr (NL == 1 && W >= 4 && (AGG & ~A_ISUM) != 0) {
                // VALU-bound instantiations: the compare's carry shifts straight into the word, qq = 2*qq + (d <= diff),
                // highest element first (one v_addc instead of a select + or per value)
#pragma unroll
                for (int j = K - 1; j >= 0; j--) {
                    U x = elem<W>(r[0][u], j);
                    if constexpr (ANYF) x = fkey<W>(x);
                    const U d = (U)(x - lo[0]);
                    if constexpr (W == 4) asm("v_cmp_ge_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(qq) : "v"(d), "s"(diff[0]) : "vcc");
                    else asm("v_cmp_ge_u64 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(qq) : "v"(d), "s"(diff[0]) : "vcc");
                }
            } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
                if constexpr (NL == 1) {
                    U x = elem<W>(r[0][u], j);
                    if constexpr (ANYF && W >= 4) x = fkey<W>(x);  // one leaf and ANYF: it is the float one
                    qq |= ((((U)(x - lo[0])) & WT<W>::MASK) <= diff[0] ? 1u : 0u) << j;
                } else {
                    uint32_t idx = 0;
#pragma unroll
                    for (int l = 0; l < NL; l++) {
                        U x = elem<W>(r[l][u], j);
                        if constexpr (ANYF && W >= 4) x = isf[l] ? fkey<W>(x) : x;
                        idx |= ((((U)(x - lo[l])) & WT<W>::MASK) <= diff[l] ? 1u : 0u) << l;
                    }
                    qq |= ((tt >> idx) & 1u) << j;
                }
            }
            }
            if constexpr (NL == 1) qq = (qq & m1) | (~qq & m0);
            q[u] = qq;
            cnt32 += __builtin_popcount(qq);
        }
        }  // !kSwar
        if (a.bits != nullptr) {
            // assemble the 32-bit words (4 independent DPP chains), stage them in LDS so the wave's
            // bitset leaves as ONE 16-byte-per-lane store
            uint32_t w[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) w[u] = q[u] << qshift;
            if constexpr (G >= 2) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0xB1>(w[u]);   // quad_perm [1,0,3,2]
            }
            if constexpr (G >= 4) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0x4E>(w[u]);   // quad_perm [2,3,0,1]
            }
            if constexpr (G >= 8) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0x141>(w[u]);  // row_half_mirror
            }
            if constexpr (G >= 16) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) w[u] |= dpp_mov<0x140>(w[u]);  // row_mirror
            }
            if ((lane & (G - 1)) == 0) {
#pragma unroll
                for (int u = 0; u < kUnroll; u++) stage[wave][parked * tile_words + it * (WAVE_ROWS / 32) + u * (64 / G) + (lane / G)] = w[u];
            }
        }
        if constexpr (AGG != 0 && !kSwar && !kSwarAgg) {
            if constexpr (SEP) agg_iter<W, AGG, FLV>(acc, wm, fl, ra, q, row0, a.row_base, lane);
            else agg_iter<W, AGG, FLV>(acc, wm, fl, r[0], q, row0, a.row_base, lane);
        }
        if (it + 1 == iters) {
        if (a.bits != nullptr) {
            // LDS ops of one wave execute in order: the flush reads see the words written above.
            // One burst of iters * WAVE_ROWS/8 contiguous bytes per wave and tile (128 B for 8-byte types at 2 iterations):
            // whole 128-B lines, 16-byte-per-lane streaming (nt) stores -- now, or with the tiles parked behind it
            if (parked == 0) parked_row0 = wrow0;
            ++parked;
            if (phase_shift != 0u) {
                // every wave of every workgroup flushes when the constant 100 MHz clock enters a new phase: the write-only
                // phases of the whole chip coincide (a full stage flushes too, which the phase length is chosen to avoid)
                const uint32_t ph = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> phase_shift);
                if (parked == defer_tiles || ph != last_phase) flush();
                last_phase = ph;
            } else if (parked == defer_tiles) {
                flush();
            }
            // this lane's share of the burst's popcount is what its running count grew by (the wave total is all
            // the per-tile count needs): no popcounts over the staged words
            uint32_t tc = cnt32 - cnt_prev;
            cnt_prev = cnt32;
            if (tcbase != nullptr) {
                // popcount of this wave tile's burst, once per tile.  The counts of 32 consecutive
                // rounds are parked one per lane and leave as ONE full 128-byte line (no partial writes).
                // (DPP row butterfly + 4 readlanes: a ds_bpermute shuffle chain here costs ~700 cycles of
                // exposed latency per tile, 5 % of the kernel at 2 workgroups/CU)
                tc += dpp_mov<0xB1>(tc);
                tc += dpp_mov<0x4E>(tc);
                tc += dpp_mov<0x141>(tc);
                tc += dpp_mov<0x140>(tc);
                const uint32_t total = (uint32_t)(__builtin_amdgcn_readlane((int)tc, 0) + __builtin_amdgcn_readlane((int)tc, 16) +
                                                  __builtin_amdgcn_readlane((int)tc, 32) + __builtin_amdgcn_readlane((int)tc, 48));
                tcreg = lane == (int)(round & 31u) ? total : tcreg;
                if ((round & 31u) == 31u && lane < 32) tcbase[(round & ~31u) + lane] = tcreg;
                round++;
            }
        }
        }
    };
    // Aggregate-heavy single-column scans are software pipelined: the loads of the NEXT wave iteration (the next tile's
    // first one at a tile's end) are in flight while the current one is evaluated -- two register sets, roles fixed by an
    // even iteration count, no register moves.  With 2-3 workgroups per CU resident, waiting out the full HBM latency
    // once per iteration is what these instantiations were bound by (measured: float32 +4 %, int16 ALL +7 %, int32 ALL
    // +2 %).  The plain filter / filter+sum kernels do so little between loads that the extra loads in flight only
    // disturb the read / write mix at the memory controllers (int64 headline -10 %): they keep the simple loop.
    // (round 3 A/B, the plain 1- / 2-byte scans pipelined too: int16 -1.4 ... -2.7 %, uint8 +1.5 % -- left as they are)
    // (round 3: two-leaf aggregate-heavy scans too -- BASELINE configs[2]: +1.4 % on identical buffers, tools/ab_kernels.py)
    // (round 3, with the bursts deferred -- tools/ab_kernels.py against a build with every plain scan pipelined: two-leaf plain
    //  trees +1.4 ... +1.9 %, uint8 filter + sum +5.5 %; int64 filter + sum -4.2 %, int16 -3 ... -7 %, int32 -1.5 %, uint8 count 0)
#ifdef KNOX_PIPE_PLAIN  // experiment builds: every plain scan pipelined
    constexpr bool kPipe = NL <= 2 && !SEP;
#else
    // (round 4: NOT the two-leaf scans of 1-byte columns -- two register sets of two leaves at 16 rows per load need 256 VGPRs,
    //  one workgroup per CU where the launch shape assumes two or three; found by the build-time occupancy check,
    //  tests/test_scan_shapes.py)
    constexpr bool kPipe = NL <= 2 && !SEP && !(W == 1 && NL == 2) && ((AGG & ~A_ISUM) != 0 || NL == 2 || (W == 1 && AGG == A_ISUM));
#endif
    if (kPipe && (iters & 1u) == 0) {
        u32x4 ra[kUnroll];
        u32x4 ra_a[NL][kUnroll], rb[NL][kUnroll];
        uint64_t vt = lb;
        if (vt < a.n_tiles) {
            uint64_t w0 = wave_row0(vt);
            load_iter(ra_a, ra, w0);
            for (;;) {
                const uint64_t vn = vt + gridDim.x;
                const bool more = vn < a.n_tiles;  // wave-uniform
                const uint64_t wn = more ? wave_row0(vn) : 0;
#pragma unroll 1
                for (uint32_t it = 0; it < iters; it += 2) {
                    load_iter(rb, ra, w0 + (uint64_t)(it + 1) * WAVE_ROWS);
                    body(ra_a, ra, w0, it);
                    // always issued (at the very end: a reload of this tile's first rows, never used), so the waits below
                    // count on a fixed number of loads in flight and only wait for the older set
                    load_iter(ra_a, ra, it + 2 < iters ? w0 + (uint64_t)(it + 2) * WAVE_ROWS : more ? wn : w0);
                    body(rb, ra, w0, it + 1);
                }
                if (!more) break;
                vt = vn;
                w0 = wn;
            }
        }
    } else {
        for (uint64_t vt = lb; vt < a.n_tiles; vt += gridDim.x) {
            const uint64_t w0 = wave_row0(vt);
#pragma unroll 1
            for (uint32_t it = 0; it < iters; it++) {
                u32x4 r[NL][kUnroll];
                u32x4 ra[kUnroll];
                load_iter(r, ra, w0 + (uint64_t)it * WAVE_ROWS);
                body(r, ra, w0, it);
            }
        }
    }
    if (a.bits != nullptr && parked != 0) flush();
    if (tcbase != nullptr && (round & 31u) != 0 && lane < (int)(round & 31u)) tcbase[(round & ~31u) + lane] = tcreg;
    acc.count = cnt32;
    if constexpr ((AGG & A_FIRSTLAST) != 0) fl.template finish<K>(acc, a.row_base, lane);
    block_reduce_store(acc, a.partials + blockIdx.x);
}

// Once per instantiation that defers its bursts: does the runtime agree that scan_shape's workgroups per CU fit?  (Registers,
// static LDS and launch bounds decide; a change to either side of the table, or a build for a part with less LDS, would
// otherwise only show as a measured loss: the chip-wide phases of the deferral need the whole grid resident.)  A mismatch
// is counted (knox_scan_occupancy_mismatches, asserted 0 by tests/test_gpu_defer.py) and reported once on stderr.
template <int W, int NL, unsigned AGG, bool SEP, bool ANYF, unsigned FLV>
static void scan_check_occupancy() {
    constexpr ScanShape sh = scan_shape(W, NL, (AGG & ~A_ISUM) != 0, FLV == FL_FLOAT, (AGG & A_FSUM) != 0);
    if constexpr (sh.park_words != 0) {
        static const bool once = [] {
            int fit = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, scan_kernel<W, NL, AGG, SEP, ANYF, FLV>, kBlock, 0) != hipSuccess) {
                (void)hipGetLastError();
                return true;
            }
            if (fit < sh.per_cu) {
                g_scan_occupancy_mismatches.fetch_add(1);
                fprintf(stderr, "knoxgpu: scan_kernel<%d,%d,%u,%d,%d,%u> fits %d workgroups per CU, its launch shape assumes %d\n", W, NL, AGG,
                        (int)SEP, (int)ANYF, FLV, fit, sh.per_cu);
            }
            return true;
        }();
        (void)once;
    }
}

template <int W, int NL, unsigned AGG, bool SEP, unsigned FLV>
static hipError_t scan_anyf(hipStream_t s, int grid, const ScanArgs &a) {
    bool anyf = false;
    for (int l = 0; l < NL; l++) anyf = anyf || a.leaf[l].is_float != 0;
    if constexpr (W >= 4) {
        if (anyf) {
            scan_check_occupancy<W, NL, AGG, SEP, true, FLV>();
            KNOX_LAUNCH((scan_kernel<W, NL, AGG, SEP, true, FLV>), grid, a);
        }
    }
    scan_check_occupancy<W, NL, AGG, SEP, false, FLV>();
    KNOX_LAUNCH((scan_kernel<W, NL, AGG, SEP, false, FLV>), grid, a);
}

template <int W, int NL, unsigned AGG, unsigned FLV>
static hipError_t scan_sep(hipStream_t s, bool sep, int grid, const ScanArgs &a) {
    return sep ? scan_anyf<W, NL, AGG, true, FLV>(s, grid, a) : scan_anyf<W, NL, AGG, false, FLV>(s, grid, a);
}

template <int W, int NL>
static hipError_t scan_agg(hipStream_t s, unsigned agg, bool sep, int grid, const ScanArgs &a) {
    // aggregate classes instantiated: none | isum | full-without-isum (per flavour) | all (ints)
    constexpr unsigned FF = A_FSUM | A_MINMAX | A_FIRSTLAST;
    if (agg == 0) return scan_anyf<W, NL, 0, false, FL_SIGNED>(s, grid, a);
    if (agg == A_ISUM) return scan_sep<W, NL, A_ISUM, FL_SIGNED>(s, sep, grid, a);
    if constexpr (W >= 4) {
        if (a.agg_flavor == FL_FLOAT) return scan_sep<W, NL, FF, FL_FLOAT>(s, sep, grid, a);
    }
    if ((agg & A_ISUM) == 0) {
        return a.agg_flavor == FL_SIGNED ? scan_sep<W, NL, FF, FL_SIGNED>(s, sep, grid, a) : scan_sep<W, NL, FF, FL_UNSIGNED>(s, sep, grid, a);
    }
    return a.agg_flavor == FL_SIGNED ? scan_sep<W, NL, A_ALL, FL_SIGNED>(s, sep, grid, a) : scan_sep<W, NL, A_ALL, FL_UNSIGNED>(s, sep, grid, a);
}

template <int W>
hipError_t scan_nl(hipStream_t s, int nl, unsigned agg, bool sep, int grid, const ScanArgs &a) {
    switch (nl) {
    case 1: return scan_agg<W, 1>(s, agg, sep, grid, a);
    case 2: return scan_agg<W, 2>(s, agg, sep, grid, a);
    case 3: return scan_agg<W, 3>(s, agg, sep, grid, a);
    case 4: return scan_agg<W, 4>(s, agg, sep, grid, a);
    default: return hipErrorInvalidValue;
    }
}

template hipError_t scan_nl<KNOX_SCAN_W>(hipStream_t s, int nl, unsigned agg, bool sep, int grid, const ScanArgs &a);

#if KNOX_SCAN_W == 8
// The allocator's placement probe (knox_abi.cpp): the headline code path under its own symbol
// (FLV is irrelevant to an integer sum), so that profiles keep probe launches and real scans apart.
hipError_t launch_probe_scan(hipStream_t s, int grid, const ScanArgs &a) {
    if (grid <= 0) return hipSuccess;
    KNOX_LAUNCH((scan_kernel<8, 1, A_ISUM, false, false, FL_UNSIGNED>), grid, a);
}
#endif

}  // namespace knox
