// knox_arena.h -- one bounded device-memory arena per context (host code, reached through knox_internal.h).
//
// WHY.  Measured on MI355X (tools/placement_vmm.py, placement_census.py, placement_finemap.py): for the
// scan's read+write mix (8 B read, 1/8 B written per row) it matters WHERE the bitset lies relative to the column.
// Physical HBM falls into a few classes -- long runs of the driver's allocation order, uniform over >= 1 GiB -- and a
// scan whose column and bitset lie in the SAME class runs at ~6.0 TB/s, one whose column and bitset lie in DIFFERENT
// classes at ~7.05 TB/s (symmetric; independent of the read/write address ratio, of alignment shifts, of VMM vs
// hipMalloc mappings; read-only passes do not see it; streams below ~1 GiB do not show it).
//
// How fine-grained the classes are depends on where the memory comes from (GiB-uniform in the long free run the big
// chunks come out of; down to ~100 MiB among the driver's leftover fragments, which small allocations are served from),
// so no label attached to a region is trusted: the ONLY thing that is ever decided is one concrete (column, write
// buffer) pair, and it is decided by running the scan kernel on that very pair.
//
// WHAT.  Physical memory is taken from the driver only as allocations need it, up to an explicit budget, and is not
// returned before the context closes (or an explicit trim).  Columns (>= 1 GiB) live in ZONES, one per placement class:
// every 1 GiB physical chunk (hipMemCreate) is probed against the reference window of each class met so far ("is a
// 1 GiB column here slow with it?": slow = that window's class; fast with all of them = a class of its own, whose
// reference window becomes the first 32 MiB of this very chunk) and mapped at the end of its class's zone, so a zone is
// one VA-contiguous heap of ONE class of memory, whatever order the driver handed the chunks out in, and a write buffer
// that is fast with one part of a column is fast with all of it.  Smaller things share an ordinary heap.  A large write
// buffer that is allocated FOR a column (knox_bits_alloc_for, the scan's scratch bitsets) is chosen among candidates:
// spare write buffers first, then a few new physical allocations of their own (small allocations are served from the
// driver's leftover fragments, which lie all over the device: usually one of a handful is in another class), then a
// range at the TOP of another class's zone (columns fill zones from the bottom) -- taking chunks, which stay as column
// space, for a few chunks if no other class has been met.  Every candidate is VERIFIED: up to sixteen 1 GiB samples of
// the column are scanned into the part of the candidate their bits really go to; the first candidate whose worst sample
// runs at >= 0.95 of its read-only form is taken.  Rejected candidates stay in the arena as spare write buffers (the
// first candidates of the next request: what is slow with one column is fast with a column of another class); nothing
// is held outside the accounting and nothing is handed back to the driver during a search (the driver clears returned
// memory, which costs the whole device bandwidth for a second or two).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

namespace knox {

enum ArenaKind { ARENA_READ = 0, ARENA_WRITE = 1 };

// scan / read-only: the rate of the scan over `rows` int64 rows at `col` into the bitset at `bits`, divided by the rate of the
// read-only form of the same scan, measured with ALTERNATING launches (whatever else disturbs the device -- the driver
// clearing freshly taken or released memory costs bandwidth for seconds -- hits both alike); <= 0 = failed
typedef double (*ArenaProbeFn)(void *user, const void *col, uint64_t rows, void *bits);
// the same region against n write buffers in ONE short launch sequence (labelling a chunk once >= 2 classes are known): 0 = ok
typedef int (*ArenaProbeMultiFn)(void *user, const void *col, uint64_t rows, void *const *bits, int n, double *ratios);

// Three facts about the HIP virtual-memory API on this stack (ROCm 7.2), all measured (tools/vmm_remap_check*.py):
//  1. a virtual address that has been mapped to one physical allocation must never be mapped to another: after
//     hipMemUnmap(va) + hipMemMap(va, another handle) kernels that write through `va` still reach the FIRST handle's
//     memory -- also when the reservation was freed (hipMemAddressFree) and the same address reserved again;
//  2. physical memory only goes back to the driver at hipMemAddressFree of the reservation it was mapped in -- unmapping
//     and releasing the handle is not enough;
//  3. hipMemAddressReserve honours its address hint.
// So: every mapping unit (a heap chunk, a write buffer, the probe mapping of a new chunk, a zone's whole region) gets a
// reservation of its own at an address NO reservation of this process has had before -- a process-wide cursor that only
// moves up hands the hints out -- mappings are only ever added inside a zone's region, and a reservation is freed as
// soon as nothing is mapped in it any more (the memory goes back, the address is not used again).
class VaBump {
  public:
    static char *take(size_t bytes) {
        static std::mutex mu;
        static uintptr_t cursor = (uintptr_t)0x100000000000ull;  // 16 TiB: far above what the runtime hands out by itself
        const size_t gib = (size_t)1 << 30;
        const size_t span = ((bytes + gib - 1) & ~(gib - 1)) + gib;  // 1 GiB aligned (large page-table fragments) + a guard
        std::lock_guard<std::mutex> lk(mu);
        for (int attempt = 0; attempt < 8 && cursor + span < (uintptr_t)0x700000000000ull; attempt++) {
            void *p = nullptr;
            const uintptr_t hint = cursor;
            cursor += span;
            if (hipMemAddressReserve(&p, bytes, gib, (void *)hint, 0) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
            if ((uintptr_t)p == hint) return (char *)p;
            (void)hipMemAddressFree(p, bytes);  // not where it was asked for: not an address this process controls
        }
        return nullptr;
    }
    static char *take_own(size_t bytes) { return take(bytes); }
    static void give_back(char *va, size_t bytes) {
        if (va) (void)hipMemAddressFree(va, bytes);
    }
};

class Arena {
  public:
    static constexpr size_t kChunk = (size_t)1 << 30;          // physical unit of columns = one labelling probe
    static constexpr size_t kSmallWrite = (size_t)16 << 20;    // write buffers below this share the heap (streams that short do not show the effect)
    static constexpr uint64_t kProbeRows = (uint64_t)1 << 27;  // one sample: 1 GiB of int64 rows, 16 MiB of bitset
    static constexpr double kFastAbove = 0.95;                 // scan / read-only: ~0.88 same class, ~0.98 different classes
    static constexpr int kNearCandidates = 6;                  // new write buffers looked at before reaching further
    static constexpr int kMaxClasses = 4;                      // placement classes told apart (three were ever seen on one device)
    static constexpr size_t kRefBytes = (size_t)32 << 20;      // a class's reference window
    static constexpr int kMaxGrowForWrite = 96;                // chunks taken at most while looking for memory of another class (round 2: 8; the
                                                               // driver's free memory comes in runs of one class: a profiled bench run that found none
                                                               // within 8 ran the headline scan at 0.72 instead of 0.87 of peak, and so did one that
                                                               // found none within 24 -- its run was > 40 GiB long; a chunk costs ~10 ms to take and label)
    static constexpr size_t kMaxSpareBytes = (size_t)6 << 30;  // spare write buffers kept (beyond: returned to the driver; they are small)

    bool usable = false;
    size_t budget = 0, mapped_total = 0, peak_used = 0;
    int chunks_created = 0, probes_run = 0, write_buffers = 0;
    size_t mixed_bytes = 0;  // chunks moved out of their class zones into the mixed region (columns no single class could hold)

    void configure(int device_, size_t budget_, ArenaProbeFn fn, void *user_, ArenaProbeMultiFn multi_ = nullptr,
                   const int *spent_launches_ = nullptr, const double *spent_ms_ = nullptr) {
        device = device_;
        budget = budget_;
        probe = fn;
        probe_multi = multi_;
        user = user_;
        spent_launches = spent_launches_;
        spent_ms = spent_ms_;
    }
    // Probe budget (round 4): what the arena may spend on measurements over the life of its context -- kernel launches and
    // device time, as the context counts them (knox_arena_info.probe_launches / probe_ms).  Placement buys 1-6 points of HBM
    // peak on multi-GiB scans (bench.py "placement_off"); it must not cost a context more than a few tens of milliseconds.
    // Once the budget is spent nothing is measured any more: a new chunk inherits the label of the chunk before it (the
    // driver hands memory out in runs of 5-40 GiB of one class) and a write buffer is taken from another class's zone by
    // label alone.  knox_ctx_set_probe_budget grants a new budget (a host does so when it has idle time to spend).
    int budget_launches = 200;
    double budget_ms = 50.0;
    int spent_base_launches = 0;
    double spent_base_ms = 0.0;
    void grant(int launches, double ms) {
        budget_launches = launches;
        budget_ms = ms;
        spent_base_launches = spent_launches ? *spent_launches : 0;
        spent_base_ms = spent_ms ? *spent_ms : 0.0;
    }
    bool may_probe() const {
        if (!probe) return false;
        if (!spent_launches || !spent_ms) return true;
        return *spent_launches - spent_base_launches < budget_launches && *spent_ms - spent_base_ms < budget_ms;
    }
    int unprobed_chunks = 0, unverified_write_buffers = 0;  // decisions taken by inheritance / by label after the budget ran out
    size_t used_bytes() const { return used; }
    size_t spare_bytes() const {
        size_t n = 0;
        for (auto &w : spares) n += w.bytes;
        return n;
    }
    size_t zone_bytes(int label) const { return label >= 1 && label <= kMaxClasses ? zones[label - 1].mapped : 0; }
    int classes_seen() const { return n_refs; }
    int class_of(const void *p) const { return zone_of(p); }  // the placement class (zone index) an address lies in, -1 = heap / mixed region / not the arena's

    // Columns / payloads of >= 1 GiB: a range of ONE zone (chunks of one label, mapped back to back).  Everything else: the heap.
    hipError_t alloc(void **out, size_t bytes, int kind) {
        *out = nullptr;
        bytes = round_up(bytes);
        if (!init_once()) return hipErrorNotSupported;
        (void)kind;  // large buffers of either kind live in a zone; write buffers that matter are verified (alloc_write_for)
        if (bytes >= kChunk) return zone_alloc(out, bytes);
        return heap_alloc(out, bytes);
    }

    // Room for columns that will be scanned together: grows the arena until ONE class zone has `bytes` contiguous free and
    // makes that zone the preferred one of the next large allocations (zone_alloc), so that e.g. both columns of a two-leaf
    // scan, or the pointer array and the pool of a bytes column, land in one placement class.  Nothing is allocated.
    hipError_t reserve(size_t bytes) {
        bytes = round_up(bytes);
        if (!init_once()) return hipErrorNotSupported;
        if (bytes < kChunk) return hipSuccess;
        for (;;) {
            for (int k = 0; k <= kMaxClasses; k++) {
                const int zi = k == 0 ? last_zone : k - 1;
                if (zi < 0 || (k > 0 && zi == last_zone)) continue;
                for (auto &e : zones[zi].extents)
                    for (auto &fr : e.free)
                        if (fr.second >= bytes) {
                            last_zone = zi;
                            dbg("reserve: zone %d has %zu MiB contiguous for a request of %zu MiB", zi, fr.second >> 20, bytes >> 20);
                            return hipSuccess;
                        }
            }
            if (mapped_total + kChunk > budget || grow_zone() < 0) return hipErrorOutOfMemory;
        }
    }

    // A write buffer FOR a column (or for the columns of one multi-leaf scan): a physical allocation of its own, verified
    // against every one of them (see the header comment).  *ratio_out: scan / read-only of the worst sample with the buffer
    // that was taken (0 = not measured); *tried_out: candidates looked at.  bytes_per_row: column bytes per bit of the buffer.
    struct ColRef {
        const void *p;
        size_t bytes;
        double bytes_per_row;
    };
    hipError_t alloc_write_for(void **out, size_t bytes, const void *col, size_t col_bytes, double bytes_per_row, double *ratio_out,
                               int *tried_out) {
        const ColRef c{col, col_bytes, bytes_per_row};
        return alloc_write_for(out, bytes, &c, 1, ratio_out, tried_out);
    }
    hipError_t alloc_write_for(void **out, size_t bytes, const ColRef *cols, int n_cols, double *ratio_out, int *tried_out) {
        *out = nullptr;
        if (ratio_out) *ratio_out = 0.0;
        if (tried_out) *tried_out = 0;
        bytes = round_up(bytes);
        if (!init_once()) return hipErrorNotSupported;
        const size_t sample = kProbeRows * 8, wrote = kProbeRows / 8;
        constexpr int kMaxSamples = 4;   // up to four 1 GiB samples spread over the column(s) (round 3: 16 -- with the bursts deferred a mixed buffer costs 2 %, not 16)
        struct Sample { const char *col; size_t boff; };
        Sample samples[kMaxSamples];
        int n_s = 0;
        bool in_zone[kMaxClasses] = {false, false, false, false};
        bool any_zone = false;
        if (probe && bytes >= std::max(kSmallWrite, wrote + 64)) {
            int usable_cols = 0;
            for (int c = 0; c < n_cols; c++) usable_cols += cols[c].p && cols[c].bytes >= sample && cols[c].bytes_per_row > 0.0;
            for (int c = 0; c < n_cols && usable_cols; c++) {
                if (!cols[c].p || cols[c].bytes < sample || cols[c].bytes_per_row <= 0.0) continue;
                const int want = (int)std::min<size_t>((size_t)std::max(2, kMaxSamples / usable_cols), cols[c].bytes / sample);
                for (int i = 0; i < want && n_s < kMaxSamples; i++) {
                    const size_t off = want > 1 ? (((cols[c].bytes - sample) / (size_t)(want - 1) * (size_t)i) & ~(size_t)0xffff) : 0;
                    size_t boff = ((size_t)((double)off / cols[c].bytes_per_row) >> 3) & ~(size_t)255;
                    if (boff + wrote + 64 > bytes) boff = (bytes - wrote - 64) & ~(size_t)255;
                    samples[n_s++] = Sample{(const char *)cols[c].p + off, boff};
                }
                const int z = zone_of(cols[c].p);
                if (z >= 0) { in_zone[z] = true; any_zone = true; }
            }
        }
        if (n_s == 0) return alloc(out, bytes, ARENA_WRITE);
        const size_t wb_bytes = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (!may_probe()) {
            // budget spent: the top of a zone none of the columns lies in, by label alone; failing that, anywhere
            for (int zi = 0; zi < kMaxClasses && any_zone; zi++) {
                if (in_zone[zi]) continue;
                char *va = nullptr;
                if (!zone_take_top(zi, wb_bytes, &va)) continue;
                unverified_write_buffers++;
                dbg("write buffer (%zu MiB) at the top of zone %d by label (probe budget spent)", wb_bytes >> 20, zi);
                *out = va;
                return hipSuccess;
            }
            return alloc(out, bytes, ARENA_WRITE);
        }
        auto measure = [&](char *va) {
            double worst = 2.0;
            for (int i = 0; i < n_s; i++) {
                if (i > 0 && !may_probe()) break;  // the budget ran out between samples: what was measured so far decides
                // the short launch sequence first (6 launches); a reading near the threshold is taken again with the long form (20)
                double r = -1.0;
                if (probe_multi) {
                    void *one[1] = {va + samples[i].boff};
                    double qk = 0.0;
                    if (probe_multi(user, samples[i].col, kProbeRows, one, 1, &qk) == 0 && (qk < kFastAbove - 0.03 || qk > kFastAbove + 0.015)) r = qk;
                }
                if (r < 0.0) r = probe(user, samples[i].col, kProbeRows, va + samples[i].boff);
                probes_run++;
                if (r <= 0.0) return -1.0;
                if (r < kFastAbove) dbg("  sample %d of %d: %.3f", i, n_s, r);
                worst = std::min(worst, r);
                if (worst < kFastAbove - 0.03) break;  // out already
            }
            if (tried_out) (*tried_out)++;
            return worst;
        };
        WBuf best;
        double best_r = -1.0;
        auto consider = [&](WBuf &w, double r) {  // keeps the better of (best, w) aside, the other one becomes a spare
            if (r > best_r) {
                if (best.va) add_spare(best);
                best = w;
                best_r = r;
            } else {
                add_spare(w);
            }
        };
        // 0. memory of a KNOWN other class, if some has been met: column chunks are labelled by class (grow_zone), so a range
        //    at the top of another class's zone is fast with this column by construction -- verified like every candidate
        bool zone_failed[kMaxClasses] = {false, false, false, false};
        int zone_best_zi = -1;        // the best zone candidate seen that was NOT fast enough to end the search: still better than
        double zone_best_r = -1.0;    // a slower fresh buffer when nothing fast turns up (round 4: a 0.933 zone was dropped for a 0.875 buffer)
        auto try_zones = [&]() {
            // every other class with room is measured: class pairs differ (0.97 against 1.00 of the read-only rate), the
            // best one is taken; the first at >= 0.99 ends the search
            char *zbest = nullptr;
            double zbest_r = -1.0;
            for (int zi = 0; zi < kMaxClasses && any_zone; zi++) {
                if (in_zone[zi] || zone_failed[zi]) continue;
                char *va = nullptr;
                if (!zone_take_top(zi, wb_bytes, &va)) continue;
                const double r = measure(va);
                dbg("write buffer (%zu MiB) at the top of zone %d vs the column(s): %.3f", wb_bytes >> 20, zi, r);
                if (r < kFastAbove) zone_failed[zi] = true;  // its label says "another class" and the pair is slow all the same: leave that zone alone
                if (r < kFastAbove && r > zone_best_r) { zone_best_r = r; zone_best_zi = zi; }
                if (r > zbest_r) {
                    if (zbest) (void)release(zbest);
                    zbest = va;
                    zbest_r = r;
                } else {
                    (void)release(va);
                }
                if (zbest_r >= kFastAbove) break;  // round 4: the first fast class is taken (class pairs differ by 0.97 against 1.00; measuring them all cost more than it bought)
            }
            if (zbest && zbest_r >= kFastAbove) {
                if (best.va) add_spare(best);
                best.va = nullptr;
                *out = zbest;
                if (ratio_out) *ratio_out = zbest_r;
                return true;
            }
            if (zbest) (void)release(zbest);
            return false;
        };
        if (try_zones()) return hipSuccess;
        // 1. spare write buffers of a fitting size (each at most once)
        {
            std::vector<WBuf> mine;
            for (size_t i = 0; i < spares.size();) {
                if (spares[i].bytes < wb_bytes || spares[i].bytes > 2 * wb_bytes) { i++; continue; }
                mine.push_back(spares[i]);
                spares.erase(spares.begin() + (long)i);
            }
            for (size_t i = 0; i < mine.size(); i++) {
                const double r = measure(mine[i].va);
                dbg("spare write buffer (%zu MiB) vs the column: %.3f", mine[i].bytes >> 20, r);
                if (r >= kFastAbove) {
                    for (size_t j = i + 1; j < mine.size(); j++) add_spare(mine[j]);
                    return take_wbuf(out, mine[i], bytes, r, ratio_out, best);
                }
                consider(mine[i], r);
            }
        }
        // 2. a few new ones
        bool hit = false;
        auto try_new = [&](int n_tries) {
            for (int k = 0; k < n_tries && may_probe(); k++) {
                WBuf w;
                if (!new_wbuf(wb_bytes, &w)) return false;
                const double r = measure(w.va);
                dbg("new write buffer (%zu MiB) vs the column: %.3f", w.bytes >> 20, r);
                if (r >= kFastAbove) {
                    hit = true;
                    (void)take_wbuf(out, w, bytes, r, ratio_out, best);
                    return true;
                }
                consider(w, r);
            }
            return true;
        };
        (void)try_new(kNearCandidates);
        if (hit) return hipSuccess;
        // 3. no other class met yet (or its zone is full): take a few more chunks -- they stay as column space of their
        //    class -- and look again
        // (columns in TWO classes -- a multi-leaf scan -- need memory of a third one: classes come in runs of 5-40 GiB of the
        //  driver's allocation order, so that search may look twice as far)
        int col_classes = 0;
        for (bool z : in_zone) col_classes += z;
        const int max_grow = col_classes >= 2 ? 2 * kMaxGrowForWrite : kMaxGrowForWrite;
        // Labels are a shortcut, not the decision: a chunk whose label says "the column's class" (or whose zone failed above)
        // is still measured against the column directly, once, at the moment it arrives -- a label that a noisy reading got
        // wrong, or a column that is not class pure, must not make the search walk past the memory it is looking for.
        char *zcand = nullptr;
        double zcand_r = -1.0;
        for (int grown = 0; any_zone && may_probe() && grown < max_grow && grown_for_write < 2 * max_grow; grown++) {
            const int zi = grow_zone();
            if (zi < 0) break;
            grown_for_write++;  // over the arena's life: a device whose free memory is one long run of one class is not mapped chunk by chunk in vain
            if (!in_zone[zi] && !zone_failed[zi]) {
                if (try_zones()) {
                    if (zcand) (void)release(zcand);
                    return hipSuccess;
                }
                continue;
            }
            char *va = nullptr;
            if (!zone_take_top(zi, wb_bytes, &va)) continue;
            const double r = measure(va);
            dbg("write buffer (%zu MiB) in the chunk just taken (zone %d%s) vs the column(s): %.3f", wb_bytes >> 20, zi,
                in_zone[zi] ? ", the column's own label" : ", a zone that failed before", r);
            if (r >= kFastAbove) {
                if (zcand) (void)release(zcand);
                if (best.va) add_spare(best);
                best.va = nullptr;
                *out = va;
                if (ratio_out) *ratio_out = r;
                return hipSuccess;
            }
            if (r > zcand_r) {
                if (zcand) (void)release(zcand);
                zcand = va;
                zcand_r = r;
            } else {
                (void)release(va);
            }
        }
        if (zcand && zcand_r > best_r) {  // the best there is lies in a zone
            if (best.va) add_spare(best);
            *out = zcand;
            if (ratio_out) *ratio_out = zcand_r;
            dbg("nothing fast within reach: the best candidate (%.3f, in a zone)", zcand_r);
            return hipSuccess;
        }
        if (zcand) (void)release(zcand);
        if (zone_best_zi >= 0 && zone_best_r > best_r) {  // a zone that measured below the bar earlier, but above everything else
            char *va = nullptr;
            if (zone_take_top(zone_best_zi, wb_bytes, &va)) {
                if (best.va) add_spare(best);
                *out = va;
                if (ratio_out) *ratio_out = zone_best_r;
                dbg("nothing fast within reach: the top of zone %d after all (%.3f)", zone_best_zi, zone_best_r);
                return hipSuccess;
            }
        }
        dbg("nothing fast within reach: the best candidate (%.3f)", best_r);
        // Nothing fast within reach (the free memory around is all of the column's class, a run longer than what was
        // looked at): the best candidate there is.  An earlier version then skipped ahead in the driver's allocation order
        // with transient 24 GiB spacers; measured on a box in exactly this state it found nothing in 144 GiB, and handing
        // the spacers back made the driver clear them, which slows the whole device for seconds -- removed.
        if (!best.va) return alloc(out, bytes, ARENA_WRITE);
        WBuf none;
        return take_wbuf(out, best, bytes, best_r, ratio_out, none);  // nothing fast within reach: the best there is
    }

    // true when p was an arena allocation (and is now free)
    bool release(void *p) {
        auto s = smalls.find((char *)p);
        if (s != smalls.end()) {
            Extent &e = s->second.zone < 0 ? heap[(size_t)s->second.extent] : zones[s->second.zone].extents[(size_t)s->second.extent];
            give(e.free, s->second.off, s->second.len);
            used -= s->second.len;
            smalls.erase(s);
            return true;
        }
        auto w = wbufs.find((char *)p);
        if (w == wbufs.end()) return false;
        used -= w->second.used;
        WBuf wb = w->second;
        wbufs.erase(w);
        add_spare(wb);  // kept: the next request of that size tries it first
        return true;
    }

    // returns every zone that holds no allocation and every spare write buffer to the driver (explicit, never automatic).
    // (A zone is one reservation: its memory goes back when the reservation is freed, i.e. all of it or nothing.)
    size_t trim() {
        size_t freed = 0;
        bool any_in_use = false;
        for (auto &kv : smalls) any_in_use = any_in_use || kv.second.zone >= 0;
        if (!any_in_use) {  // the zones of classes 1.. hold their class's reference window: all of them go, or none
            for (auto &z : zones) {
                if (!z.region) continue;
                for (auto &e : z.extents) drop_extent(e);
                for (auto &rg : z.regions) VaBump::give_back(rg.first, rg.second);
                freed += z.mapped;
                mapped_total -= z.mapped;
                z = Zone();
            }
            for (int c = 1; c < kMaxClasses; c++) ref_va[c] = nullptr;
            last_zone = -1;
            if (n_refs > 1) {
                used -= (size_t)(n_refs - 1) * kRefBytes;
                n_refs = 1;
            }
        }
        for (auto &w : spares) {
            freed += w.bytes;
            drop_wbuf(w);
        }
        spares.clear();
        return freed;
    }

    void destroy() {
        for (auto &kv : wbufs) drop_wbuf(kv.second);
        wbufs.clear();
        for (auto &w : spares) drop_wbuf(w);
        spares.clear();
        drop_wbuf(ref);
        for (auto &e : heap) {
            drop_extent(e);
            VaBump::give_back(e.va, kChunk);
        }
        heap.clear();
        for (auto &z : zones) {
            for (auto &e : z.extents) drop_extent(e);
            for (auto &rg : z.regions) VaBump::give_back(rg.first, rg.second);
            z = Zone();
        }
        smalls.clear();
        last_zone = -1;
        for (auto &r : ref_va) r = nullptr;
        n_refs = 0;
        mapped_total = used = 0;
        usable = false;
        inited = false;
    }

  private:
    struct Extent {  // a VA-contiguous run of chunks with a free list
        char *va = nullptr;
        size_t len = 0;
        std::vector<hipMemGenericAllocationHandle_t> chunks;
        std::map<size_t, size_t> free;  // offset -> length
    };
    struct Zone {  // chunks of one label, mapped back to back (one extent unless trim() cut the top off)
        char *region = nullptr;
        size_t region_len = 0, region_used = 0, mapped = 0;
        bool sealed = false;
        std::vector<Extent> extents;
        std::vector<std::pair<char *, size_t>> regions;  // every reservation of the zone (one, unless the budget was raised later)
    };
    struct Small { int zone, extent; size_t off, len; };  // zone: -1 = the heap, else zones[zone]
    struct WBuf {  // a write buffer: one physical allocation, one mapping
        hipMemGenericAllocationHandle_t h;
        char *va = nullptr;
        size_t bytes = 0, used = 0;
    };

    int device = 0;
    ArenaProbeFn probe = nullptr;
    ArenaProbeMultiFn probe_multi = nullptr;
    const int *spent_launches = nullptr;   // the context's probe counters (knox_arena_info): what the budget is held against
    const double *spent_ms = nullptr;
    int last_label = 1;                    // label of the chunk labelled last: inherited by chunks taken after the budget is spent
    void *user = nullptr;
    bool inited = false;
    size_t used = 0;
    WBuf ref;  // class 0's reference window (32 MiB of its own, made with the first column): "class 0" is whatever it lies in
    char *ref_va[kMaxClasses] = {nullptr, nullptr, nullptr, nullptr};  // class c's reference window; c >= 1: the first 32 MiB of zone c
    int n_refs = 0;
    int grown_for_write = 0;
    int last_zone = -1;  // the class zone of the latest column (or reserve()): the next column tries it first
    Zone zones[kMaxClasses + 1];  // [kMaxClasses]: the mixed region (chunks of several classes, see mixed_alloc)
    bool fake_labels = false;     // test hook (KNOX_ARENA_FAKE_LABELS=1): chunks are labelled class 0, 1, 0, 1 ... without probing
    std::vector<Extent> heap;
    std::vector<WBuf> spares;
    std::map<char *, Small> smalls;
    std::map<char *, WBuf> wbufs;

    static size_t round_up(size_t bytes) {
        if (bytes == 0) bytes = 256;
        const size_t align = bytes >= ((size_t)1 << 20) ? 4096 : 256;
        return (bytes + align - 1) & ~(align - 1);
    }
    static void dbg(const char *fmt, ...) __attribute__((format(printf, 1, 2))) {
        // KNOX_ARENA_LOG=<path>: every decision appended to that file (bench.py keeps one per run); KNOX_ARENA_DEBUG: to stderr
        static FILE *const sink = [] {
            const char *path = getenv("KNOX_ARENA_LOG");
            FILE *f = path && *path ? fopen(path, "a") : nullptr;
            return f ? f : (getenv("KNOX_ARENA_DEBUG") != nullptr ? stderr : nullptr);
        }();
        if (!sink) return;
        va_list ap;
        va_start(ap, fmt);
        fprintf(sink, "[arena] ");
        vfprintf(sink, fmt, ap);
        fprintf(sink, "\n");
        if (sink != stderr) fflush(sink);
        va_end(ap);
    }
    static void drop_extent(Extent &e) {  // mapping by mapping, as they were made
        for (size_t i = 0; i < e.chunks.size(); i++) {
            if (!e.chunks[i]) continue;  // moved into the mixed region (mixed_alloc): unmapped there, released there
            (void)hipMemUnmap(e.va + i * kChunk, kChunk);
            (void)hipMemRelease(e.chunks[i]);
        }
        e.chunks.clear();
    }
    void account(size_t bytes) {
        used += bytes;
        peak_used = std::max(peak_used, used);
    }
    static bool take(std::map<size_t, size_t> &fl, size_t bytes, size_t *off) {
        for (auto it = fl.begin(); it != fl.end(); ++it) {
            if (it->second < bytes) continue;
            const size_t o = it->first, l = it->second;
            fl.erase(it);
            if (l > bytes) fl[o + bytes] = l - bytes;
            *off = o;
            return true;
        }
        return false;
    }
    // the highest fitting range (write buffers inside column zones: columns fill a zone from the bottom)
    static bool take_top(std::map<size_t, size_t> &fl, size_t bytes, size_t *off) {
        for (auto it = fl.rbegin(); it != fl.rend(); ++it) {
            if (it->second < bytes) continue;
            const size_t o = it->first, l = it->second;
            fl.erase(std::next(it).base());
            if (l > bytes) fl[o] = l - bytes;
            *off = o + l - bytes;
            return true;
        }
        return false;
    }
    static bool take_at(std::map<size_t, size_t> &fl, size_t at, size_t bytes) {  // exactly [at, at + bytes)
        auto it = fl.upper_bound(at);
        if (it == fl.begin()) return false;
        --it;
        const size_t o = it->first, l = it->second;
        if (o > at || o + l < at + bytes) return false;
        fl.erase(it);
        if (at > o) fl[o] = at - o;
        if (o + l > at + bytes) fl[at + bytes] = o + l - (at + bytes);
        return true;
    }
    int zone_of(const void *p) const {
        for (int zi = 0; zi < kMaxClasses; zi++)
            for (auto &rg : zones[zi].regions)
                if ((const char *)p >= rg.first && (const char *)p < rg.first + rg.second) return zi;
        return -1;
    }
    bool zone_take_top(int zi, size_t bytes, char **va) {
        Zone &z = zones[zi];
        for (size_t ei = z.extents.size(); ei-- > 0;) {
            size_t off = 0;
            if (!take_top(z.extents[ei].free, bytes, &off)) continue;
            *va = z.extents[ei].va + off;
            smalls[*va] = Small{zi, (int)ei, off, bytes};
            account(bytes);
            return true;
        }
        return false;
    }
    static void give(std::map<size_t, size_t> &fl, size_t off, size_t len) {
        auto nx = fl.lower_bound(off);
        if (nx != fl.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) {
                off = pv->first;
                len += pv->second;
                fl.erase(pv);
            }
        }
        if (nx != fl.end() && off + len == nx->first) {
            len += nx->second;
            fl.erase(nx);
        }
        fl[off] = len;
    }
    bool map_at(char *va, hipMemGenericAllocationHandle_t h, size_t bytes) {
        hipError_t e = hipMemMap(va, bytes, 0, h, 0);
        if (e != hipSuccess) {
            dbg("hipMemMap(%p, %zu) failed: %d", (void *)va, bytes, (int)e);
            (void)hipGetLastError();
            return false;
        }
        hipMemAccessDesc d;
        d.location.type = hipMemLocationTypeDevice;
        d.location.id = device;
        d.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, bytes, &d, 1);
        if (e != hipSuccess) {
            dbg("hipMemSetAccess(%p, %zu) failed: %d", (void *)va, bytes, (int)e);
            (void)hipGetLastError();
            (void)hipMemUnmap(va, bytes);
            return false;
        }
        return true;
    }
    bool create(size_t bytes, hipMemGenericAllocationHandle_t *h) {
        if (mapped_total + bytes > budget) return false;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        if (hipMemCreate(h, bytes, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        mapped_total += bytes;
        return true;
    }

    bool init_once() {
        if (inited) return usable;
        inited = true;
        fake_labels = getenv("KNOX_ARENA_FAKE_LABELS") != nullptr && atoi(getenv("KNOX_ARENA_FAKE_LABELS")) != 0;
        usable = grow_heap();  // no virtual-memory API, or not even one chunk: the caller falls back to plain hipMalloc
        return usable;
    }
    bool grow_heap() {
        Extent e;
        hipMemGenericAllocationHandle_t h;
        if (!create(kChunk, &h)) return false;
        e.va = VaBump::take(kChunk);
        if (!e.va || !map_at(e.va, h, kChunk)) {
            VaBump::give_back(e.va, kChunk);
            (void)hipMemRelease(h);
            mapped_total -= kChunk;
            return false;
        }
        chunks_created++;
        e.len = kChunk;
        e.chunks.push_back(h);
        e.free[0] = kChunk;
        heap.push_back(e);
        return true;
    }
    hipError_t heap_alloc(void **out, size_t bytes) {
        if (bytes > kChunk) return hipErrorOutOfMemory;
        for (int pass = 0; pass < 2; pass++) {
            for (size_t i = 0; i < heap.size(); i++) {
                size_t off = 0;
                if (!take(heap[i].free, bytes, &off)) continue;
                smalls[heap[i].va + off] = Small{-1, (int)i, off, bytes};
                account(bytes);
                *out = heap[i].va + off;
                return hipSuccess;
            }
            if (pass == 1 || !grow_heap()) break;
        }
        return hipErrorOutOfMemory;
    }

    // One more chunk: mapped once at a fresh address to be labelled against the reference window, then (re)mapped -- at
    // another fresh address -- at the end of its label's zone.  Returns the zone index, -1 = no chunk.
    int grow_zone() {
        hipMemGenericAllocationHandle_t h;
        if (!create(kChunk, &h)) return -1;
        chunks_created++;
        int label = 1;
        bool founder = false;  // the first chunk of a class not met before: its first 32 MiB become that class's reference window
        if (fake_labels) {
            label = 1 + (chunks_created & 1);  // tests: classes that interleave chunk by chunk, the worst case for class-pure zones
        } else if (probe && !may_probe()) {
            label = last_label;  // budget spent: the class of the chunk before (memory comes in runs of one class)
            unprobed_chunks++;
            dbg("chunk %d takes the label of the chunk before it (class %d): probe budget spent", chunks_created, label - 1);
        } else if (probe) {
            if (!ref.va) {
                if (!new_wbuf(kRefBytes, &ref)) {
                    (void)hipMemRelease(h);
                    mapped_total -= kChunk;
                    return -1;
                }
                ref_va[0] = ref.va;
                n_refs = 1;
            }
            char *pva = VaBump::take(kChunk);
            if (!pva || !map_at(pva, h, kChunk)) {
                VaBump::give_back(pva, kChunk);
                (void)hipMemRelease(h);
                mapped_total -= kChunk;
                return -1;
            }
            // fresh physical memory may still be under the driver's own clearing pass: touch all of it once and wait
            (void)hipMemsetAsync(pva, 0, kChunk, nullptr);
            (void)hipDeviceSynchronize();
            // a chunk belongs to the class whose reference window it is SLOW with (same class = slow); fast with all of
            // them = a class of its own
            int cls = -1, closest = 0;
            double lowest = 2.0;
            // two or more classes known: all of them in ONE short launch sequence (8 launches for two classes); only a
            // reading near the threshold goes through the long form below
            double quick[kMaxClasses] = {0, 0, 0, 0};
            bool have_quick = false;
            static const bool quick_on = !(getenv("KNOX_ARENA_QUICK") && atoi(getenv("KNOX_ARENA_QUICK")) == 0);  // 0: long form only (A/B)
            if (probe_multi && n_refs >= 1 && quick_on) {  // round 4: also while ONE class is known (6 launches instead of 20)
                void *refs[kMaxClasses];
                for (int c = 0; c < n_refs; c++) refs[c] = ref_va[c];
                have_quick = probe_multi(user, pva, kProbeRows, refs, n_refs, quick) == 0;
                if (have_quick) probes_run++;
            }
            for (int c = 0; c < n_refs && cls < 0; c++) {
                double r = 0.0;
                for (int rep = 0; rep < 2; rep++) {  // a reading near the threshold is taken again and averaged
                    if (rep == 0 && have_quick && (quick[c] < 0.91 || quick[c] > 0.96)) {  // same class reads 0.86-0.90, another one 0.97-1.02
                        r = quick[c];
                        break;
                    }
                    const double x = probe(user, pva, kProbeRows, ref_va[c]);
                    probes_run++;
                    if (x <= 0.0) { r = 0.0; break; }
                    r = rep == 0 ? x : 0.5 * (r + x);
                    if (r < 0.915 || r > 0.965) break;
                }
                dbg("chunk %d vs the reference window of class %d: %.3f%s", chunks_created, c, r, have_quick ? (r == quick[c] ? " (quick)" : " (long form after a quick reading near the threshold)") : "");
                if (r > 0.0 && r < lowest) { lowest = r; closest = c; }
                if (r > 0.0 && r < 0.94) cls = c;
            }
            if (cls < 0) {
                if (n_refs < kMaxClasses && lowest <= 1.5) { cls = n_refs; founder = true; }
                else cls = closest;
            }
            label = cls + 1;
            last_label = label;
            (void)hipDeviceSynchronize();
            (void)hipMemUnmap(pva, kChunk);
            VaBump::give_back(pva, kChunk);  // the probe mapping had a reservation of its own
        }
        Zone &z = zones[label - 1];
        if (!z.region || z.region_used + kChunk > z.region_len) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)288 << 30;
            z.region_len = std::min(budget, total_b) + kChunk;
            z.region = VaBump::take(z.region_len);
            if (z.region) z.regions.push_back({z.region, z.region_len});
            z.region_used = 0;
            z.sealed = true;
            if (!z.region) {
                (void)hipMemRelease(h);
                mapped_total -= kChunk;
                return -1;
            }
        }
        char *va = z.region + z.region_used;
        if (!map_at(va, h, kChunk)) {
            (void)hipMemRelease(h);
            mapped_total -= kChunk;
            return -1;
        }
        z.region_used += kChunk;
        if (z.sealed || z.extents.empty()) {
            Extent e;
            e.va = va;
            z.extents.push_back(e);
            z.sealed = false;
        }
        Extent &e = z.extents.back();
        give(e.free, e.len, kChunk);
        e.len += kChunk;
        e.chunks.push_back(h);
        z.mapped += kChunk;
        if (founder && take_at(e.free, e.len - kChunk, kRefBytes)) {
            ref_va[label - 1] = va;
            n_refs = label;
            account(kRefBytes);
            (void)hipMemsetAsync(va, 0, kRefBytes, nullptr);
            (void)hipDeviceSynchronize();
            dbg("class %d met: its reference window is the start of chunk %d", label - 1, chunks_created);
        }
        return label - 1;
    }
    // A column that fits no single class zone although the arena has the room (classes that interleave every few GiB
    // strand free chunks in every zone): wholly free chunks are taken OUT of their zones -- unmapped there, the hole in the
    // zone's address range is never used again (VMM fact 1) -- and mapped back to back into a fresh reservation of the mixed
    // region.  Such a column is not class pure; the write buffers of its scans are verified per (column, buffer) pair like
    // all others, they just cannot rely on "another class's zone" being fast by construction.
    hipError_t mixed_alloc(void **out, size_t bytes) {
        const size_t need = (bytes + kChunk - 1) / kChunk;
        struct Loc { int zi; size_t ei, ci; };
        std::vector<Loc> found;
        for (int zi = 0; zi < kMaxClasses && found.size() < need; zi++)
            for (size_t ei = 0; ei < zones[zi].extents.size() && found.size() < need; ei++) {
                Extent &e = zones[zi].extents[ei];
                for (auto &fr : e.free) {
                    size_t c0 = (fr.first + kChunk - 1) / kChunk, c1 = (fr.first + fr.second) / kChunk;
                    for (size_t ci = c0; ci < c1 && found.size() < need; ci++)
                        if (e.chunks[ci]) found.push_back(Loc{zi, ei, ci});
                }
            }
        // what the free chunks do not cover comes fresh from the driver (labelled or not, it ends up in the mixed region)
        std::vector<hipMemGenericAllocationHandle_t> fresh;
        while (found.size() + fresh.size() < need) {
            hipMemGenericAllocationHandle_t h;
            if (!create(kChunk, &h)) break;
            chunks_created++;
            fresh.push_back(h);
        }
        if (found.size() + fresh.size() < need) {
            for (auto h : fresh) {
                (void)hipMemRelease(h);
                mapped_total -= kChunk;
            }
            return hipErrorOutOfMemory;
        }
        char *va = VaBump::take(need * kChunk);
        if (!va) {
            for (auto h : fresh) {
                (void)hipMemRelease(h);
                mapped_total -= kChunk;
            }
            return hipErrorOutOfMemory;
        }
        Zone &mz = zones[kMaxClasses];
        mz.regions.push_back({va, need * kChunk});
        Extent me;
        me.va = va;
        size_t k = 0;
        for (const Loc &l : found) {
            Extent &e = zones[l.zi].extents[l.ei];
            if (!take_at(e.free, l.ci * kChunk, kChunk)) continue;  // cannot happen: the range was free a moment ago
            (void)hipMemUnmap(e.va + l.ci * kChunk, kChunk);
            const hipMemGenericAllocationHandle_t h = e.chunks[l.ci];
            e.chunks[l.ci] = nullptr;
            zones[l.zi].mapped -= kChunk;
            if (!map_at(va + k * kChunk, h, kChunk)) return hipErrorOutOfMemory;
            me.chunks.push_back(h);
            k++;
        }
        for (auto h : fresh) {
            if (!map_at(va + k * kChunk, h, kChunk)) return hipErrorOutOfMemory;
            me.chunks.push_back(h);
            k++;
        }
        me.len = k * kChunk;
        me.free[0] = me.len;
        mz.mapped += me.len;
        mixed_bytes += me.len;
        mz.extents.push_back(me);
        dbg("mixed region: %zu chunks (%zu out of class zones, %zu fresh) for a request of %zu bytes", k, found.size(), fresh.size(), bytes);
        size_t off = 0;
        Extent &ne = mz.extents.back();
        if (!take(ne.free, bytes, &off)) return hipErrorOutOfMemory;
        smalls[ne.va + off] = Small{kMaxClasses, (int)(mz.extents.size() - 1), off, bytes};
        account(bytes);
        *out = ne.va + off;
        return hipSuccess;
    }
    hipError_t zone_alloc(void **out, size_t bytes) {
        for (;;) {
            // the zone of the previous column first: columns that are scanned TOGETHER (the leaves of one tree, the pointer
            // array and the pool of a bytes column) read 4-5 % faster when they lie in ONE placement class (two 8 GiB float64
            // columns, read-only two-leaf scan: 6.88-6.93 TB/s in one zone, 6.60-6.64 TB/s in two; tools/two_stream_skew.py)
            for (int k = 0; k <= kMaxClasses + 1; k++) {
                const int zi = k == 0 ? last_zone : k - 1;
                if (zi < 0 || (k > 0 && zi == last_zone)) continue;
                for (size_t ei = 0; ei < zones[zi].extents.size(); ei++) {
                    size_t off = 0;
                    if (!take(zones[zi].extents[ei].free, bytes, &off)) continue;
                    if (zi < kMaxClasses) last_zone = zi;
                    smalls[zones[zi].extents[ei].va + off] = Small{zi, (int)ei, off, bytes};
                    account(bytes);
                    *out = zones[zi].extents[ei].va + off;
                    dbg("%zu MiB in zone %d at offset %zu MiB (zone: %zu MiB mapped)", bytes >> 20, zi, off >> 20, zones[zi].mapped >> 20);
                    return hipSuccess;
                }
            }
            // one more chunk while the budget allows it AND a class zone can still come to hold the request; otherwise the
            // free chunks of all zones together (mixed region)
            if (mapped_total + kChunk > budget || grow_zone() < 0) {
                const hipError_t e = mixed_alloc(out, bytes);
                if (e != hipSuccess) dbg("zone_alloc: %zu bytes do not fit (mapped %zu of %zu)", bytes, mapped_total, budget);
                return e;
            }
        }
    }

    bool new_wbuf(size_t bytes, WBuf *w) {
        if (!create(bytes, &w->h)) return false;
        w->va = VaBump::take_own(bytes);
        if (!w->va || !map_at(w->va, w->h, bytes)) {
            VaBump::give_back(w->va, bytes);
            (void)hipMemRelease(w->h);
            mapped_total -= bytes;
            w->va = nullptr;
            return false;
        }
        w->bytes = bytes;
        write_buffers++;
        // fresh physical memory may still be under the driver's own clearing pass (it runs on the DMA engines,
        // unsynchronised with user queues): touch all of it once and wait, then measure
        (void)hipMemsetAsync(w->va, 0, bytes, nullptr);
        (void)hipDeviceSynchronize();
        return true;
    }
    void drop_wbuf(WBuf &w) {
        if (!w.va) return;
        (void)hipMemUnmap(w.va, w.bytes);
        (void)hipMemRelease(w.h);
        VaBump::give_back(w.va, w.bytes);
        mapped_total -= w.bytes;
        w.va = nullptr;
    }
    void add_spare(WBuf &w) {
        if (!w.va) return;
        if (spare_bytes() + w.bytes > kMaxSpareBytes) {
            drop_wbuf(w);
            return;
        }
        w.used = 0;
        spares.push_back(w);
    }
    hipError_t take_wbuf(void **out, WBuf &w, size_t bytes, double r, double *ratio_out, WBuf &other_best) {
        if (other_best.va && other_best.va != w.va) add_spare(other_best);
        other_best.va = nullptr;
        w.used = bytes;
        wbufs[w.va] = w;
        account(bytes);
        *out = w.va;
        if (ratio_out) *ratio_out = r;
        return hipSuccess;
    }
};

}  // namespace knox
