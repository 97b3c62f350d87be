// knox_arena.h -- one bounded device-memory arena per context (host code, included by knox_abi.cpp only).
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
// returned before the context closes (or an explicit trim).  Columns (>= 1 GiB) are assembled from 1 GiB physical
// chunks mapped back to back (hipMemCreate + the virtual-memory API; any chunks, in any order) -- all of ONE kind:
// every chunk is probed once against one fixed 16 MiB reference window ("is a 1 GiB column here slow with it?": slow =
// the window's class, fast = some other class), so that a write buffer that is fast with one part of a column is fast
// with all of it.  Smaller things share an ordinary heap.  A large
// write buffer that is allocated FOR a column (knox_bits_alloc_for, the scan's scratch bitsets) is a physical
// allocation of its own, chosen among candidates: spare write buffers first, then a few new ones (small allocations
// are served from the driver's leftover fragments, which lie all over the device: usually one of a handful is in
// another class), then candidates from FURTHER AWAY in the driver's allocation order, reached with transient spacers
// (physical memory that is never mapped and is released again before the call returns).  Every candidate is VERIFIED:
// up to four 1 GiB samples of the column are scanned into the part of the candidate their bits really go to; the first
// candidate whose worst sample runs at >= 0.95 of its read-only form is taken.  Rejected candidates stay in the arena
// as spare write buffers (the first candidates of the next request: what is slow with one column is fast with a
// column of another class); nothing is held outside the accounting.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <vector>

namespace knox {

enum ArenaKind { ARENA_READ = 0, ARENA_WRITE = 1 };

// scan / read-only: the rate of the scan over `rows` int64 rows at `col` into the bitset at `bits`, divided by the rate of the
// read-only form of the same scan, measured with ALTERNATING launches (whatever else disturbs the device -- the driver
// clearing freshly taken or released memory costs bandwidth for seconds -- hits both alike); <= 0 = failed
typedef double (*ArenaProbeFn)(void *user, const void *col, uint64_t rows, void *bits);

class Arena {
  public:
    static constexpr size_t kChunk = (size_t)1 << 30;          // physical unit of columns
    static constexpr size_t kSmallWrite = (size_t)16 << 20;    // write buffers below this share the heap (streams that short do not show the effect)
    static constexpr uint64_t kProbeRows = (uint64_t)1 << 27;  // one sample: 1 GiB of int64 rows, 16 MiB of bitset
    static constexpr double kFastAbove = 0.95;                 // scan / read-only: ~0.88 same class, ~0.98 different classes
    static constexpr int kNearCandidates = 6;                  // new write buffers looked at before reaching further
    static constexpr size_t kSpacer = (size_t)24 << 30;        // transient, never mapped: skips ahead in the driver's allocation order
    static constexpr int kMaxSpacers = 6;
    static constexpr size_t kMaxSpareBytes = (size_t)6 << 30;  // spare write buffers kept (beyond: returned to the driver; they are small)

    bool usable = false;
    size_t budget = 0, mapped_total = 0, peak_used = 0;
    int chunks_created = 0, probes_run = 0, write_buffers = 0;
    size_t spacer_bytes_used = 0;  // transient spacers taken and released again so far (reporting)

    void configure(int device_, size_t budget_, ArenaProbeFn fn, void *user_) {
        device = device_;
        budget = budget_;
        probe = fn;
        user = user_;
    }
    size_t used_bytes() const { return used; }
    size_t spare_bytes() const {
        size_t n = 0;
        for (auto &w : spares) n += w.bytes;
        return n;
    }

    // Columns / payloads of >= 1 GiB: whole 1 GiB chunks + a small tail, mapped back to back.  Everything else: the heap.
    hipError_t alloc(void **out, size_t bytes, int kind) {
        *out = nullptr;
        bytes = round_up(bytes);
        if (!init_once()) return hipErrorNotSupported;
        if (kind == ARENA_READ && bytes >= kChunk) return big_alloc(out, bytes);
        return heap_alloc(out, bytes);
    }

    // A write buffer FOR a column: a physical allocation of its own, verified against the column (see the header comment).
    // *ratio_out: scan / read-only of the worst sample with the buffer that was taken (0 = not measured); *tried_out:
    // candidates looked at.  bytes_per_row: column bytes per bit of the buffer.
    hipError_t alloc_write_for(void **out, size_t bytes, const void *col, size_t col_bytes, double bytes_per_row, double *ratio_out,
                               int *tried_out) {
        *out = nullptr;
        if (ratio_out) *ratio_out = 0.0;
        if (tried_out) *tried_out = 0;
        bytes = round_up(bytes);
        if (!init_once()) return hipErrorNotSupported;
        const size_t sample = kProbeRows * 8, wrote = kProbeRows / 8;
        if (!probe || !col || col_bytes < sample || bytes < std::max(kSmallWrite, wrote + 64) || bytes_per_row <= 0.0) return heap_alloc(out, bytes);
        const size_t wb_bytes = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        // the column's samples and their read-only rates, once
        constexpr int kMaxSamples = 4;
        const int n_s = (int)std::min<size_t>(kMaxSamples, col_bytes / sample);
        size_t offs[kMaxSamples] = {0}, boffs[kMaxSamples] = {0};
        for (int i = 0; i < n_s; i++) {
            offs[i] = n_s > 1 ? (((col_bytes - sample) / (size_t)(n_s - 1) * (size_t)i) & ~(size_t)0xffff) : 0;
            boffs[i] = ((size_t)((double)offs[i] / bytes_per_row) >> 3) & ~(size_t)255;
            if (boffs[i] + wrote + 64 > bytes) boffs[i] = (bytes - wrote - 64) & ~(size_t)255;
        }
        auto measure = [&](char *va) {
            double worst = 2.0;
            for (int i = 0; i < n_s; i++) {
                const double r = probe(user, (const char *)col + offs[i], kProbeRows, va + boffs[i]);
                probes_run++;
                if (r <= 0.0) return -1.0;
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
        // 1. spare write buffers of a fitting size
        for (size_t i = 0; i < spares.size();) {
            if (spares[i].bytes < wb_bytes || spares[i].bytes > 2 * wb_bytes) { i++; continue; }
            WBuf w = spares[i];
            spares.erase(spares.begin() + (long)i);
            const double r = measure(w.va);
            dbg("spare write buffer (%zu MiB) vs the column: %.3f", w.bytes >> 20, r);
            if (r >= kFastAbove) return take_wbuf(out, w, bytes, r, ratio_out, best);
            consider(w, r);
            spares_rotated++;
            if (spares_rotated > 64) break;
        }
        spares_rotated = 0;
        // 2. a few new ones; 3. further away, behind transient spacers
        auto try_new = [&](int n_tries, bool *hit) {
            for (int k = 0; k < n_tries; k++) {
                WBuf w;
                if (!new_wbuf(wb_bytes, &w)) return false;
                const double r = measure(w.va);
                dbg("new write buffer (%zu MiB) vs the column: %.3f", w.bytes >> 20, r);
                if (r >= kFastAbove) {
                    *hit = true;
                    return take_wbuf(out, w, bytes, r, ratio_out, best) == hipSuccess;
                }
                consider(w, r);
            }
            return true;
        };
        bool hit = false;
        bool can_go_on = try_new(kNearCandidates, &hit);
        if (hit) return hipSuccess;
        std::vector<hipMemGenericAllocationHandle_t> spacers;
        for (int step = 0; step < kMaxSpacers && can_go_on && !hit; step++) {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = device;
            hipMemGenericAllocationHandle_t sp;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < kSpacer + ((size_t)8 << 30)) break;
            if (hipMemCreate(&sp, kSpacer, &prop, 0) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            spacers.push_back(sp);
            dbg("spacer %d taken (%zu GiB ahead in the allocation order)", step, (spacers.size() * kSpacer) >> 30);
            can_go_on = try_new(3, &hit);
        }
        if (!spacers.empty()) {
            (void)hipDeviceSynchronize();
            for (auto &sp : spacers) (void)hipMemRelease(sp);
            spacer_bytes_used += spacers.size() * kSpacer;
            usleep(1500000);  // the driver clears what was released; it costs the whole device bandwidth while it runs
        }
        if (hit) return hipSuccess;
        if (!best.va) return heap_alloc(out, bytes);
        WBuf none;
        return take_wbuf(out, best, bytes, best_r, ratio_out, none);  // nothing fast within reach: the best there is
    }

    // true when p was an arena allocation (and is now free)
    bool release(void *p) {
        auto s = smalls.find((char *)p);
        if (s != smalls.end()) {
            give(small.free, s->second.off, s->second.len);
            used -= s->second.len;
            smalls.erase(s);
            return true;
        }
        auto w = wbufs.find((char *)p);
        if (w != wbufs.end()) {
            used -= w->second.used;
            WBuf wb = w->second;
            wbufs.erase(w);
            add_spare(wb);  // kept: the next request of that size tries it first
            return true;
        }
        auto b = bigs.find((char *)p);
        if (b == bigs.end()) return false;
        Big big = b->second;
        bigs.erase(b);
        unmap_big(big);
        (void)hipMemAddressFree(big.va, big.va_bytes);
        for (int id : big.ids) free_chunks.push_back(id);
        if (big.tail_bytes) {
            (void)hipMemRelease(big.tail);  // a few MiB
            mapped_total -= big.tail_bytes;
        }
        used -= big.bytes;
        return true;
    }

    // returns every free column chunk and every spare write buffer to the driver (explicit, never automatic)
    size_t trim() {
        size_t freed = 0;
        for (int id : free_chunks) {
            (void)hipMemRelease(chunks[(size_t)id].h);
            chunks[(size_t)id].gone = true;
            mapped_total -= kChunk;
            freed += kChunk;
        }
        free_chunks.clear();
        for (auto &w : spares) {
            freed += w.bytes;
            drop_wbuf(w);
        }
        spares.clear();
        return freed;
    }

    void destroy() {
        for (auto &kv : bigs) {
            unmap_big(kv.second);
            (void)hipMemAddressFree(kv.second.va, kv.second.va_bytes);
            if (kv.second.tail_bytes) (void)hipMemRelease(kv.second.tail);
        }
        bigs.clear();
        for (auto &kv : wbufs) drop_wbuf(kv.second);
        wbufs.clear();
        for (auto &w : spares) drop_wbuf(w);
        spares.clear();
        drop_wbuf(ref);
        if (scratch_va) (void)hipMemAddressFree(scratch_va, kChunk);
        scratch_va = nullptr;
        if (small.va) {
            for (size_t o = 0; o < small.mapped; o += kChunk) (void)hipMemUnmap(small.va + o, kChunk);  // mapping by mapping
            (void)hipMemAddressFree(small.va, kSmallVa);
        }
        small = Heap();
        for (auto &c : chunks)
            if (!c.gone) (void)hipMemRelease(c.h);
        chunks.clear();
        free_chunks.clear();
        smalls.clear();
        mapped_total = used = 0;
        usable = false;
        inited = false;
    }

  private:
    struct Chunk {
        hipMemGenericAllocationHandle_t h;
        bool gone = false;
        int label = 0;  // vs the reference window: 0 not probed, 1 slow (the window's class), 2 fast (another class)
    };
    struct Heap {  // VA-contiguous heap over whole chunks
        char *va = nullptr;
        size_t mapped = 0;
        std::map<size_t, size_t> free;  // offset -> length
    };
    struct Small { size_t off, len; };
    struct WBuf {  // a write buffer: one physical allocation, one mapping
        hipMemGenericAllocationHandle_t h;
        char *va = nullptr;
        size_t bytes = 0, used = 0;
    };
    struct Big {
        std::vector<int> ids;
        hipMemGenericAllocationHandle_t tail;
        size_t tail_bytes = 0;
        char *va = nullptr;
        size_t va_bytes = 0, bytes = 0;
    };
    static constexpr size_t kSmallVa = (size_t)64 << 30;

    int device = 0;
    ArenaProbeFn probe = nullptr;
    void *user = nullptr;
    bool inited = false;
    size_t used = 0;
    int spares_rotated = 0;
    char *scratch_va = nullptr;   // where a new chunk is mapped while it is probed
    WBuf ref;                     // the reference window every chunk is labelled against (32 MiB, made with the first column)
    std::vector<Chunk> chunks;
    std::vector<int> free_chunks;
    Heap small;
    std::vector<WBuf> spares;
    std::map<char *, Small> smalls;
    std::map<char *, WBuf> wbufs;
    std::map<char *, Big> bigs;

    static size_t round_up(size_t bytes) {
        if (bytes == 0) bytes = 256;
        const size_t align = bytes >= ((size_t)1 << 20) ? 4096 : 256;
        return (bytes + align - 1) & ~(align - 1);
    }
    static void dbg(const char *fmt, ...) __attribute__((format(printf, 1, 2))) {
        static const bool debug = getenv("KNOX_ARENA_DEBUG") != nullptr;
        if (!debug) return;
        va_list ap;
        va_start(ap, fmt);
        fprintf(stderr, "[arena] ");
        vfprintf(stderr, fmt, ap);
        fprintf(stderr, "\n");
        va_end(ap);
    }
    void account(size_t bytes) {
        used += bytes;
        peak_used = std::max(peak_used, used);
    }
    static void unmap_big(const Big &b) {  // mapping by mapping, as they were made
        for (size_t i = 0; i < b.ids.size(); i++) (void)hipMemUnmap(b.va + i * kChunk, kChunk);
        if (b.tail_bytes) (void)hipMemUnmap(b.va + b.ids.size() * kChunk, b.tail_bytes);
    }
    static void *reserve(size_t bytes) {
        void *p = nullptr;
        if (hipMemAddressReserve(&p, bytes, (size_t)1 << 30, nullptr, 0) == hipSuccess) return p;  // 1 GiB aligned: large page-table fragments
        (void)hipGetLastError();
        if (hipMemAddressReserve(&p, bytes, 0, nullptr, 0) == hipSuccess) return p;
        (void)hipGetLastError();
        return nullptr;
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
    bool set_access(char *va, size_t bytes) {
        hipMemAccessDesc d;
        d.location.type = hipMemLocationTypeDevice;
        d.location.id = device;
        d.flags = hipMemAccessFlagsProtReadWrite;
        const hipError_t e = hipMemSetAccess(va, bytes, &d, 1);
        if (e != hipSuccess) {
            dbg("hipMemSetAccess(%p, %zu) failed: %d", (void *)va, bytes, (int)e);
            (void)hipGetLastError();
            return false;
        }
        return true;
    }
    // with_access = false: the caller sets the access rights of the whole reservation in ONE call after the last mapping
    // (per-mapping calls fail for some sub-GiB pieces that follow GiB pieces in one reservation, ROCm 7.2)
    bool map_at(char *va, hipMemGenericAllocationHandle_t h, size_t bytes, bool with_access = true) {
        const hipError_t e = hipMemMap(va, bytes, 0, h, 0);
        if (e != hipSuccess) {
            dbg("hipMemMap(%p, %zu) failed: %d", (void *)va, bytes, (int)e);
            (void)hipGetLastError();
            return false;
        }
        if (with_access && !set_access(va, bytes)) {
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
        small.va = (char *)reserve(kSmallVa);
        usable = small.va != nullptr && grow_small();
        if (!usable && small.va) {  // no virtual-memory API, or not even one chunk: the caller falls back to plain hipMalloc
            (void)hipMemAddressFree(small.va, kSmallVa);
            small.va = nullptr;
        }
        return usable;
    }
    int new_chunk() {
        Chunk c;
        if (!create(kChunk, &c.h)) return -1;
        chunks_created++;
        chunks.push_back(c);
        return (int)chunks.size() - 1;
    }
    // the chunk's label against the reference window (see the header comment)
    void label_chunk(int id) {
        Chunk &c = chunks[(size_t)id];
        if (c.label || !probe) return;
        if (!scratch_va) scratch_va = (char *)reserve(kChunk);
        if (!ref.va && !new_wbuf((size_t)32 << 20, &ref)) return;
        if (!scratch_va || !map_at(scratch_va, c.h, kChunk)) return;
        // fresh physical memory may still be under the driver's own clearing pass: touch all of it once and wait
        (void)hipMemsetAsync(scratch_va, 0, kChunk, nullptr);
        (void)hipDeviceSynchronize();
        double r = 0.0;
        for (int rep = 0; rep < 2; rep++) {  // a reading near the threshold is taken again and averaged
            const double x = probe(user, scratch_va, kProbeRows, ref.va);
            probes_run++;
            if (x <= 0.0) { r = 0.0; break; }
            r = rep == 0 ? x : 0.5 * (r + x);
            if (r < 0.915 || r > 0.965) break;
        }
        dbg("chunk %d vs the reference window: %.3f", id, r);
        if (r > 0.0) c.label = r < 0.94 ? 1 : 2;
        (void)hipDeviceSynchronize();
        (void)hipMemUnmap(scratch_va, kChunk);
    }
    bool grow_small() {
        if (small.mapped + kChunk > kSmallVa) return false;
        int id = -1;
        if (!free_chunks.empty()) {
            id = free_chunks.back();
            free_chunks.pop_back();
        } else {
            id = new_chunk();
        }
        if (id < 0) return false;
        if (!map_at(small.va + small.mapped, chunks[(size_t)id].h, kChunk)) {
            free_chunks.push_back(id);
            return false;
        }
        give(small.free, small.mapped, kChunk);
        small.mapped += kChunk;
        return true;
    }
    hipError_t heap_alloc(void **out, size_t bytes) {
        size_t off = 0;
        while (!take(small.free, bytes, &off))
            if (bytes > kChunk || !grow_small()) return hipErrorOutOfMemory;
        smalls[small.va + off] = Small{off, bytes};
        account(bytes);
        *out = small.va + off;
        return hipSuccess;
    }

    bool new_wbuf(size_t bytes, WBuf *w) {
        if (!create(bytes, &w->h)) return false;
        w->va = (char *)reserve(bytes);
        if (!w->va || !map_at(w->va, w->h, bytes)) {
            if (w->va) (void)hipMemAddressFree(w->va, bytes);
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
        (void)hipMemAddressFree(w.va, w.bytes);
        (void)hipMemRelease(w.h);
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
        w.used = bytes;
        wbufs[w.va] = w;
        account(bytes);
        *out = w.va;
        if (ratio_out) *ratio_out = r;
        return hipSuccess;
    }

    // A column: n whole chunks + a small tail chunk, mapped back to back
    hipError_t big_alloc(void **out, size_t bytes) {
        const size_t n = bytes / kChunk;
        const size_t tail = (bytes - n * kChunk + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        // n free chunks of ONE label (whichever label gets there first); new chunks as long as the budget allows
        std::vector<int> pick;
        auto gather = [&]() {
            for (int label = 1; label <= 2; label++) {
                pick.clear();
                for (int id : free_chunks)
                    if (pick.size() < n && chunks[(size_t)id].label == label) pick.push_back(id);
                if (pick.size() >= n) return true;
            }
            return false;
        };
        for (int id : free_chunks) label_chunk(id);
        bool one_kind = gather();
        while (!one_kind) {
            if (mapped_total + kChunk + tail + ((size_t)256 << 20) > budget) break;
            const int id = new_chunk();
            if (id < 0) break;
            label_chunk(id);
            free_chunks.push_back(id);
            one_kind = gather();
        }
        if (!one_kind) {  // the budget does not hold n chunks of one kind (or nothing can be probed): whatever is free
            pick.clear();
            for (int id : free_chunks)
                if (pick.size() < n) pick.push_back(id);
            dbg("big_alloc: no %zu chunks of one kind within the budget, %zu free", n, free_chunks.size());
        }
        if (pick.size() < n) {
            dbg("big_alloc: %zu free chunks for %zu wanted (mapped %zu of %zu)", pick.size(), n, mapped_total, budget);
            return hipErrorOutOfMemory;
        }
        Big b;
        b.bytes = bytes;
        b.va_bytes = n * kChunk + tail;
        b.va = (char *)reserve(b.va_bytes);
        if (!b.va) return hipErrorOutOfMemory;
        if (tail && !create(tail, &b.tail)) {
            (void)hipMemAddressFree(b.va, b.va_bytes);
            return hipErrorOutOfMemory;
        }
        b.tail_bytes = tail;
        b.ids = pick;
        bool ok = true, tail_mapped = false;
        size_t done = 0;
        for (; done < n && ok; done++) ok = map_at(b.va + done * kChunk, chunks[(size_t)b.ids[done]].h, kChunk, false);
        if (!ok) done--;  // the last one is not mapped
        if (ok && tail) ok = tail_mapped = map_at(b.va + n * kChunk, b.tail, tail, false);
        if (ok) ok = set_access(b.va, b.va_bytes);
        if (!ok) {
            dbg("big_alloc: mapping failed after %zu of %zu chunks", done, n);
            for (size_t i = 0; i < done; i++) (void)hipMemUnmap(b.va + i * kChunk, kChunk);
            if (tail_mapped) (void)hipMemUnmap(b.va + n * kChunk, tail);
            if (tail) {
                (void)hipMemRelease(b.tail);
                mapped_total -= tail;
            }
            (void)hipMemAddressFree(b.va, b.va_bytes);
            return hipErrorOutOfMemory;
        }
        for (int id : pick) free_chunks.erase(std::find(free_chunks.begin(), free_chunks.end(), id));
        bigs[b.va] = b;
        account(bytes);
        *out = b.va;
        return hipSuccess;
    }
};

}  // namespace knox
