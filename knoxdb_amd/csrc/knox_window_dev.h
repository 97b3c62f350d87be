// knox_window_dev.h -- device helpers shared by the out-of-order window kernels: the stable sort and its compaction pass
// (knox_sort.hip) and the sort-free folds (knox_fold.hip).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace knox {
namespace {
constexpr uint32_t kNoKey = 0xffffffffu;

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// window index of a timestamp: (t - t_from) / step when that is below n_buckets, else kNoKey.  The 64-bit division the compiler
// emits is ~150 instructions per row (it was most of the key pass: 0.68 ms per 2^28 rows at 6 % VALU "busy", and 1.0 of the
// compaction pass's first 1.3 ms); the quotient of interest is below 2^26, so a double-precision estimate is off by at most one and a
// multiply-back settles it.  span = n_buckets * step (the host checked that every window edge fits int64), inv = 1.0 / step.
__device__ __forceinline__ uint32_t window_index(int64_t t, int64_t t_from, uint64_t step, uint64_t span, double inv) {
    const uint64_t d = (uint64_t)t - (uint64_t)t_from;
    // the quotient is below 2^26 (n_buckets <= 2^26, checked by the host): the estimate is within one of it and is CONVERTED AS 32 BITS --
    // one instruction where the 64-bit conversion is a dozen (this function was most of the 480 vector instructions the compaction
    // spent per 512 rows)
    uint32_t q = (uint32_t)((double)d * inv);
    const uint64_t back = (uint64_t)q * step;
    if (back > d) q--;                              // estimate one too high
    else if (d - back >= step) q++;                 // one too low
    return (t < t_from || d >= span) ? kNoKey : q;
}

template <typename U>
struct CompactPairOf {   // two neighbouring values of the column in one load
    typedef U type __attribute__((ext_vector_type(2)));
};
template <typename U>
using CompactPair = typename CompactPairOf<U>::type;

}  // namespace
}  // namespace knox
