// Out-of-order time windows (internal/reducer/bucket_native.go:123-141: Push finds or inserts the window's
// reducer by time, so rows may arrive in any time order).  The device brings the SELECTED rows into window
// order with a stable sort of (window index, row id) pairs; within a window the rows keep their stream order,
// which is the only order a reducer can observe.  The sort itself is rocPRIM's device radix sort (a plain
// library primitive, like a BLAS call -- the knox-specific kernels around it are hand-written): it is not on
// the headline path, ordered time columns never come here.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "knox_kernels.h"

namespace knox {

namespace {
constexpr int kSortBlock = 256;
}

// key = window index of the row's timestamp; rows that are not selected or fall outside every window get
// the sentinel n_buckets and sort to the end
__global__ void __launch_bounds__(kSortBlock) window_keys_kernel(const int64_t *__restrict__ time, uint64_t n_rows, const uint8_t *__restrict__ mask,
                                                                 int64_t t_from, int64_t step, uint32_t n_buckets, uint32_t *__restrict__ keys,
                                                                 uint32_t *__restrict__ idx) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride) {
        uint32_t k = n_buckets;
        if (mask == nullptr || ((mask[i >> 3] >> (i & 7)) & 1)) {
            const int64_t t = time[i];
            if (t >= t_from) {
                const uint64_t q = (uint64_t)(t - t_from) / (uint64_t)step;  // host checked: every window edge fits int64
                if (q < n_buckets) k = (uint32_t)q;
            }
        }
        keys[i] = k;
        idx[i] = (uint32_t)i;
    }
}

hipError_t launch_window_keys(hipStream_t s, const int64_t *time, uint64_t n_rows, const uint8_t *mask, int64_t t_from, int64_t step,
                              uint32_t n_buckets, uint32_t *keys, uint32_t *idx) {
    if (n_rows == 0) return hipSuccess;
    uint64_t g = (n_rows + kSortBlock - 1) / kSortBlock;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(window_keys_kernel, dim3((unsigned)g), dim3(kSortBlock), 0, s, time, n_rows, mask, t_from, step, n_buckets, keys, idx);
    return hipGetLastError();
}

// stable LSD radix sort over the low key_bits of the keys; tmp == nullptr: size query
hipError_t launch_sort_pairs_u32(void *tmp, size_t &tmp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                                 uint32_t *vals_out, size_t n, unsigned key_bits, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, key_bits, s);
}

template <typename U>
__global__ void __launch_bounds__(kSortBlock) gather_rows_kernel(const U *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t n, U *__restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

hipError_t launch_gather_rows(hipStream_t s, int width, const void *src, const uint32_t *idx, uint64_t n, void *dst) {
    if (n == 0) return hipSuccess;
    uint64_t g = (n + kSortBlock - 1) / kSortBlock;
    if (g > 8192) g = 8192;
    switch (width) {
    case 8: hipLaunchKernelGGL((gather_rows_kernel<uint64_t>), dim3((unsigned)g), dim3(kSortBlock), 0, s, (const uint64_t *)src, idx, n, (uint64_t *)dst); break;
    case 4: hipLaunchKernelGGL((gather_rows_kernel<uint32_t>), dim3((unsigned)g), dim3(kSortBlock), 0, s, (const uint32_t *)src, idx, n, (uint32_t *)dst); break;
    case 2: hipLaunchKernelGGL((gather_rows_kernel<uint16_t>), dim3((unsigned)g), dim3(kSortBlock), 0, s, (const uint16_t *)src, idx, n, (uint16_t *)dst); break;
    case 1: hipLaunchKernelGGL((gather_rows_kernel<uint8_t>), dim3((unsigned)g), dim3(kSortBlock), 0, s, (const uint8_t *)src, idx, n, (uint8_t *)dst); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// start[k] = first sorted position whose key is >= k, k = 0..n_buckets (start[n_buckets] = rows inside windows)
__global__ void __launch_bounds__(kSortBlock) key_bounds_kernel(const uint32_t *__restrict__ keys, uint64_t n, uint32_t n_buckets, uint64_t *__restrict__ start) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_buckets) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    start[k] = lo;
}

hipError_t launch_key_bounds(hipStream_t s, const uint32_t *keys, uint64_t n, uint32_t n_buckets, uint64_t *start) {
    hipLaunchKernelGGL(key_bounds_kernel, dim3((n_buckets + 1 + kSortBlock - 1) / kSortBlock), dim3(kSortBlock), 0, s, keys, n, n_buckets, start);
    return hipGetLastError();
}

// the states were reduced over sorted positions: map their row ids back to the caller's rows
__global__ void __launch_bounds__(kSortBlock) remap_rows_kernel(AggStateDev *states, uint32_t n_buckets, const uint32_t *__restrict__ idx, uint64_t row_base) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_buckets) return;
    AggStateDev &s = states[k];
    if (s.min_row >= 0) s.min_row = (int64_t)(idx[s.min_row] + row_base);
    if (s.max_row >= 0) s.max_row = (int64_t)(idx[s.max_row] + row_base);
    if (s.first_row >= 0) s.first_row = (int64_t)(idx[s.first_row] + row_base);
    if (s.last_row >= 0) s.last_row = (int64_t)(idx[s.last_row] + row_base);
}

hipError_t launch_remap_rows(hipStream_t s, AggStateDev *states, uint32_t n_buckets, const uint32_t *idx, uint64_t row_base) {
    hipLaunchKernelGGL(remap_rows_kernel, dim3((n_buckets + kSortBlock - 1) / kSortBlock), dim3(kSortBlock), 0, s, states, n_buckets, idx, row_base);
    return hipGetLastError();
}

}  // namespace knox
