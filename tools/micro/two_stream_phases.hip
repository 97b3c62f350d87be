// Microbenchmark (not product): does it pay to separate TWO read streams in time?  Same access shape as the scan kernel
// (persistent workgroups, tile t of round r = workgroup t % grid, 4 x 16-byte nontemporal loads per lane per wave iteration,
// 2 iterations per tile): (1) one 8 GiB array, (2) two arrays read tile by tile together (what a two-leaf scan does),
// (3) two arrays in alternating phases of T tiles (T tiles of B, then the same T tiles of A).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/two_stream_phases.hip -o gpurun_out/two_stream_phases ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256, kWaves = 4, kUnroll = 4, kIters = 2;
constexpr uint64_t kWaveBytes = 64 * 16 * kUnroll;                     // 4 KiB per wave iteration
constexpr uint64_t kTileBytes = kWaveBytes * kIters * kWaves;           // 32 KiB per workgroup tile

__device__ __forceinline__ uint32_t fold(u32x4 x) { return x.x ^ x.y ^ x.z ^ x.w; }

__device__ __forceinline__ uint32_t read_tile(const uint8_t *base, uint64_t tile, int wave, int lane) {
    uint32_t acc = 0;
    const u32x4 *p = (const u32x4 *)(base + tile * kTileBytes + (uint64_t)wave * kIters * kWaveBytes) + lane;
#pragma unroll
    for (int it = 0; it < kIters; it++) {
        u32x4 r[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) r[u] = __builtin_nontemporal_load(p + it * (kWaveBytes / 16) + u * 64);
#pragma unroll
        for (int u = 0; u < kUnroll; u++) acc ^= fold(r[u]);
    }
    return acc;
}

// mode 0: A only; 1: A and B per tile; 2: phases of T tiles (B first, then A)
__global__ void __launch_bounds__(kBlock) reader(const uint8_t *a, const uint8_t *b, uint64_t n_tiles, int mode, uint32_t T, uint32_t *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    if (mode == 0) {
        for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) acc ^= read_tile(a, t, wave, lane);
    } else if (mode == 1) {
        for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            // both tiles' loads are issued before either is consumed (as the two-leaf scan does)
            const u32x4 *pa = (const u32x4 *)(a + t * kTileBytes + (uint64_t)wave * kIters * kWaveBytes) + lane;
            const u32x4 *pb = (const u32x4 *)(b + t * kTileBytes + (uint64_t)wave * kIters * kWaveBytes) + lane;
#pragma unroll
            for (int it = 0; it < kIters; it++) {
                u32x4 ra[kUnroll], rb[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; u++) ra[u] = __builtin_nontemporal_load(pa + it * (kWaveBytes / 16) + u * 64);
#pragma unroll
                for (int u = 0; u < kUnroll; u++) rb[u] = __builtin_nontemporal_load(pb + it * (kWaveBytes / 16) + u * 64);
#pragma unroll
                for (int u = 0; u < kUnroll; u++) acc ^= fold(ra[u]) + fold(rb[u]);
            }
        }
    } else {
        for (uint64_t t0 = blockIdx.x; t0 < n_tiles; t0 += (uint64_t)gridDim.x * T) {
            for (uint32_t j = 0; j < T; j++) {
                const uint64_t t = t0 + (uint64_t)j * gridDim.x;
                if (t < n_tiles) acc ^= read_tile(b, t, wave, lane);
            }
            for (uint32_t j = 0; j < T; j++) {
                const uint64_t t = t0 + (uint64_t)j * gridDim.x;
                if (t < n_tiles) acc += read_tile(a, t, wave, lane);
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;  // keeps the loads alive
}

int main() {
    const uint64_t bytes = 8ull << 30, n_tiles = bytes / kTileBytes;
    uint8_t *a, *b;
    uint32_t *out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 512;
    auto run = [&](int mode, uint32_t T, const char *name, double gib) {
        std::vector<float> ms;
        for (int i = 0; i < 24; i++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(reader, dim3(grid), dim3(kBlock), 0, 0, a, b, n_tiles, mode, T, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            if (i >= 8) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const float med = ms[ms.size() / 2];
        printf("%-44s %8.4f ms  %7.0f GB/s\n", name, med, gib * 1.073741824 / med * 1e3);
    };
    for (int rep = 0; rep < 2; rep++) {
        run(0, 0, "one stream (8 GiB)", 8);
        run(1, 0, "two streams, tile by tile (16 GiB)", 16);
        run(2, 8, "two streams, phases of 8 tiles", 16);
        run(2, 32, "two streams, phases of 32 tiles", 16);
        run(2, 120, "two streams, phases of 120 tiles", 16);
        run(2, 512, "two streams, phases of 512 tiles (= all)", 16);
    }
    return 0;
}
