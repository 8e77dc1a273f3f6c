// Random-sector read rate of the memory system, the bound under the walk sampler's later steps: every lane reads 16 bytes of ITS
// OWN random 64-byte row (one sector per lane and load, like a bucket-record fetch), U independent loads in flight per lane, 8 waves
// per SIMD.  Two access patterns: independent (addresses from a hash: the memory system's throughput) and dependent (the next row
// is a function of the loaded data: a chain, like walk steps).  Table sizes from L2-resident to the sampler's 1.6 GB of bucket records.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/gather_rate.hip -o tools/ubench/gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int U, bool DEP>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ table, uint32_t rows, int iters, uint32_t *out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t r[U], acc = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = mix(tid * U + u + 1);
    for (int it = 0; it < iters; ++it) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = table[(size_t)(r[u] % rows) * 4 + (threadIdx.x & 3)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc += v[u].x;
            r[u] = DEP ? mix(r[u] + v[u].y) : mix(r[u] + it + 1);
        }
    }
    out[tid] = acc;
}

template <int U, bool DEP>
void run(const uint4 *table, uint32_t rows, uint32_t *out, const char *name) {
    const int blocks = 256 * 8, iters = 256;                       // 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<U, DEP>), dim3(blocks), dim3(256), 0, 0, table, rows, iters, out);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<U, DEP>), dim3(blocks), dim3(256), 0, 0, table, rows, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double n = 5.0 * blocks * 256.0 * iters * U;
    printf("  %-34s %7.1f G sectors/s = %5.2f TB/s of 64-byte sectors\n", name, n / ms / 1e6, n * 64 / ms / 1e9);
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    const double sizes_gb[] = {0.002, 0.03, 0.2, 1.6, 4.8, 16.0};
    for (double gb : sizes_gb) {
        const uint32_t rows = (uint32_t)(gb * (1ull << 30) / 64);
        uint4 *table;
        if (hipMalloc(&table, (size_t)rows * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(table, 1, (size_t)rows * 64);
        printf("table %.3f GB\n", gb);
        run<1, false>(table, rows, out, "1 load in flight per lane");
        run<4, false>(table, rows, out, "4 loads in flight per lane");
        run<1, true>(table, rows, out, "dependent chain, 1 per lane");
        run<4, true>(table, rows, out, "dependent chains, 4 per lane");
        hipFree(table);
    }
    return 0;
}
