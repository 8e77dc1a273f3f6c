// Issue-rate probe: v_mfma_f32_32x32x2_f32 with 4 independent accumulators per wave (one-off experiment).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = seed * threadIdx.x, bv = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu) {
    float *d; hipMalloc(&d, 256 * 16 * 256 * 4);
    const int iters = 2048, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC><<<blocks, 256>>>(d, iters, 1e-3f); hipDeviceSynchronize();
    hipEventRecord(a); k<NACC><<<blocks, 256>>>(d, iters, 1e-3f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double mf = (double)blocks * 4 * iters * 4 * NACC, fl = mf * 4096.0;
    printf("NACC=%d blocks/CU=%d: %.3f ms -> %.1f TFLOP/s, %.1f cycles per MFMA per SIMD @2.4GHz\n", NACC, blocks_per_cu, ms,
           fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024.0));
    hipFree(d);
}
int main() { run<4>(1); run<4>(2); run<2>(2); run<1>(2); run<1>(4); return 0; }
