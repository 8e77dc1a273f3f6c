// Instruction-rate probe: v_bcnt_u32_b32 vs v_xor_b32 vs v_add_u32 on gfx950 (one-off experiment).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    uint32_t x = seed ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_popcount(a[i] ^ x) + a[i];        // xor + bcnt(acc)
            if (OP == 1) a[i] = (a[i] ^ x) + (a[i] >> 1);                    // xor + shift-add (full-rate ops)
            if (OP == 2) { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 3) { asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 4) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int nops_per_iter) {
    uint32_t *d; hipMalloc(&d, 4096 * 256 * 4);
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(d, iters, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, iters, 12345u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double waves = blocks * 4.0, instr = waves * iters * 8.0 * nops_per_iter;
    double per_simd_per_ns = instr / (ms * 1e6) / 1024.0;
    printf("%-28s %.3f ms  -> %.2f wave-instr/ns/SIMD  (%.2f cycles/instr @2.4GHz)\n", name, ms, per_simd_per_ns, 2.4 / per_simd_per_ns);
    hipFree(d);
}
int main() {
    run<2>("v_bcnt_u32_b32 (asm)", 1);
    run<3>("v_xor_b32 (asm)", 1);
    run<4>("v_add_u32 (asm)", 1);
    run<0>("xor+bcnt (C)", 2);
    run<1>("xor+shift+add (C)", 3);
    return 0;
}
