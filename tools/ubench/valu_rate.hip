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
            if (OP == 5) { asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 6) { asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 7) { uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a[i]), "v"(x) : "vcc"); a[i] = (uint32_t)r ^ (uint32_t)(r >> 32); }
            if (OP == 8) { asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
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
    run<5>("v_mul_lo_u32 (asm)", 1);
    run<6>("v_mul_hi_u32 (asm)", 1);
    run<7>("v_mad_u64_u32 + xor", 2);
    run<8>("v_mul_u32_u24 (asm)", 1);
    run<0>("xor+bcnt (C)", 2);
    run<1>("xor+shift+add (C)", 3);
    return 0;
}
