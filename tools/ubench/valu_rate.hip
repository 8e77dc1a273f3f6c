// Instruction-rate probe for the popcount scan's bound: v_bcnt_u32_b32 / v_xor_b32 / v_add_u32 issue rates on gfx950,
// after >= 0.5 s of settle launches, for 1 / 2 / 4 / 8 waves per SIMD, with the in-kernel shader clock
// (delta s_memtime / delta s_memrealtime x 100 MHz, median over waves) printed beside every figure.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint64_t *stamps, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    uint32_t x = seed ^ threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 1) { asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 2) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 3) { asm volatile("v_xor_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x)); }
            if (OP == 4) { asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 5) { asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            if (OP == 6) {                                   // full 32 x 32 -> 64 product + 64-bit addend in ONE instruction (Philox rounds)
                uint64_t w = ((uint64_t)a[i] << 32) | a[(i + 1) & 7];
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(x), "v"(a[i]) : "vcc");
                a[i] = (uint32_t)(w >> 32) ^ (uint32_t)w;
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP>
void run(const char *name, int nops, int waves_per_simd) {
    const int iters = 8192;
    const int blocks = 256 * waves_per_simd;               // 256 CUs x waves_per_simd blocks of 4 waves = w waves per SIMD
    uint32_t *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    uint64_t *st; hipMalloc(&st, (size_t)blocks * 4 * 2 * 8);
    auto t_start = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < 0.6) {   // settle the clock
        for (int i = 0; i < 8; ++i) k<OP><<<blocks, 256>>>(d, st, iters, 12345u);
        hipDeviceSynchronize();
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    for (int rep = 0; rep < 9; ++rep) {
        hipEventRecord(a);
        k<OP><<<blocks, 256>>>(d, st, iters, 12345u);
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<uint64_t> h((size_t)blocks * 4 * 2);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (size_t w = 0; w < (size_t)blocks * 4; ++w) {
        if (h[2 * w + 1] > 0) clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);   // GHz
        cyc.push_back((double)h[2 * w]);
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double instr_per_wave = (double)iters * 8.0 * nops;
    // issue interval seen by one SIMD = wave cycles x (1 / waves per SIMD) per instruction of one wave
    const double cyc_per_instr_simd = cyc[cyc.size() / 2] / instr_per_wave / waves_per_simd;
    const double total = (double)blocks * 4.0 * instr_per_wave;
    printf("%-22s %d waves/SIMD  median %.4f ms  %.3f wave-instr/ns/SIMD  in-kernel clock %.3f GHz  -> %.2f shader cycles per wave-instruction per SIMD\n",
           name, waves_per_simd, ms[ms.size() / 2], total / (ms[ms.size() / 2] * 1e6) / 1024.0, ghz, cyc_per_instr_simd);
    hipFree(d); hipFree(st);
}

int main() {
    for (int w : {1, 4}) {                                   // integer multiplies (the walk sampler's Philox rounds)
        run<4>("v_mul_lo_u32", 1, w);
        run<5>("v_mul_hi_u32", 1, w);
        run<6>("v_mad_u64_u32 (+3 moves)", 1, w);
    }
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_bcnt_u32_b32", 1, w);
        run<1>("v_xor_b32", 1, w);
        run<2>("v_add_u32", 1, w);
        run<3>("v_xor + v_bcnt pair", 2, w);
    }
    return 0;
}
