// Is the fused-norm quotient (csrc/dense_mfma.hip: y = 1/b correctly rounded, then q = a y and two fma corrections) the IEEE
// quotient a / b?  Brute force over random significands and exponent gaps inside the fast path's domain
// (b in [2^-40, 2^40], a = 0 or |a| in [b 2^-60, b]); every mismatch is counted.  hipcc -O3 -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return (uint32_t)x;
}
__global__ void k(unsigned long long *bad, unsigned long long *first, int rounds, uint64_t seed, int hard) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (int it = 0; it < rounds; ++it) {
        const uint64_t s = seed + (tid * rounds + it) * 0x9E3779B97F4A7C15ULL;
        const uint32_t r0 = mix(s), r1 = mix(s ^ 0xabcdef1234567ULL), r2 = mix(s + 77);
        const int eb = 127 - 40 + (int)(r2 % 81);                       // b in [2^-40, 2^40]
        const float b = __uint_as_float((uint32_t)eb << 23 | (r0 & 0x7fffff));
        const int gap = (r2 >> 8) % 4 == 0 ? (int)((r2 >> 12) % 60) : (int)((r2 >> 12) % 3);   // mostly near b, sometimes far below
        float a = __uint_as_float((uint32_t)(eb - gap) << 23 | (r1 & 0x7fffff) | (r2 & 0x80000000u));
        if (fabsf(a) > b) a = __uint_as_float(__float_as_uint(a) - (1u << 23));
        if (hard) {      // a / b as close to a rounding boundary (midpoint of two floats) as this b allows: A = floor or ceil of M B / 2^k
            const uint64_t B = 0x800000u | (r0 & 0x7fffff), M = 2 * (uint64_t)(0x800000u | (r1 & 0x7fffff)) + 1, P = M * B;
            const int sh = (P >> 48) ? 25 : 24;                          // keep A in [2^23, 2^24)
            uint64_t A = (P >> sh) + ((r2 >> 30) & 1);
            if (A >= (1u << 24) || A < (1u << 23)) A = 0x800000u | (r1 & 0x7fffff);
            a = __uint_as_float((uint32_t)(eb - 1 - (int)((r2 >> 12) % 3)) << 23 | ((uint32_t)A & 0x7fffff) | (r2 & 0x80000000u));
        }
        const float y = 1.0f / b;
        const float q0 = a * y, e0 = fmaf(-b, q0, a), q1 = fmaf(e0, y, q0), e1 = fmaf(-b, q1, a), q2 = fmaf(e1, y, q1);
        const float want = a / b;
        if (__float_as_uint(q2) != __float_as_uint(want)) {
            if (nb == 0) { first[2 * tid] = __float_as_uint(a); first[2 * tid + 1] = __float_as_uint(b); }
            ++nb;
        }
    }
    if (nb) atomicAdd(bad, nb);
}
int main() {
    unsigned long long *bad, *first, h = 0;
    const int blocks = 4096, threads = 256, rounds = 4096;
    hipMalloc(&bad, 8); hipMalloc(&first, (size_t)blocks * threads * 16);
    hipMemset(bad, 0, 8); hipMemset(first, 0, (size_t)blocks * threads * 16);
    for (int rep = 0; rep < 8; ++rep) k<<<blocks, threads>>>(bad, first, rounds, 0x1234567ULL + rep * 0x1000003ULL, 0);
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("fused-norm quotient vs a / b: %llu mismatches in %.3g random pairs\n", h, 8.0 * blocks * threads * rounds);
    for (int rep = 0; rep < 8; ++rep) k<<<blocks, threads>>>(bad, first, rounds, 0x7654321ULL + rep * 0x1000003ULL, 1);
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("                              %llu mismatches after %.3g more pairs next to rounding boundaries\n", h, 8.0 * blocks * threads * rounds);
    if (h) {
        static unsigned long long f[4096 * 256 * 2];
        hipMemcpy(f, first, sizeof f, hipMemcpyDeviceToHost);
        int shown = 0;
        for (size_t i = 0; i < 4096 * 256 && shown < 5; ++i) if (f[2 * i + 1]) { printf("  a=0x%08llx b=0x%08llx\n", f[2 * i], f[2 * i + 1]); ++shown; }
    }
    return h != 0;
}
