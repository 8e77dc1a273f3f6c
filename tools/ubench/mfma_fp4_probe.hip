// Probe for a +1/-1 contraction on the block-scaled fp4 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, cbsz = blgp = 4,
// E8M0 scale 127 = 1.0): (1) exactness against an integer dot product with the sign-plane lane layout the Hamming scan
// uses (lane l = row l & 31, K half l >> 5, 32 nibbles per lane: +1 = 0x2, -1 = 0xA, padding 0x0), (2) its issue rate
// next to v_mfma_i32_32x32x32_i8 on random sign data.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_fp4_probe.hip -o tools/ubench/mfma_fp4_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void check_kernel(const v4i *a, const v4i *b, float *out) {
    const int lane = threadIdx.x & 63;
    const v4i av = a[lane], bv = b[lane];
    v8i A = {av[0], av[1], av[2], av[3], 0, 0, 0, 0};
    v8i B = {bv[0], bv[1], bv[2], bv[3], 0, 0, 0, 0};
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

template <int FP4>
__global__ __launch_bounds__(512) void rate_kernel(const v4i *a, const v4i *b, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    v4i av[8], bv[8];
    for (int s = 0; s < 8; ++s) { av[s] = a[(s * 64 + lane) % 4096]; bv[s] = b[(s * 64 + lane + threadIdx.x / 64) % 4096]; }
    v16f cf; v16i ci;
    for (int i = 0; i < 16; ++i) { cf[i] = 0.f; ci[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (FP4) {
                v8i A = {av[s][0], av[s][1], av[s][2], av[s][3], 0, 0, 0, 0};
                v8i B = {bv[s][0], bv[s][1], bv[s][2], bv[s][3], 0, 0, 0, 0};
                cf = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, cf, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else {
                ci = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s], bv[s], ci, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += FP4 ? cf[i] : (float)ci[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    // ---- exactness ----
    std::vector<int8_t> sa(32 * 64), sb(32 * 64);           // sa[row][k], sb[col][k] in {-1, +1}
    srand(7);
    for (auto &v : sa) v = (rand() & 1) ? 1 : -1;
    for (auto &v : sb) v = (rand() & 1) ? 1 : -1;
    for (int k = 50; k < 64; ++k) sa[5 * 64 + k] = 0;       // a few zeros (padding nibbles)
    auto pack = [](const std::vector<int8_t> &s) {
        std::vector<uint8_t> p(64 * 16, 0);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 32; ++j) {
                const int8_t v = s[(lane & 31) * 64 + 32 * (lane >> 5) + j];
                const uint8_t nib = v > 0 ? 0x2 : v < 0 ? 0xA : 0x0;
                p[lane * 16 + j / 2] |= nib << (4 * (j & 1));
            }
        return p;
    };
    auto pa = pack(sa), pb = pack(sb);
    v4i *da, *db; float *dout;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, pa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, pb.data(), 1024, hipMemcpyHostToDevice);
    check_kernel<<<1, 64>>>(da, db, dout);
    std::vector<float> out(64 * 16);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            int dot = 0;
            for (int k = 0; k < 64; ++k) dot += sa[row * 64 + k] * sb[col * 64 + k];
            if (out[lane * 16 + r] != (float)dot) { if (bad < 5) printf("mismatch lane %d r %d: %f vs %d\n", lane, r, out[lane * 16 + r], dot); ++bad; }
        }
    printf("fp4 +1/-1 contraction vs integer dot (A = rows, B = columns, C col = lane & 31): %s (%d mismatches of 1024)\n", bad ? "WRONG" : "exact", bad);
    // ---- rate ----
    std::vector<uint8_t> ra(4096 * 16), rb(4096 * 16), ia(4096 * 16), ib(4096 * 16);
    for (size_t i = 0; i < ra.size(); ++i) {
        ra[i] = ((rand() & 1) ? 0x2 : 0xA) | (((rand() & 1) ? 0x2 : 0xA) << 4); rb[i] = ((rand() & 1) ? 0x2 : 0xA) | (((rand() & 1) ? 0x2 : 0xA) << 4);
        ia[i] = (rand() & 1) ? 0x01 : 0xFF; ib[i] = (rand() & 1) ? 0x01 : 0xFF;
    }
    v4i *xa, *xb; float *o2;
    hipMalloc(&xa, ra.size()); hipMalloc(&xb, rb.size()); hipMalloc(&o2, 256 * 512 * 4);
    for (int fp4 = 1; fp4 >= 0; --fp4) {
        hipMemcpy(xa, fp4 ? ra.data() : ia.data(), ra.size(), hipMemcpyHostToDevice);
        hipMemcpy(xb, fp4 ? rb.data() : ib.data(), rb.size(), hipMemcpyHostToDevice);
        const int iters = 4096;
        auto launch = [&]() { if (fp4) rate_kernel<1><<<256, 512>>>(xa, xb, o2, iters); else rate_kernel<0><<<256, 512>>>(xa, xb, o2, iters); };
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.6) { launch(); hipDeviceSynchronize(); }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double mfma = 256.0 * 8 * iters * 8;                      // 8 waves per CU (2 per SIMD), 8 MFMAs per iteration
        const double ops = mfma * 2.0 * 32 * 32 * (fp4 ? 64 : 32);
        printf("%s: %.3f ms, %.1f ns per MFMA per SIMD (2 waves per SIMD, one chain each), %.0f TOP/s of sign products\n",
               fp4 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (fp4)" : "v_mfma_i32_32x32x32_i8", ms, ms * 1e6 / (mfma / 1024.0), ops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
