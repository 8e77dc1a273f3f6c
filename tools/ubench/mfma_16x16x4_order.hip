// Does v_mfma_f32_16x16x4_f32 accumulate its four k in ascending order, one fma after the other (acc = fma(a_k, b_k, acc), k = 0..3)?
// The GEMMs of csrc/dense_mfma.hip rest on v_mfma_f32_32x32x2_f32 being exactly that chain (two k per instruction); a 16-row tile
// for small shards would need the same property of the 16 x 16 x 4 shape.  Compares one 16 x 16 output block over K = 64 with the
// host's fmaf chain in ascending k, and with the other plausible orders, bit for bit.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_16x16x4_order.hip -o tools/ubench/mfma_16x16x4_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 64;
// A [16][K], B [K][16] -> C [16][16]; lane l: row/col i = l & 15, k block = l >> 4 (k = 4 s + (l >> 4) for step s)
__global__ void k(const float *A, const float *B, float *C) {
    const int l = threadIdx.x, i = l & 15, kb = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + 4 * s + kb], B[(4 * s + kb) * 16 + i], acc, 0, 0, 0);
    // C layout: col = l & 15, rows 4 * (l >> 4) + r
    for (int r = 0; r < 4; ++r) C[(4 * kb + r) * 16 + i] = acc[r];
}
int main() {
    std::vector<float> A(16 * K), B(K * 16), C(256);
    srand(7);
    int bad_asc = 0, bad_desc = 0, bad_pair = 0, bad_tree = 0, trials = 200;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    for (int t = 0; t < trials; ++t) {
        for (auto &x : A) x = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
        for (auto &x : B) x = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                float asc = 0.f, desc = 0.f, pair = 0.f, tree = 0.f;
                for (int s = 0; s < K / 4; ++s) {
                    const float *a = &A[r * K + 4 * s];
                    auto b = [&](int j) { return B[(4 * s + j) * 16 + c]; };
                    for (int j = 0; j < 4; ++j) asc = fmaf(a[j], b(j), asc);
                    for (int j = 3; j >= 0; --j) desc = fmaf(a[j], b(j), desc);
                    pair = fmaf(a[1], b(1), fmaf(a[0], b(0), pair)); pair = fmaf(a[3], b(3), fmaf(a[2], b(2), pair));   // = asc; kept as a control
                    tree = tree + (fmaf(a[1], b(1), a[0] * b(0)) + fmaf(a[3], b(3), a[2] * b(2)));
                }
                const float got = C[r * 16 + c];
                bad_asc += memcmp(&got, &asc, 4) != 0; bad_desc += memcmp(&got, &desc, 4) != 0;
                bad_pair += memcmp(&got, &pair, 4) != 0; bad_tree += memcmp(&got, &tree, 4) != 0;
            }
    }
    printf("v_mfma_f32_16x16x4_f32 over K = %d, %d random blocks x 256 outputs: mismatches vs fmaf chain k ascending %d, k descending %d, "
           "(control) %d, pairwise tree %d\n", K, trials, bad_asc, bad_desc, bad_pair, bad_tree);
    return 0;
}
