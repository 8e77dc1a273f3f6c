// dot_topk.hip -- exact (brute-force) cosine/dot top-k on gfx950.
//
// Replaces the exact search of the reference: `sim = E[q] @ E.T; sim[q] = -inf; torch.topk(sim, k)`
// (inference.py:112-118, utils/evaluation.py:106-132, main.py:226-230, demo.py:150-163).
// Queries are processed in chunks: gather the query rows, one fp32-MFMA GEMM (ps_linear's kernel,
// k-ordered fma chain) into a [chunk, N] similarity slab in the workspace, then one wave per query
// row selects the k best: every lane keeps a sorted k-list of its strided elements in an LDS
// column, and the 64 columns are merged by k wave-wide min-reductions.
// Order: similarity descending, ties by ascending id (torch.topk leaves tie order unspecified).
#include "ps_common.h"

namespace {

constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;

__global__ void gather_rows_kernel(const float *__restrict__ E, int D, const int64_t *__restrict__ qidx, int64_t nq,
                                   float *__restrict__ Q) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq * D; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        Q[i] = E[qidx[r] * D + (i - r * D)];
    }
}

__device__ __forceinline__ uint32_t desc_key(float v) {      // smaller key = larger value
    const uint32_t b = __float_as_uint(v);
    const uint32_t u = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // monotone increasing in v
    return ~u;
}
__device__ __forceinline__ float key_value(uint32_t kk) {
    const uint32_t u = ~kk;
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

// MODE 0: largest similarity first (exact inner-product search).  MODE 1: smallest L2 distance first,
// dist = |q|^2 + |x|^2 - 2 q.x, optionally restricted to the items whose inverted list (assign) is probed.
template <int MODE>
__global__ __launch_bounds__(256) void row_topk_kernel(const float *__restrict__ sims, int64_t N, int64_t rows,
                                                       const int64_t *__restrict__ qidx, int exclude_self, int k, int kcap,
                                                       const float *__restrict__ qn, const float *__restrict__ xn,
                                                       const int32_t *__restrict__ assign,
                                                       const uint32_t *__restrict__ probe, int words,
                                                       float *__restrict__ vals, int64_t *__restrict__ ids) {
    extern __shared__ uint64_t skeys[];   // [4 waves][kcap][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t *col = skeys + (size_t)wv * kcap * 64 + lane;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    if (row >= rows) return;
    const float *s = sims + row * N;
    const int64_t self = exclude_self ? qidx[row] : -1;
    const float qnr = (MODE == 1) ? qn[row] : 0.f;
    // k > kcap (the per-lane columns of one sweep fill the LDS at 32 keys): further sweeps over the row, each admitting only
    // keys AFTER the last one emitted -- keys are unique (id in the low word), so the sweeps partition the order exactly
    uint64_t after = 0;
    bool first = true, dry = false;
    for (int base = 0; base < k; base += kcap) {
        const int kk = (k - base) < kcap ? (k - base) : kcap;
        for (int p = 0; p < kk; ++p) col[p * 64] = EMPTY_KEY;
        uint64_t worst = EMPTY_KEY;
        int filled = 0;
        if (!dry)
            for (int64_t j = lane; j < N; j += 64) {
                float v = s[j];
                if (MODE == 1) {
                    if (assign) {
                        const int32_t a = assign[j];
                        if (!((probe[row * words + (a >> 5)] >> (a & 31)) & 1u)) continue;   // list not probed
                    }
                    v = (qnr + xn[j]) - 2.f * v;
                }
                if (j == self) v = -INFINITY;
                const uint64_t key = ((uint64_t)(MODE == 1 ? ~desc_key(v) : desc_key(v)) << 32) | (uint32_t)j;
                if (!first && key <= after) continue;
                if (key < worst) {
                    int p = filled < kk ? filled : kk - 1;
                    while (p > 0 && col[(p - 1) * 64] > key) { col[p * 64] = col[(p - 1) * 64]; --p; }
                    col[p * 64] = key;
                    if (filled < kk) ++filled;
                    if (filled == kk) worst = col[(kk - 1) * 64];
                }
            }
        int head = 0;
        for (int r = 0; r < kk; ++r) {
            uint64_t mine = head < filled ? col[head * 64] : EMPTY_KEY;
            uint64_t best = mine;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)best, o, 64);
                const uint32_t hi = __shfl_xor((uint32_t)(best >> 32), o, 64);
                const uint64_t other = ((uint64_t)hi << 32) | lo;
                best = other < best ? other : best;
            }
            if (mine == best && best != EMPTY_KEY) ++head;      // keys are unique (id in the low word)
            if (best != EMPTY_KEY) after = best; else dry = true;   // nothing left: the remaining slots are padding
            if (lane == 0) {
                const uint32_t kb = (uint32_t)(best >> 32);
                vals[row * k + base + r] = best != EMPTY_KEY ? key_value(MODE == 1 ? ~kb : kb) : (MODE == 1 ? 3.4028234663852886e38f : -INFINITY);
                ids[row * k + base + r] = best != EMPTY_KEY ? (int64_t)(uint32_t)best : -1;
            }
        }
        first = false;
    }
}

__global__ void row_sqnorm_kernel(const float *__restrict__ x, int64_t n, int D, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nw) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) ss = fmaf(x[i * D + d], x[i * D + d], ss);
        ss = ps_wave_sum_f32(ss);
        if (lane == 0) out[i] = ss;
    }
}

constexpr int TOPK_SWEEP = 32;     // keys per lane and sweep: 4 waves x 32 x 64 lanes x 8 B = 64 KiB of LDS

int64_t chunk_rows(int64_t nq, int64_t N) {
    int64_t c = ((int64_t)1 << 28) / (N > 0 ? N : 1);
    if (c < 64) c = 64;
    return c < nq ? c : nq;
}

}  // namespace

extern "C" size_t ps_dot_topk_workspace_bytes(int64_t nq, int64_t N, int D, int k) {
    if (nq <= 0 || N <= 0) return 256;
    const int64_t c = chunk_rows(nq, N);
    return (size_t)c * N * sizeof(float) + (size_t)c * D * sizeof(float) + 1024;
}

extern "C" int ps_dot_topk(const float *E, int64_t N, int D, const int64_t *qidx, int64_t nq, int k, int exclude_self,
                           float *vals, int64_t *ids, void *workspace, size_t workspace_bytes, ps_stream_t stream) {
    if (N < 0 || D <= 0 || nq < 0 || k <= 0) return PS_EINVAL;
    if (N >= ((int64_t)1 << 32)) return PS_EUNSUPPORTED;
    if (nq == 0) return PS_OK;
    if (!E || !qidx || !vals || !ids || !workspace) return PS_EINVAL;
    if (workspace_bytes < ps_dot_topk_workspace_bytes(nq, N, D, k)) return PS_EWORKSPACE;
    hipStream_t st = ps_stream(stream);
    const int64_t c = chunk_rows(nq, N);
    char *base = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    float *sims = reinterpret_cast<float *>(base);
    float *Q = reinterpret_cast<float *>(base + ((size_t)c * N * sizeof(float) + 255) / 256 * 256);
    const int kcap = k < TOPK_SWEEP ? k : TOPK_SWEEP;
    const size_t lds = (size_t)4 * kcap * 64 * sizeof(uint64_t);
    for (int64_t q0 = 0; q0 < nq; q0 += c) {
        const int64_t rows = (nq - q0) < c ? (nq - q0) : c;
        int64_t g = ps_cdiv(rows * D, 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, E, D, qidx + q0, rows, Q);
        PS_CHECK_LAUNCH();
        const int rc = ps_linear(Q, rows, D, E, D, nullptr, (int)N, nullptr, 0, nullptr, 0, 0, sims, stream);
        if (rc != PS_OK) return rc;
        hipLaunchKernelGGL(row_topk_kernel<0>, dim3((unsigned)ps_cdiv(rows, 4)), dim3(256), lds, st, sims, N, rows,
                           qidx + q0, exclude_self, k, kcap, (const float *)nullptr, (const float *)nullptr,
                           (const int32_t *)nullptr, (const uint32_t *)nullptr, 0, vals + q0 * k, ids + q0 * k);
        PS_CHECK_LAUNCH();
    }
    return PS_OK;
}

extern "C" size_t ps_l2_topk_workspace_bytes(int64_t nq, int64_t N, int D, int k) {
    if (nq <= 0 || N <= 0) return 256;
    const int64_t c = chunk_rows(nq, N);
    return (size_t)c * N * sizeof(float) + (size_t)(N + nq) * sizeof(float) + 2048;
}

extern "C" int ps_l2_topk(const float *X, int64_t N, int D, const float *Q, int64_t nq, int k, const int32_t *assign,
                          const uint32_t *probe, int words, float *dist, int64_t *ids, void *workspace,
                          size_t workspace_bytes, ps_stream_t stream) {
    if (N < 0 || D <= 0 || nq < 0 || k <= 0) return PS_EINVAL;
    if (N >= ((int64_t)1 << 32)) return PS_EUNSUPPORTED;
    if (nq == 0) return PS_OK;
    if (!X || !Q || !dist || !ids || !workspace) return PS_EINVAL;
    if ((assign == nullptr) != (probe == nullptr) || (probe && words <= 0)) return PS_EINVAL;
    if (workspace_bytes < ps_l2_topk_workspace_bytes(nq, N, D, k)) return PS_EWORKSPACE;
    hipStream_t st = ps_stream(stream);
    const int64_t c = chunk_rows(nq, N);
    char *base = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    float *sims = reinterpret_cast<float *>(base);
    float *xn = reinterpret_cast<float *>(base + ((size_t)c * N * sizeof(float) + 255) / 256 * 256);
    float *qn = xn + (N + 63) / 64 * 64;
    int64_t g = ps_cdiv(N, 4);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)g), dim3(256), 0, st, X, N, D, xn);
    PS_CHECK_LAUNCH();
    g = ps_cdiv(nq, 4);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)g), dim3(256), 0, st, Q, nq, D, qn);
    PS_CHECK_LAUNCH();
    const int kcap = k < TOPK_SWEEP ? k : TOPK_SWEEP;
    const size_t lds = (size_t)4 * kcap * 64 * sizeof(uint64_t);
    for (int64_t q0 = 0; q0 < nq; q0 += c) {
        const int64_t rows = (nq - q0) < c ? (nq - q0) : c;
        const int rc = ps_linear(Q + q0 * D, rows, D, X, D, nullptr, (int)N, nullptr, 0, nullptr, 0, 0, sims, stream);
        if (rc != PS_OK) return rc;
        hipLaunchKernelGGL(row_topk_kernel<1>, dim3((unsigned)ps_cdiv(rows, 4)), dim3(256), lds, st, sims, N, rows,
                           (const int64_t *)nullptr, 0, k, kcap, qn + q0, xn, assign, probe ? probe + q0 * words : nullptr, words,
                           dist + q0 * k, ids + q0 * k);
        PS_CHECK_LAUNCH();
    }
    return PS_OK;
}
