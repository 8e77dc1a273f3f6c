// importance_pool.hip -- weighted sparse neighbour gather on gfx950 (HBM/L2-bound row gather).
//
// Replaces ImportancePooling.forward (reference model/pinsage.py:101-150) and the gather +
// weighted-reduce part of WeightedAggregator / MeanAggregator / ImportanceAggregator
// (model/aggregators.py:13-91, 233-287).  One wave per output row; every neighbour row is
// fetched as one coalesced 16 B-per-lane sweep (H = 256 floats = one 1 KiB wave instruction),
// up to UNROLL rows in flight; accumulation in fp32 in neighbour order.
#include "ps_common.h"

namespace {

constexpr int UNROLL = 8;

template <int VEC>   // floats per lane per chunk (4 -> float4 path, 1 -> scalar path)
__global__ __launch_bounds__(256) void importance_pool_kernel(const float *__restrict__ x, int H,
                                                              const int32_t *__restrict__ ids,
                                                              const int32_t *__restrict__ counts,
                                                              const float *__restrict__ wts,
                                                              const int32_t *__restrict__ nvalid, int64_t B, int T,
                                                              int64_t max_idx, int renorm,
                                                              float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const int chunk = 64 * VEC;
    for (int64_t i = wave; i < B; i += nwaves) {
        // the row's metadata is requested at once (T <= 64: one id / count / weight per lane, kept in registers), so the
        // chain before the row gathers is one global round trip, not one per quantity
        const bool small = T <= 64;
        int32_t id0 = -1, cnt0 = 0;
        float wt0 = 0.f;
        if (small && lane < T) {
            id0 = ids[i * T + lane];
            if (counts) cnt0 = counts[i * T + lane];
            if (wts) wt0 = wts[i * T + lane];
        }
        const int k = __builtin_amdgcn_readfirstlane(nvalid[i]);
        // ---- weights: total of the kept counts (reference: weights = count / sum(top counts)) ----
        int tot = 0;
        if (small) {
            tot = lane < k ? cnt0 : 0;
        } else {
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int j = j0 + lane;
                tot += (counts && j < k) ? counts[i * T + j] : 0;
            }
        }
        tot = ps_wave_sum_i32(tot);
        float wsum = 0.f;
        if (small) {
            if (lane < k && id0 >= 0 && (int64_t)id0 <= max_idx) wsum = wts ? wt0 : (float)((double)cnt0 / (double)tot);
        } else {
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int j = j0 + lane;
                float wj = 0.f;
                if (j < k) {
                    const int32_t id = ids[i * T + j];
                    if (id >= 0 && (int64_t)id <= max_idx)
                        wj = wts ? wts[i * T + j] : (float)((double)counts[i * T + j] / (double)tot);
                }
                wsum += wj;
            }
        }
        wsum = ps_wave_sum_f32(wsum);
        const bool do_norm = renorm && wsum > 0.f;        // w /= w.sum() only if sum > 0 (pinsage.py:141-143)
        // ---- gather + reduce ----
        for (int c0 = 0; c0 < H; c0 += chunk) {
            const int col = c0 + lane * VEC;
            const bool cact = col < H;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int jj = j0 + lane;
                int32_t myid = -1;
                float myw = 0.f;
                if (jj < k) {
                    myid = small ? id0 : ids[i * T + jj];
                    if (myid >= 0 && (int64_t)myid <= max_idx) {
                        if (small) myw = wts ? wt0 : (float)((double)cnt0 / (double)tot);
                        else myw = wts ? wts[i * T + jj] : (float)((double)counts[i * T + jj] / (double)tot);
                        if (do_norm) myw = myw / wsum;
                    } else {
                        myid = -1;
                    }
                }
                const int kk = (k - j0) < 64 ? (k - j0) : 64;
                for (int t0 = 0; t0 < kk; t0 += UNROLL) {
                    float r[UNROLL][VEC];
                    float wv[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int t = t0 + u;
                        const int32_t id = (t < kk) ? __builtin_amdgcn_readlane(myid, t < 64 ? t : 0) : -1;
                        wv[u] = (t < kk) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myw), t < 64 ? t : 0)) : 0.f;
                        const bool ok = id >= 0 && cact;
                        if (VEC == 4) {
                            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (ok) q = *reinterpret_cast<const float4 *>(x + (int64_t)id * H + col);
                            r[u][0] = q.x; r[u][1 % VEC] = q.y; r[u][2 % VEC] = q.z; r[u][3 % VEC] = q.w;
                        } else {
                            r[u][0] = ok ? x[(int64_t)id * H + col] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(r[u][v], wv[u], acc[v]);
                }
            }
            if (cact) {
                if (VEC == 4) {
                    *reinterpret_cast<float4 *>(out + i * H + col) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
                } else {
                    out[i * H + col] = acc[0];
                }
            }
        }
    }
}

}  // namespace

extern "C" int ps_importance_pool(const float *x, int64_t N, int H, const int32_t *ids, const int32_t *counts,
                                  const float *wts, const int32_t *nvalid, int64_t B, int T, int64_t max_idx,
                                  int renorm, float *out, ps_stream_t stream) {
    if (B < 0 || H <= 0 || T <= 0 || N < 0) return PS_EINVAL;
    if (B == 0) return PS_OK;
    if (!x || !ids || !nvalid || !out || (!counts && !wts)) return PS_EINVAL;
    if (max_idx > N - 1) max_idx = N - 1;
    int64_t grid = ps_cdiv(B, 4);
    if (grid > 256 * 16) grid = 256 * 16;
    hipStream_t st = ps_stream(stream);
    const bool vec4 = (H % 4 == 0) && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(out)) % 16 == 0);
    if (vec4)
        hipLaunchKernelGGL(importance_pool_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, x, H, ids, counts, wts,
                           nvalid, B, T, max_idx, renorm, out);
    else
        hipLaunchKernelGGL(importance_pool_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, x, H, ids, counts, wts,
                           nvalid, B, T, max_idx, renorm, out);
    PS_CHECK_LAUNCH();
    return PS_OK;
}
