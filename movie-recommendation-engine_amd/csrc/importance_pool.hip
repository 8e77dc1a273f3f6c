// importance_pool.hip -- weighted sparse neighbour gather on gfx950 (HBM/L2-bound row gather).
//
// Replaces ImportancePooling.forward (reference model/pinsage.py:101-150) and the gather +
// weighted-reduce part of WeightedAggregator / MeanAggregator / ImportanceAggregator
// (model/aggregators.py:13-91, 233-287).  Two kernels: importance_pool4_kernel (T <= 64, 16-byte aligned rows: four output
// rows per wave, below) and importance_pool_kernel (any shape: one wave per output row; every neighbour row one coalesced
// 16 B-per-lane sweep, up to UNROLL rows in flight); accumulation in fp32 in neighbour order in both.
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

// ---- four output rows per wave (r04) ------------------------------------------------------------------------------------
// On the catalogue graphs most of a start item's top-T neighbours are USERS, which the reference drops when x holds item rows only
// (model/pinsage.py:123-129): SYN-25M at T = 10 keeps 0.85 neighbours per row on average, so the one-wave-per-row kernel above spent a
// whole wave and two dependent round trips on copying <= 1 row.  Here a 16-lane group owns an output row (16 lanes x 16 B = 256 B per
// sweep, four sweeps per 1 KiB row), a wave four rows: their ids / counts come in ONE round trip (entry e of a row in lane e & 15 of
// its group, T <= 16 * PAGES), the per-row sums are DPP reductions over the group's 16 lanes (quad_perm, row_half_mirror,
// row_mirror -- for T <= 16 the same butterfly as ps_wave_sum over a wave whose other lanes hold zeros: identical bits), an entry is
// broadcast to its group by DPP row_newbcast, and the gathers of four entries x four sweeps x four rows are in flight together
// (16 KiB per wave; masked lanes request nothing, so dropped neighbours cost an instruction slot, not traffic).  Rows that keep no
// neighbour are stored as zeros without touching x.  Same operation order per output element as the kernel above (fp32 fma in
// neighbour order).
template <int CTRL>
__device__ __forceinline__ int row_bcast_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row16_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    return v;
}
__device__ __forceinline__ float row16_sum_f32(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

template <int PAGES>
__global__ __launch_bounds__(256) void importance_pool4_kernel(const float *__restrict__ x, int H, const int32_t *__restrict__ ids,
                                                               const int32_t *__restrict__ counts, const float *__restrict__ wts,
                                                               const int32_t *__restrict__ nvalid, int64_t B, int T, int64_t max_idx,
                                                               int renorm, float *__restrict__ out) {
    constexpr int TU = 4;                                     // entries gathered per batch
    const int lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i0 = wave0 * 4; i0 < B; i0 += nwaves * 4) {
        const int64_t i = i0 + g;
        const bool rok = i < B;
        // ---- one round trip: nvalid, ids, counts / weights of the wave's four rows ----
        int k = rok ? nvalid[i] : 0;
        int32_t id[PAGES], cn[PAGES];
        float w[PAGES];
#pragma unroll
        for (int p = 0; p < PAGES; ++p) {
            const int e = p * 16 + l;
            const bool in = rok && e < T;
            id[p] = in ? ids[i * T + e] : -1;
            cn[p] = (in && counts) ? counts[i * T + e] : 0;
            w[p] = (in && wts) ? wts[i * T + e] : 0.f;
        }
        k = k < T ? k : T;
        // ---- weights: count / sum of the row's kept counts in fp64 -> fp32 (utils/random_walk.py:113-115 -> pinsage.py:140),
        // ids beyond max_idx dropped (:123-129), renormalised by their fp32 sum when it is positive (:141-143) ----
        int tot = 0;
#pragma unroll
        for (int p = 0; p < PAGES; ++p) tot += (p * 16 + l < k) ? cn[p] : 0;
        tot = row16_sum_i32(tot);
        float wsum = 0.f;
#pragma unroll
        for (int p = 0; p < PAGES; ++p) {
            const bool keep = (p * 16 + l < k) && id[p] >= 0 && (int64_t)id[p] <= max_idx;
            w[p] = keep ? (wts ? w[p] : (float)((double)cn[p] / (double)tot)) : 0.f;
            id[p] = keep ? id[p] : -1;
            wsum += w[p];
        }
        wsum = row16_sum_f32(wsum);
        if (renorm && wsum > 0.f) {
#pragma unroll
            for (int p = 0; p < PAGES; ++p) w[p] = id[p] >= 0 ? w[p] / wsum : 0.f;
        }
        // entries to walk: the largest k of the four rows (wave-uniform)
        int kmax = __builtin_amdgcn_readlane(k, 0);
        { const int k1 = __builtin_amdgcn_readlane(k, 16), k2 = __builtin_amdgcn_readlane(k, 32), k3 = __builtin_amdgcn_readlane(k, 48);
          kmax = kmax > k1 ? kmax : k1; kmax = kmax > k2 ? kmax : k2; kmax = kmax > k3 ? kmax : k3; }
        // ---- gather + reduce: 256 columns (four sweeps) at a time ----
        for (int cb = 0; cb < H; cb += 256) {
            float4 acc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
#define PS_POOL_ENTRY(P, TT, U)                                                                            \
            {                                                                                              \
                nid[U] = row_bcast_i32<0x150 + (TT)>(id[P]);                                               \
                nw[U] = __builtin_bit_cast(float, row_bcast_i32<0x150 + (TT)>(__builtin_bit_cast(int, w[P]))); \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                            \
                    const int col = cb + s * 64 + l * 4;                                                   \
                    r[U][s] = make_float4(0.f, 0.f, 0.f, 0.f);                                             \
                    if (nid[U] >= 0 && col < H) r[U][s] = *reinterpret_cast<const float4 *>(x + (int64_t)nid[U] * H + col); \
                }                                                                                          \
            }
#define PS_POOL_BATCH(P, T0)                                                                               \
            if ((P) * 16 + (T0) < kmax) {                                                                  \
                int32_t nid[TU];                                                                           \
                float nw[TU];                                                                              \
                float4 r[TU][4];                                                                           \
                PS_POOL_ENTRY(P, (T0) + 0, 0) PS_POOL_ENTRY(P, (T0) + 1, 1) PS_POOL_ENTRY(P, (T0) + 2, 2) PS_POOL_ENTRY(P, (T0) + 3, 3) \
                _Pragma("unroll") for (int u = 0; u < TU; ++u)                                             \
                    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                        \
                        acc[s].x = fmaf(r[u][s].x, nw[u], acc[s].x); acc[s].y = fmaf(r[u][s].y, nw[u], acc[s].y); \
                        acc[s].z = fmaf(r[u][s].z, nw[u], acc[s].z); acc[s].w = fmaf(r[u][s].w, nw[u], acc[s].w); \
                    }                                                                                      \
            }
#define PS_POOL_PAGE(P) PS_POOL_BATCH(P, 0) PS_POOL_BATCH(P, 4) PS_POOL_BATCH(P, 8) PS_POOL_BATCH(P, 12)
            PS_POOL_PAGE(0)
            if constexpr (PAGES > 1) { PS_POOL_PAGE(1) }
            if constexpr (PAGES > 2) { PS_POOL_PAGE(2) }
            if constexpr (PAGES > 3) { PS_POOL_PAGE(3) }
#undef PS_POOL_PAGE
#undef PS_POOL_BATCH
#undef PS_POOL_ENTRY
            if (rok) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int col = cb + s * 64 + l * 4;
                    if (col < H) *reinterpret_cast<float4 *>(out + i * H + col) = acc[s];
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
    const char *pe = getenv("PS_POOL_ROWS_PER_WAVE");         // 1: the one-row-per-wave kernel (experiments / cross-check)
    const bool four = vec4 && T <= 64 && !(pe && atoi(pe) == 1);
    if (four) {
        int64_t g4 = ps_cdiv(B, 16);                           // 4 waves x 4 rows per 256-thread workgroup
        if (g4 > 256 * 64) g4 = 256 * 64;
        if (T <= 16)
            hipLaunchKernelGGL(importance_pool4_kernel<1>, dim3((unsigned)g4), dim3(256), 0, st, x, H, ids, counts, wts, nvalid, B, T,
                               max_idx, renorm, out);
        else
            hipLaunchKernelGGL(importance_pool4_kernel<4>, dim3((unsigned)g4), dim3(256), 0, st, x, H, ids, counts, wts, nvalid, B, T,
                               max_idx, renorm, out);
    } else if (vec4)
        hipLaunchKernelGGL(importance_pool_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, x, H, ids, counts, wts,
                           nvalid, B, T, max_idx, renorm, out);
    else
        hipLaunchKernelGGL(importance_pool_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, x, H, ids, counts, wts,
                           nvalid, B, T, max_idx, renorm, out);
    PS_CHECK_LAUNCH();
    return PS_OK;
}
