// spmm_csr.hip -- CSR row gather-reduce on gfx950: out[r] = sum_{e in row r} val[e] * x[col[e]].
//
// The aggregation of GraphConv.propagate (reference model/pinsage.py:53-54, 70-92: PyG MessagePassing with
// aggr='add', flow source->target): rows = target nodes, col = source nodes, val = edge_weight *
// importance_weight (1 if absent).  Work is cut by EDGES, not rows (rating graphs have 10^5-edge hubs): wave w
// owns the edge slots [w*SLICE, (w+1)*SLICE), finds its first row by bisection of rowptr and walks the rows that
// intersect its range.  Every source row is one coalesced 16 B-per-lane sweep, 8 rows in flight.  A row that lies
// completely inside one slice is stored; a row cut by a slice boundary adds its partial sums with fp32 atomics
// (PyG's scatter-add is atomic too); out is zeroed first (empty rows, atomics).
#include "ps_common.h"

namespace {

constexpr int SLICE = 512;
constexpr int UNROLL = 8;

template <int VEC>
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       const float *__restrict__ val, const float *__restrict__ x, int64_t N, int H,
                                                       int64_t V, int64_t E, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int chunk = 64 * VEC;
    int64_t e_begin = wave * SLICE;
    if (e_begin >= E) return;
    const int64_t e_end = (e_begin + SLICE < E) ? (e_begin + SLICE) : E;
    // r = last row with rowptr[r] <= e_begin
    int64_t a = 0, b = V;                                    // invariant: rowptr[a] <= e_begin < rowptr[b]
    while (b - a > 1) {
        const int64_t m = (a + b) >> 1;
        if (rowptr[m] <= e_begin) a = m; else b = m;
    }
    int64_t r = a;
    while (e_begin < e_end) {
        const int64_t lo = rowptr[r], hi = rowptr[r + 1];
        const int64_t s0 = e_begin, s1 = (hi < e_end) ? hi : e_end;
        if (s1 <= s0) { ++r; continue; }
        const bool whole = (s0 == lo) && (s1 == hi);
        for (int c0 = 0; c0 < H; c0 += chunk) {
            const int cc = c0 + lane * VEC;
            const bool cact = cc < H;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            for (int64_t e0 = s0; e0 < s1; e0 += 64) {
                const int64_t e = e0 + lane;
                int32_t myid = -1;
                float myw = 0.f;
                if (e < s1) { myid = col[e]; myw = val ? val[e] : 1.f; }
                const int kk = (s1 - e0) < 64 ? (int)(s1 - e0) : 64;
                for (int t0 = 0; t0 < kk; t0 += UNROLL) {
                    float rr[UNROLL][VEC], wv[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int t = t0 + u;
                        const int tl = t < 64 ? t : 0;
                        const int32_t id = (t < kk) ? __builtin_amdgcn_readlane(myid, tl) : -1;
                        wv[u] = (t < kk) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myw), tl)) : 0.f;
                        const bool ok = id >= 0 && (int64_t)id < N && cact;       // a source id outside x is skipped, never read
                        if (VEC == 4) {
                            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (ok) q = *reinterpret_cast<const float4 *>(x + (int64_t)id * H + cc);
                            rr[u][0] = q.x; rr[u][1 % VEC] = q.y; rr[u][2 % VEC] = q.z; rr[u][3 % VEC] = q.w;
                        } else {
                            rr[u][0] = ok ? x[(int64_t)id * H + cc] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(rr[u][v], wv[u], acc[v]);
                }
            }
            if (cact) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if (whole) out[r * H + cc + v] = acc[v];
                    else atomicAdd(&out[r * H + cc + v], acc[v]);
                }
            }
        }
        e_begin = s1;
        ++r;
    }
}

}  // namespace

extern "C" int ps_spmm_csr(const int64_t *rowptr, const int32_t *col, const float *val, const float *x, int64_t N, int H,
                           int64_t V, int64_t E, float *out, ps_stream_t stream) {
    if (V < 0 || H <= 0 || N < 0 || E < 0) return PS_EINVAL;
    if (V == 0) return PS_OK;
    if (!out) return PS_EINVAL;
    hipStream_t st = ps_stream(stream);
    if (hipMemsetAsync(out, 0, (size_t)V * H * sizeof(float), st) != hipSuccess) return PS_ELAUNCH;
    if (E == 0) return PS_OK;
    if (!rowptr || !col || !x) return PS_EINVAL;
    const int64_t waves = ps_cdiv(E, SLICE);
    const int64_t grid = ps_cdiv(waves, 4);
    if (grid > 0x7fffffff) return PS_EUNSUPPORTED;
    const bool vec4 = (H % 4 == 0) && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(out)) % 16 == 0);
    if (vec4) hipLaunchKernelGGL(spmm_csr_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val, x, N, H, V, E, out);
    else hipLaunchKernelGGL(spmm_csr_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, val, x, N, H, V, E, out);
    PS_CHECK_LAUNCH();
    return PS_OK;
}
