// csr_build.hip -- adjacency ingest for the random-walk sampler on gfx950.
//
// Replaces RandomWalkSampler._prepare_adjacency_list (reference utils/random_walk.py:33-50):
// the python adj_list (append in edge-column order) is a CSR stably sorted by src.  The stable
// sort of (src, edge#) pairs is rocPRIM's radix sort (one-time ingest, not a hot kernel); the
// row pointer comes from boundary detection on the sorted keys (no atomics -> deterministic).
//
// ps_cdf_build evaluates, per row, exactly what np.random.choice(dest, p=w/w.sum()) computes on
// every step of the reference (utils/random_walk.py:76,79): p = w / numpy_sum(w); cdf =
// cumsum(p); cdf /= cdf[-1] -- fp64, the same operation order, no FMA contraction (this file is
// compiled with -ffp-contract=off), so the result is bit-identical with numpy's.
#include "ps_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

__global__ void prep_keys_kernel(const int64_t *src, int64_t E, uint32_t *keys, uint32_t *vals) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        keys[e] = (uint32_t)src[e];
        vals[e] = (uint32_t)e;
    }
}

// rowptr[v] = first sorted position whose key >= v
__global__ void rowptr_kernel(const uint32_t *keys, int64_t E, int64_t V, int64_t *rowptr) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= E; p += (int64_t)gridDim.x * blockDim.x) {
        // keys >= V (an edge whose source id is out of range: the host wrappers reject such input, this is the
        // last line of defence) are clamped to V, so nothing is ever written past rowptr[V]
        int64_t prev = (p == 0) ? -1 : (int64_t)keys[p - 1];
        int64_t cur = (p == E) ? V : (int64_t)keys[p];
        if (prev > V) prev = V;
        if (cur > V) cur = V;
        for (int64_t v = prev + 1; v <= cur; ++v) rowptr[v] = p;
    }
}

__global__ void gather_kernel(const uint32_t *perm, const int64_t *dst, const float *w, int64_t E, int32_t *col,
                              double *wsorted) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t e = perm[p];
        col[p] = (int32_t)dst[e];
        wsorted[p] = w ? (double)w[e] : 1.0;   // utils/random_walk.py:45-48
    }
}

// ---- numpy ndarray.sum() for a contiguous fp64 vector (see oracle/pinsage_oracle.py) --------
__device__ double pairwise_leaf(const double *a, int64_t n) {   // n <= 128
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

__device__ double pairwise_block(const double *a, int64_t n) {   // n <= 8192: depth <= 7
    int64_t off[10], len[10];
    double left[10];
    int phase[10];
    int sp = 0;
    off[0] = 0; len[0] = n; phase[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        if (phase[sp] == 0) {
            if (len[sp] <= 128) { ret = pairwise_leaf(a + off[sp], len[sp]); --sp; continue; }
            int64_t n2 = len[sp] / 2; n2 -= n2 % 8;
            phase[sp] = 1;
            off[sp + 1] = off[sp]; len[sp + 1] = n2; phase[sp + 1] = 0; ++sp;
        } else if (phase[sp] == 1) {
            left[sp] = ret;
            int64_t n2 = len[sp] / 2; n2 -= n2 % 8;
            phase[sp] = 2;
            off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; phase[sp + 1] = 0; ++sp;
        } else {
            ret = left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

__device__ double numpy_sum(const double *a, int64_t n) {
    double res = 0.0;                          // ufunc buffer: 8192-element chunks, sequential
    for (int64_t i = 0; i < n; i += 8192) {
        const int64_t m = (n - i < 8192) ? (n - i) : 8192;
        res += pairwise_block(a + i, m);
    }
    return res;
}

constexpr int SMALL_ROW = 32;

// rows with degree <= SMALL_ROW: one lane per row
__global__ void cdf_small_kernel(const int64_t *rowptr, const double *w, int64_t V, double *cdf) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = rowptr[v], hi = rowptr[v + 1], n = hi - lo;
        if (n == 0 || n > SMALL_ROW) continue;
        const double S = numpy_sum(w + lo, n);
        double acc = 0.0;
        for (int64_t i = lo; i < hi; ++i) {
            const double p = w[i] / S;
            acc = (i == lo) ? p : acc + p;
            cdf[i] = acc;
        }
        const double last = acc;
        for (int64_t i = lo; i < hi; ++i) cdf[i] = cdf[i] / last;
    }
}

// rows with degree > SMALL_ROW: one wave per row.  The order-dependent parts (numpy sum, cumsum)
// run redundantly on all lanes (uniform control flow, broadcast loads); the divisions are spread
// over the lanes.
__global__ __launch_bounds__(256) void cdf_large_kernel(const int64_t *rowptr, const double *w, int64_t V,
                                                        double *cdf) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = wave; v < V; v += nwaves) {
        const int64_t lo = rowptr[v], hi = rowptr[v + 1], n = hi - lo;
        if (n <= SMALL_ROW) continue;
        const double S = numpy_sum(w + lo, n);
        for (int64_t i = lo + lane; i < hi; i += 64) cdf[i] = w[i] / S;      // p
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // cumsum: lane 0 of the wave walks the row (stores), every lane tracks acc
        double acc = 0.0;
        if (lane == 0) {
            const volatile double *pc = cdf;
            acc = pc[lo];
            for (int64_t i = lo + 1; i < hi; ++i) { acc = acc + pc[i]; cdf[i] = acc; }
        }
        const double last = __shfl(acc, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (int64_t i = lo + lane; i < hi; i += 64) {
            const volatile double *pc = cdf;
            cdf[i] = pc[i] / last;
        }
    }
}

// one wave per row: guide[lo + j] = #{k : cdf[lo + k] <= (j / deg) * (1 - 2^-50)}; nodeinfo[v] = (lo, deg).
// Every u with (uint32)(u * deg) == j satisfies u >= (j / deg) * (1 - 2^-53) > that threshold (even with a
// couple of ulps of error in the division), so guide[lo + j] <= searchsorted(cdf, u, 'right'): a safe start.
__global__ __launch_bounds__(256) void guide_build_kernel(const int64_t *rowptr, const double *cdf, int64_t V,
                                                          uint32_t *nodeinfo, int32_t *guide) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = wave; v < V; v += nwaves) {
        const int64_t lo = rowptr[v], hi = rowptr[v + 1], d = hi - lo;
        if (lane == 0) { nodeinfo[2 * v] = (uint32_t)lo; nodeinfo[2 * v + 1] = (uint32_t)d; }
        for (int64_t j = lane; j < d; j += 64) {
            int64_t l = lo, h = hi;
            if (j == 0) {
                h = lo;
            } else {
                const double t = ((double)j / (double)d) * (1.0 - 0x1p-50);
                while (l < h) {
                    const int64_t mid = l + ((h - l) >> 1);
                    if (cdf[mid] <= t) l = mid + 1; else h = mid;
                }
            }
            guide[lo + j] = (int32_t)(h - lo);
        }
    }
}

__global__ void pack_edges_kernel(const int32_t *col, const double *cdf, const int32_t *guide, int64_t E,
                                  unsigned char *packed) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ((E + 7) / 8) * 8;
         e += (int64_t)gridDim.x * blockDim.x) {
        unsigned char *blk = packed + (e >> 3) * 128;
        const int sl = (int)(e & 7);
        const bool ok = e < E;
        reinterpret_cast<double *>(blk)[sl] = ok ? cdf[e] : 2.0;
        reinterpret_cast<int32_t *>(blk + 64)[sl] = ok ? col[e] : -1;
        reinterpret_cast<int32_t *>(blk + 96)[sl] = ok ? guide[e] : 0;
    }
}

// 64-byte bucket records (see ps_bucket_build in pinsage_hip.h): [c0 c1 | c2 c3 | k0 k1 k2 k3 | c4 k4 -]
__global__ __launch_bounds__(256) void bucket_build_kernel(const int64_t *rowptr, const int32_t *col, const double *cdf,
                                                           const int32_t *guide, int64_t V, unsigned char *buckets) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = wave; v < V; v += nwaves) {
        const int64_t lo = rowptr[v], hi = rowptr[v + 1];
        for (int64_t e = lo + lane; e < hi; e += 64) {
            const int64_t first = lo + guide[e];
            double c[5];
            int32_t k[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int64_t idx = first + i;
                c[i] = idx < hi ? cdf[idx] : 2.0;                 // past the row end: always "> u", destination = last edge
                k[i] = col[idx < hi ? idx : hi - 1];
            }
            unsigned char *r = buckets + (size_t)e * 64;
            reinterpret_cast<double2 *>(r)[0] = make_double2(c[0], c[1]);
            reinterpret_cast<double2 *>(r)[1] = make_double2(c[2], c[3]);
            reinterpret_cast<int4 *>(r)[2] = make_int4(k[0], k[1], k[2], k[3]);
            reinterpret_cast<double *>(r)[6] = c[4];
            reinterpret_cast<int2 *>(r)[7] = make_int2(k[4], 0);
        }
    }
}

// 32-byte half records (see ps_bucket_build_half in pinsage_hip.h): [c0 c1 c2 c3 | k0 k1 k2 k3] = the four CDF entries from the
// bucket's guide position on, each ROUNDED DOWN to fp32, and their destinations.  With lo = (double)c_i and hi = the next float
// above it, cdf_i lies in [lo, hi): `u < lo` proves `u < cdf_i` and `u >= hi` proves `u >= cdf_i`, so the walk takes k_i when
// u >= hi_{i-1} and u < lo_i -- exactly the searchsorted answer -- and repeats the search through the packed blocks in the sliver
// in between (2^-24 of the value: ~10^-5 of a bucket for a row of 100 edges) or when the answer lies beyond the fourth candidate.
__device__ __forceinline__ float round_down_f32(double c) {
    float f = (float)c;                                       // round to nearest, then step down if that went up
    if ((double)f > c) f = __uint_as_float(__float_as_uint(f) - 1u);     // c > 0: positive floats order like their bits
    return f;
}
__global__ __launch_bounds__(256) void bucket_half_build_kernel(const int64_t *rowptr, const int32_t *col, const double *cdf,
                                                                const int32_t *guide, int64_t V, unsigned char *buckets) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = wave; v < V; v += nwaves) {
        const int64_t lo = rowptr[v], hi = rowptr[v + 1];
        for (int64_t e = lo + lane; e < hi; e += 64) {
            const int64_t first = lo + guide[e];
            float c[4];
            int32_t k[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t idx = first + i;
                c[i] = idx < hi ? round_down_f32(cdf[idx]) : 2.0f;  // past the row end: always "> u", destination = last edge
                k[i] = col[idx < hi ? idx : hi - 1];
            }
            unsigned char *r = buckets + (size_t)e * 32;
            reinterpret_cast<float4 *>(r)[0] = make_float4(c[0], c[1], c[2], c[3]);
            reinterpret_cast<int4 *>(r)[1] = make_int4(k[0], k[1], k[2], k[3]);
        }
    }
}

int radix_bits(int64_t V) {
    int bits = 1;
    while (((int64_t)1 << bits) < V && bits < 32) ++bits;
    return bits;
}

}  // namespace

extern "C" size_t ps_csr_build_workspace_bytes(int64_t E, int64_t V) {
    if (E <= 0) return ALIGN;
    size_t temp = 0;
    uint32_t *kn = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, temp, kn, kn, kn, kn, (size_t)E, 0u, (unsigned)radix_bits(V), (hipStream_t)0);
    return 4 * align_up((size_t)E * sizeof(uint32_t)) + align_up(temp) + ALIGN;
}

extern "C" int ps_csr_build(const int64_t *src, const int64_t *dst, const float *w, int64_t E, int64_t V,
                            int64_t *rowptr, int32_t *col, double *wsorted, void *workspace, size_t workspace_bytes,
                            ps_stream_t stream) {
    if (E < 0 || V < 0 || V >= ((int64_t)1 << 31) || E >= ((int64_t)1 << 32) || !rowptr) return PS_EINVAL;
    hipStream_t st = ps_stream(stream);
    if (E == 0) {
        if (hipMemsetAsync(rowptr, 0, (size_t)(V + 1) * sizeof(int64_t), st) != hipSuccess) return PS_ELAUNCH;
        return PS_OK;
    }
    if (!src || !dst || !col || !wsorted || !workspace) return PS_EINVAL;
    const size_t seg = align_up((size_t)E * sizeof(uint32_t));
    if (workspace_bytes < 4 * seg + ALIGN) return PS_EWORKSPACE;
    char *base = reinterpret_cast<char *>(align_up(reinterpret_cast<size_t>(workspace)));
    uint32_t *keys_in = reinterpret_cast<uint32_t *>(base);
    uint32_t *keys_out = reinterpret_cast<uint32_t *>(base + seg);
    uint32_t *vals_in = reinterpret_cast<uint32_t *>(base + 2 * seg);
    uint32_t *vals_out = reinterpret_cast<uint32_t *>(base + 3 * seg);
    void *temp = base + 4 * seg;
    size_t temp_bytes = workspace_bytes - (size_t)((base + 4 * seg) - reinterpret_cast<char *>(workspace));
    size_t need = 0;
    const unsigned bits = (unsigned)radix_bits(V);
    if (rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)E, 0u, bits, st) != hipSuccess)
        return PS_ELAUNCH;
    if (need > temp_bytes) return PS_EWORKSPACE;
    int64_t grid = ps_cdiv(E, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(prep_keys_kernel, dim3((unsigned)grid), dim3(256), 0, st, src, E, keys_in, vals_in);
    PS_CHECK_LAUNCH();
    if (rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, (size_t)E, 0u, bits, st) != hipSuccess)
        return PS_ELAUNCH;
    hipLaunchKernelGGL(rowptr_kernel, dim3((unsigned)grid), dim3(256), 0, st, keys_out, E, V, rowptr);
    PS_CHECK_LAUNCH();
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)grid), dim3(256), 0, st, vals_out, dst, w, E, col, wsorted);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_guide_build(const int64_t *rowptr, const double *cdf, int64_t V, uint32_t *nodeinfo, int32_t *guide,
                              ps_stream_t stream) {
    if (!rowptr || V < 0) return PS_EINVAL;
    if (V == 0) return PS_OK;
    if (!nodeinfo || !cdf || !guide) return PS_EINVAL;
    int64_t g = ps_cdiv(V, 4);
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(guide_build_kernel, dim3((unsigned)g), dim3(256), 0, ps_stream(stream), rowptr, cdf, V, nodeinfo, guide);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

namespace {
__global__ void dest_info_kernel(const int32_t *__restrict__ col, const uint2 *__restrict__ nodeinfo, int64_t E, int64_t V,
                                 uint2 *__restrict__ dest) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t d = col[e];
        dest[e] = (d >= 0 && d < V) ? nodeinfo[d] : make_uint2(0u, 0u);
    }
}
}  // namespace

extern "C" int ps_dest_info_build(const int32_t *col, const uint32_t *nodeinfo, int64_t E, int64_t V, void *dest_info,
                                  ps_stream_t stream) {
    if (E < 0 || V < 0) return PS_EINVAL;
    if (E == 0) return PS_OK;
    if (!col || !nodeinfo || !dest_info || reinterpret_cast<size_t>(dest_info) % 8 != 0) return PS_EINVAL;
    int64_t grid = ps_cdiv(E, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(dest_info_kernel, dim3((unsigned)grid), dim3(256), 0, ps_stream(stream), col,
                       reinterpret_cast<const uint2 *>(nodeinfo), E, V, reinterpret_cast<uint2 *>(dest_info));
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_bucket_build(const int64_t *rowptr, const int32_t *col, const double *cdf, const int32_t *guide,
                               int64_t V, int64_t E, void *buckets, ps_stream_t stream) {
    if (V < 0 || E < 0) return PS_EINVAL;
    if (V == 0 || E == 0) return PS_OK;
    if (!rowptr || !col || !cdf || !guide || !buckets || reinterpret_cast<size_t>(buckets) % 64 != 0) return PS_EINVAL;
    int64_t g = ps_cdiv(V, 4);
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(bucket_build_kernel, dim3((unsigned)g), dim3(256), 0, ps_stream(stream), rowptr, col, cdf, guide, V,
                       reinterpret_cast<unsigned char *>(buckets));
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_bucket_build_half(const int64_t *rowptr, const int32_t *col, const double *cdf, const int32_t *guide,
                                    int64_t V, int64_t E, void *buckets, ps_stream_t stream) {
    if (V < 0 || E < 0) return PS_EINVAL;
    if (V == 0 || E == 0) return PS_OK;
    if (!rowptr || !col || !cdf || !guide || !buckets || reinterpret_cast<size_t>(buckets) % 64 != 0) return PS_EINVAL;
    int64_t g = ps_cdiv(V, 4);
    if (g > 256 * 32) g = 256 * 32;
    hipLaunchKernelGGL(bucket_half_build_kernel, dim3((unsigned)g), dim3(256), 0, ps_stream(stream), rowptr, col, cdf, guide, V,
                       reinterpret_cast<unsigned char *>(buckets));
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_pack_edges(const int32_t *col, const double *cdf, const int32_t *guide, int64_t E, void *packed,
                             ps_stream_t stream) {
    if (E < 0) return PS_EINVAL;
    if (E == 0) return PS_OK;
    if (!col || !cdf || !guide || !packed || reinterpret_cast<size_t>(packed) % 128 != 0) return PS_EINVAL;
    int64_t g = ps_cdiv(E, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(pack_edges_kernel, dim3((unsigned)g), dim3(256), 0, ps_stream(stream), col, cdf, guide, E,
                       reinterpret_cast<unsigned char *>(packed));
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_cdf_build(const int64_t *rowptr, const double *wsorted, int64_t V, double *cdf, ps_stream_t stream) {
    if (!rowptr || V < 0) return PS_EINVAL;
    if (V == 0) return PS_OK;
    if (!wsorted || !cdf) return PS_EINVAL;
    hipStream_t st = ps_stream(stream);
    int64_t g1 = ps_cdiv(V, 256);
    if (g1 > 8192) g1 = 8192;
    hipLaunchKernelGGL(cdf_small_kernel, dim3((unsigned)g1), dim3(256), 0, st, rowptr, wsorted, V, cdf);
    PS_CHECK_LAUNCH();
    int64_t g2 = ps_cdiv(V, 4);
    if (g2 > 256 * 8) g2 = 256 * 8;
    hipLaunchKernelGGL(cdf_large_kernel, dim3((unsigned)g2), dim3(256), 0, st, rowptr, wsorted, V, cdf);
    PS_CHECK_LAUNCH();
    return PS_OK;
}
