// hamming_topk.hip -- brute-force Hamming k-NN over bit codes on gfx950.
//
// Replaces faiss.IndexLSH.search as called by LSHIndex.search (reference
// utils/nearest_neighbors.py:47-68): every query is compared with ALL ntotal codes
// (popcount of XOR) and the k smallest by (distance, id) are returned in ascending order.
//
// Mapping: one lane = one query (its code lives in registers), a wave = 64 queries sweeping a
// contiguous slice of the code table.  The table word is wave-uniform, so it is fetched through
// the scalar cache and each 32-bit word costs two VALU ops per 64 queries (v_xor + v_bcnt
// accumulate).  Each lane keeps its k best (distance<<32 | local id) keys sorted in an LDS column;
// ids ascend during the sweep, so "key < current worst" is exactly faiss' strict-less admission.
// Slices of the table are swept by different waves and merged by ps_topk_merge's kernel.
#include "ps_common.h"

namespace {

constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;

template <int WORDS>
__global__ __launch_bounds__(256) void hamming_scan_kernel(const uint32_t *__restrict__ q, int64_t nq,
                                                           const uint32_t *__restrict__ codes, int64_t N, int k,
                                                           int splits, int64_t id_offset, int32_t *__restrict__ odist,
                                                           int64_t *__restrict__ oids) {
    extern __shared__ uint64_t skeys[];   // [4 waves][k][64 lanes]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t *col = skeys + (size_t)wv * k * 64 + lane;   // this lane's column: col[p * 64]
    const int64_t gw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    const int64_t ngroups = (nq + 63) / 64;
    if (gw >= ngroups * splits) return;
    const int64_t qg = gw / splits;
    const int split = (int)(gw % splits);
    const int64_t per = (N + splits - 1) / splits;
    const int64_t j0 = split * per;
    const int64_t j1 = (j0 + per < N) ? (j0 + per) : N;
    const int64_t qi = qg * 64 + lane;
    const bool qact = qi < nq;

    uint32_t qc[WORDS];
#pragma unroll
    for (int w = 0; w < WORDS; ++w) qc[w] = qact ? q[qi * WORDS + w] : 0u;
    for (int p = 0; p < k; ++p) col[p * 64] = EMPTY_KEY;
    uint64_t worst = qact ? EMPTY_KEY : 0ull;
    int filled = 0;

    for (int64_t j = j0; j < j1; ++j) {
        const uint32_t *c = codes + j * WORDS;   // wave-uniform address -> scalar loads
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) d += __builtin_popcount(qc[w] ^ c[w]);
        const uint64_t key = ((uint64_t)d << 32) | (uint32_t)(j - j0);
        if (key < worst) {                       // rare once the list is warm
            int p = filled < k ? filled : k - 1;
            while (p > 0 && col[(p - 1) * 64] > key) { col[p * 64] = col[(p - 1) * 64]; --p; }
            col[p * 64] = key;
            if (filled < k) ++filled;
            if (filled == k) worst = col[(k - 1) * 64];
        }
    }
    if (qact) {
        for (int p = 0; p < k; ++p) {
            const uint64_t key = col[p * 64];
            const int64_t o = ((int64_t)split * nq + qi) * k + p;
            if (p < filled) {
                odist[o] = (int32_t)(key >> 32);
                oids[o] = (int64_t)(uint32_t)key + j0 + id_offset;
            } else {
                odist[o] = 0x7fffffff;
                oids[o] = -1;
            }
        }
    }
}

// one wave per query: k rounds of (lane-local min over strided candidates) + wave min-reduce.
__global__ __launch_bounds__(256) void topk_merge_kernel(const int32_t *__restrict__ din, const int64_t *__restrict__ iin,
                                                         int P, int64_t nq, int k, int32_t *__restrict__ dout,
                                                         int64_t *__restrict__ iout) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int n = P * k;
    for (int64_t qi = wave; qi < nq; qi += nw) {
        int32_t ld = -1;          // last selected (dist, id): strictly increasing selection
        int64_t li = -1;
        for (int r = 0; r < k; ++r) {
            int32_t bd = 0x7fffffff;
            int64_t bi = 0x7fffffffffffffffll;
            for (int c = lane; c < n; c += 64) {
                const int p = c / k, t = c - p * k;
                const int64_t o = ((int64_t)p * nq + qi) * k + t;
                const int32_t d = din[o];
                const int64_t id = iin[o];
                if (id < 0) continue;
                const bool after = (d > ld) || (d == ld && id > li);
                const bool better = (d < bd) || (d == bd && id < bi);
                if (after && better) { bd = d; bi = id; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int32_t od = __shfl_xor(bd, o, 64);
                const int64_t oi = __shfl_xor(bi, o, 64);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            const bool found = bi != 0x7fffffffffffffffll;
            if (lane == 0) {
                dout[qi * k + r] = found ? bd : 0x7fffffff;
                iout[qi * k + r] = found ? bi : -1;
            }
            if (found) { ld = bd; li = bi; } else { ld = 0x7fffffff; li = 0x7fffffffffffffffll; }
        }
    }
}

int pick_splits(int64_t nq, int64_t N) {
    const int64_t groups = (nq + 63) / 64;
    int64_t s = (256 * 8 + groups - 1) / groups;   // aim at >= 2048 waves
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    while (s > 1 && N / s < 64) s >>= 1;           // keep slices worth sweeping
    return (int)s;
}

}  // namespace

extern "C" size_t ps_hamming_topk_workspace_bytes(int64_t nq, int64_t N, int cs, int k) {
    if (nq <= 0 || k <= 0) return 256;
    const int s = pick_splits(nq, N);
    return (size_t)s * (size_t)nq * (size_t)k * (sizeof(int32_t) + sizeof(int64_t)) + 512;
}

extern "C" int ps_topk_merge(const int32_t *dist_in, const int64_t *ids_in, int P, int64_t nq, int k, int32_t *dist,
                             int64_t *ids, ps_stream_t stream) {
    if (P <= 0 || nq < 0 || k <= 0) return PS_EINVAL;
    if (nq == 0) return PS_OK;
    if (!dist_in || !ids_in || !dist || !ids) return PS_EINVAL;
    int64_t grid = ps_cdiv(nq, 4);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)grid), dim3(256), 0, ps_stream(stream), dist_in, ids_in, P, nq,
                       k, dist, ids);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_hamming_topk(const uint8_t *qcodes, int64_t nq, const uint8_t *codes, int64_t N, int cs, int k,
                               int64_t id_offset, int32_t *dist, int64_t *ids, void *workspace, size_t workspace_bytes,
                               ps_stream_t stream) {
    if (nq < 0 || N < 0 || cs <= 0 || k <= 0) return PS_EINVAL;
    if (cs % 4 != 0) return PS_EUNSUPPORTED;
    if (k > 160) return PS_EUNSUPPORTED;            // 4 waves * k * 64 lanes * 8 B of LDS
    if (nq == 0) return PS_OK;
    if (!qcodes || !dist || !ids || (N > 0 && !codes)) return PS_EINVAL;
    if ((reinterpret_cast<size_t>(qcodes) | reinterpret_cast<size_t>(codes)) % 4 != 0) return PS_EINVAL;
    if (N >= ((int64_t)1 << 32)) return PS_EUNSUPPORTED;
    const int words = cs / 4;
    const int s = pick_splits(nq, N);
    hipStream_t st = ps_stream(stream);
    int32_t *cd = dist;
    int64_t *ci = ids;
    if (s > 1) {
        const size_t need = (size_t)s * nq * k * (sizeof(int32_t) + sizeof(int64_t)) + 512;
        if (!workspace || workspace_bytes < need) return PS_EWORKSPACE;
        char *base = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
        ci = reinterpret_cast<int64_t *>(base);
        cd = reinterpret_cast<int32_t *>(base + (size_t)s * nq * k * sizeof(int64_t));
    }
    const int64_t waves = ((nq + 63) / 64) * s;
    const unsigned grid = (unsigned)ps_cdiv(waves, 4);
    const size_t lds = (size_t)4 * k * 64 * sizeof(uint64_t);
    const uint32_t *q32 = reinterpret_cast<const uint32_t *>(qcodes);
    const uint32_t *c32 = reinterpret_cast<const uint32_t *>(codes);
#define PS_LAUNCH_SCAN(WORDS_)                                                                                     \
    hipLaunchKernelGGL(hamming_scan_kernel<WORDS_>, dim3(grid), dim3(256), lds, st, q32, nq, c32, N, k, s, id_offset, \
                       cd, ci)
    switch (words) {
        case 1: PS_LAUNCH_SCAN(1); break;
        case 2: PS_LAUNCH_SCAN(2); break;
        case 4: PS_LAUNCH_SCAN(4); break;
        case 8: PS_LAUNCH_SCAN(8); break;
        case 16: PS_LAUNCH_SCAN(16); break;
        case 32: PS_LAUNCH_SCAN(32); break;
        default: return PS_EUNSUPPORTED;
    }
#undef PS_LAUNCH_SCAN
    PS_CHECK_LAUNCH();
    if (s > 1) return ps_topk_merge(cd, ci, s, nq, k, dist, ids, stream);
    return PS_OK;
}
