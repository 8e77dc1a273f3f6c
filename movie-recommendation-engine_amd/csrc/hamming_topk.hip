// hamming_topk.hip -- brute-force Hamming k-NN over bit codes on gfx950.
//
// Replaces faiss.IndexLSH.search as called by LSHIndex.search (reference
// utils/nearest_neighbors.py:47-68): every query is compared with ALL ntotal codes
// (popcount of XOR) and the k smallest by (distance, id) are returned in ascending order.
//
// Slices of the table are swept by different waves (see hamming_scan_kernel) and merged by
// ps_topk_merge's kernel.
#include "ps_common.h"

namespace {


// One wave = up to QT (<= 64) queries x one slice of the code table.  Lanes own db items: a chunk of G
// 64-item groups is loaded once (coalesced, G * WORDS registers per lane) and compared with every query of
// the tile; the query code is wave-uniform, so it comes in through scalar loads (the next query's code is
// requested before the current one is consumed) and costs no vector registers or LDS.  Per 32-bit word
// and 64 (query, item) pairs: v_xor + v_bcnt accumulate.
// Top-k: a candidate is ONE 32-bit key, distance << SHIFT | slice-local id (the host cuts the table so that a
// slice has at most 2^SHIFT items), so "better" is a single unsigned compare.  Lane qi of `tau` holds query
// qi's admission bound (its current k-th best key); a ballot of "distance < bound's distance" is almost always
// empty.  Otherwise the wave-uniform rare path pulls that query's sorted list out of LDS (lane p = p-th best),
// inserts the candidates in lane (= id) order -- position by ballot + popcount, shift by one lane -- and
// stores it back.  No divergence, no atomics; ids ascend during the sweep, so "key < current worst" is exactly
// faiss' strict-less admission.
// lane p receives lane p-1's value (lane 0 keeps its own): DPP wave_shr:1, no LDS crossbar round trip
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}

constexpr uint32_t EMPTY_KEY = 0xffffffffu;

template <int WORDS, int G>
__global__ __launch_bounds__(256) void hamming_scan_kernel(const uint32_t *__restrict__ q, int64_t nq,
                                                           const uint32_t *__restrict__ codes, int64_t N, int k,
                                                           int kcap, int QT, int splits, int shift, int64_t id_offset,
                                                           int32_t *__restrict__ odist, int64_t *__restrict__ oids) {
    extern __shared__ uint32_t lists_all[];                           // [4 waves][QT][kcap]
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    uint32_t *L = lists_all + (size_t)wv * QT * kcap;
    const int64_t gw = (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    const int64_t ntiles = (nq + QT - 1) / QT;
    if (gw >= ntiles * splits) return;
    const int64_t tile = gw / splits;
    const int split = (int)(gw - tile * splits);
    const int64_t per = (N + splits - 1) / splits;
    const int64_t j0 = split * per;
    const int64_t j1 = (j0 + per < N) ? (j0 + per) : N;
    const int64_t q0 = tile * QT;
    const int nqt = (int)((nq - q0) < QT ? (nq - q0) : QT);
    const bool vec_ok = (WORDS % 4 == 0) && (reinterpret_cast<size_t>(codes) % 16 == 0);

    for (int i = lane; i < QT * kcap; i += 64) L[i] = EMPTY_KEY;
    ps_wave_lds_sync();
    uint32_t tau = EMPTY_KEY;                                          // lane qi: admission bound of query qi

    for (int64_t g = j0; g < j1; g += 64 * G) {
        uint32_t it[G][WORDS];
        uint32_t klo[G];                                              // slice-local id, all ones when out of range
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            const int64_t j = g + gg * 64 + lane;
            const bool valid = j < j1;
            klo[gg] = valid ? (uint32_t)(j - j0) : EMPTY_KEY;
            if (vec_ok) {
#pragma unroll
                for (int w = 0; w < WORDS; w += 4) {
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (valid) v = *reinterpret_cast<const uint4 *>(codes + j * WORDS + w);
                    it[gg][w] = v.x; it[gg][w + 1 < WORDS ? w + 1 : w] = v.y;
                    it[gg][w + 2 < WORDS ? w + 2 : w] = v.z; it[gg][w + 3 < WORDS ? w + 3 : w] = v.w;
                }
            } else {
#pragma unroll
                for (int w = 0; w < WORDS; ++w) it[gg][w] = valid ? codes[j * WORDS + w] : 0u;
            }
        }
        uint32_t cur[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; ++w) cur[w] = q[q0 * WORDS + w];   // wave-uniform -> scalar load
        for (int qi = 0; qi < nqt; ++qi) {                            // scalar loop
            const int qn = (qi + 1 < nqt) ? qi + 1 : qi;
            uint32_t nxt[WORDS];
#pragma unroll
            for (int w = 0; w < WORDS; ++w) nxt[w] = q[(q0 + qn) * WORDS + w];   // in flight during the compares
            uint32_t tk = __builtin_amdgcn_readlane(tau, qi);
            const uint32_t thi = tk >> shift;
            uint32_t d[G];
            uint64_t cand[G];
            uint64_t many = 0ull;
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                uint32_t acc = 0;
#pragma unroll
                for (int w = 0; w < WORDS; ++w) acc += __builtin_popcount(it[gg][w] ^ cur[w]);
                d[gg] = acc;
                // ids ascend during the sweep: an item that ties the list's worst distance has a larger id and
                // can never be admitted, so "distance < bound.distance" is the exact fast-path test
                cand[gg] = __ballot(acc < thi);
                many |= cand[gg];
            }
            if (many != 0ull) {                                        // rare, wave-uniform
                uint32_t lk = (lane < kcap) ? L[qi * kcap + lane] : EMPTY_KEY;
#pragma unroll
                for (int gg = 0; gg < G; ++gg) {
                    if (cand[gg] == 0ull) continue;                    // nothing here beat even the bound at entry
                    // candidates of this group against the CURRENT bound (it tightens with every insertion,
                    // so stale candidates drop out of the ballot instead of being visited one by one)
                    const uint32_t key = (d[gg] << shift) | klo[gg];      // out-of-range lanes: all ones = EMPTY_KEY
                    uint64_t mask = __ballot(key < tk);
                    while (mask != 0ull) {
                        const int b = __builtin_ctzll(mask);
                        const uint32_t c = __builtin_amdgcn_readlane(key, b);
                        const int pos = __popcll(__ballot(lk < c));     // the list is sorted: a prefix
                        const uint32_t sh = wave_shr1(lk);
                        lk = (lane > pos) ? sh : lk;
                        lk = (lane == pos) ? c : lk;
                        tk = __builtin_amdgcn_readlane(lk, k - 1);
                        const uint64_t above = (b == 63) ? 0ull : (~0ull << (b + 1));
                        mask = __ballot(key < tk) & above;
                    }
                }
                if (lane < kcap) L[qi * kcap + lane] = lk;
                if (lane == qi) tau = tk;
            }
#pragma unroll
            for (int w = 0; w < WORDS; ++w) cur[w] = nxt[w];
        }
    }
    ps_wave_lds_sync();
    const uint32_t idmask = (1u << shift) - 1u;
    for (int i = lane; i < nqt * k; i += 64) {
        const int qi = i / k, p = i - qi * k;
        const uint32_t key = L[qi * kcap + p];
        const int64_t o = ((int64_t)split * nq + q0 + qi) * k + p;
        const bool has = key != EMPTY_KEY;
        odist[o] = has ? (int32_t)(key >> shift) : 0x7fffffff;
        oids[o] = has ? (int64_t)(key & idmask) + j0 + id_offset : -1;
    }
}

// Merge by selection with 16 lanes per query (four queries per wave): the P * k <= 16 KPL candidates of a query sit in
// registers (candidate c in lane c % 16, register c / 16); round r picks the smallest (distance, id) after the previous
// winner: a lane-local scan and a row minimum in four DPP steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every
// lane ends with the minimum of its 16-lane row; no LDS).  203 VALU instructions per query at P = 8, k = 11; r01's rank merge
// (every candidate counts the candidates before it, broadcast one by one through v_readlane) spent 1 760: 53 us for 10 000 queries.  Lists need not be sorted; ids are distinct across lists.
struct MergeKey { uint32_t d, hi, lo; };     // (distance, id) as unsigned words; none = all ones
__device__ __forceinline__ bool key_less(const MergeKey &a, const MergeKey &b) {
    return a.d < b.d || (a.d == b.d && (a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo)));
}
__device__ __forceinline__ MergeKey row_min_key(MergeKey v) {
#define PS_STEP(ctrl)                                                                          \
    {                                                                                          \
        MergeKey o;                                                                            \
        o.d = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.d, ctrl, 0xf, 0xf, true);        \
        o.hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.hi, ctrl, 0xf, 0xf, true);      \
        o.lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.lo, ctrl, 0xf, 0xf, true);      \
        if (key_less(o, v)) v = o;                                                             \
    }
    PS_STEP(0xB1) PS_STEP(0x4E) PS_STEP(0x141) PS_STEP(0x140)
#undef PS_STEP
    return v;
}
template <int KPL>
__global__ __launch_bounds__(256) void topk_select16_kernel(const int32_t *__restrict__ din, const int64_t *__restrict__ iin,
                                                            int64_t dstride, int64_t istride, int P, int64_t nq, int k,
                                                            int32_t *__restrict__ dout, int64_t *__restrict__ iout) {
    const int sub = threadIdx.x & 15;
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int n = P * k;
    constexpr MergeKey NONE = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    MergeKey c[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int ci = r * 16 + sub;
        c[r] = NONE;
        if (q < nq && ci < n) {
            const int p = ci / k, t = ci - p * k;
            const int64_t o = q * k + t;
            const int64_t id = iin[(int64_t)p * istride + o];
            if (id >= 0) c[r] = MergeKey{(uint32_t)din[(int64_t)p * dstride + o], (uint32_t)((uint64_t)id >> 32), (uint32_t)id};
        }
    }
    MergeKey last = NONE;
    bool first = true;
    for (int r = 0; r < k; ++r) {
        MergeKey best = NONE;
#pragma unroll
        for (int j = 0; j < KPL; ++j)
            if ((first || key_less(last, c[j])) && key_less(c[j], best)) best = c[j];
        const MergeKey m = row_min_key(best);
        const bool found = !(m.d == NONE.d && m.hi == NONE.hi && m.lo == NONE.lo);
        if (sub == 0 && q < nq) {
            dout[q * k + r] = found ? (int32_t)m.d : 0x7fffffff;
            iout[q * k + r] = found ? (int64_t)(((uint64_t)m.hi << 32) | m.lo) : -1;
        }
        last = m;              // none once the candidates are exhausted: nothing is "after" it
        first = false;
    }
}

// one wave per query: k rounds of (lane-local min over strided candidates) + wave min-reduce.
__global__ __launch_bounds__(256) void topk_merge_kernel(const int32_t *__restrict__ din, const int64_t *__restrict__ iin,
                                                         int64_t dstride, int64_t istride, int P, int64_t nq, int k,
                                                         int32_t *__restrict__ dout, int64_t *__restrict__ iout) {
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
                const int64_t o = qi * k + t;
                const int32_t d = din[(int64_t)p * dstride + o];
                const int64_t id = iin[(int64_t)p * istride + o];
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

// Queries per wave.  Small tiles need fewer table slices for the same number of waves (every slice pays its
// own top-k warm-up): measured on MI355X with an L2-resident table (59 047 x 512 bit) 4 beats 8 / 16 / 32 for
// 10 K and for 59 K queries alike.  A table far beyond the caches (>= 256 MB) is re-read from HBM once per tile,
// so there the tile is 32 queries.
int pick_tile(int64_t nq, int64_t table_bytes) {
    (void)nq;
    return table_bytes >= ((int64_t)256 << 20) ? 32 : 4;
}

// key = distance << shift | slice-local id: distances of cs-byte codes need log2(8 cs) + 1 bits
int key_shift(int cs) {
    int dbits = 1;
    while ((1 << (dbits - 1)) < cs * 8) ++dbits;
    return 32 - dbits;
}

int pick_splits(int64_t nq, int64_t N, int cs) {
    const int qt = pick_tile(nq, N * cs);
    const int64_t tiles = (nq + qt - 1) / qt;
    int64_t s = (4096 + tiles - 1) / tiles;          // aim at >= 4096 waves (measured best of 2048 / 4096 / 8192)
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    while (s > 1 && N / s < 1024) s >>= 1;           // keep slices worth sweeping
    const int64_t cap = (int64_t)1 << key_shift(cs);  // a slice's local ids must fit under the distance bits
    const int64_t smin = (N + cap - 1) / cap;
    if (s < smin) s = smin;
    return (int)s;
}

}  // namespace

extern "C" size_t ps_hamming_topk_workspace_bytes(int64_t nq, int64_t N, int cs, int k) {
    if (nq <= 0 || k <= 0) return 256;
    const int s = pick_splits(nq, N, cs);
    return (size_t)s * (size_t)nq * (size_t)k * (sizeof(int32_t) + sizeof(int64_t)) + 512;
}

extern "C" int ps_topk_merge_strided(const int32_t *dist_in, int64_t dist_stride, const int64_t *ids_in, int64_t ids_stride,
                                     int P, int64_t nq, int k, int32_t *dist, int64_t *ids, ps_stream_t stream) {
    if (P <= 0 || nq < 0 || k <= 0) return PS_EINVAL;
    if (nq == 0) return PS_OK;
    if (!dist_in || !ids_in || !dist || !ids) return PS_EINVAL;
    if (dist_stride < nq * k || ids_stride < nq * k) return PS_EINVAL;     // shard lists must not overlap
    const int64_t n = (int64_t)P * k;
    hipStream_t st = ps_stream(stream);
    if (n <= 256) {
        const unsigned g16 = (unsigned)ps_cdiv(nq, 16);                          // 16 lanes per query, 256 threads per block
#define PS_SEL16(KPL_) hipLaunchKernelGGL(topk_select16_kernel<KPL_>, dim3(g16), dim3(256), 0, st, dist_in, ids_in, dist_stride, \
                                          ids_stride, P, nq, k, dist, ids)
        if (n <= 32) PS_SEL16(2);
        else if (n <= 64) PS_SEL16(4);
        else if (n <= 128) PS_SEL16(8);
        else PS_SEL16(16);
#undef PS_SEL16
    } else {
        int64_t grid = ps_cdiv(nq, 4);
        if (grid > 256 * 16) grid = 256 * 16;
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)grid), dim3(256), 0, st, dist_in, ids_in, dist_stride, ids_stride, P,
                           nq, k, dist, ids);
    }
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_topk_merge(const int32_t *dist_in, const int64_t *ids_in, int P, int64_t nq, int k, int32_t *dist,
                             int64_t *ids, ps_stream_t stream) {
    if (nq < 0 || k <= 0) return PS_EINVAL;
    return ps_topk_merge_strided(dist_in, nq * k, ids_in, nq * k, P, nq, k, dist, ids, stream);
}

extern "C" int ps_hamming_topk(const uint8_t *qcodes, int64_t nq, const uint8_t *codes, int64_t N, int cs, int k,
                               int64_t id_offset, int32_t *dist, int64_t *ids, void *workspace, size_t workspace_bytes,
                               ps_stream_t stream) {
    if (nq < 0 || N < 0 || cs <= 0 || k <= 0) return PS_EINVAL;
    if (cs % 4 != 0) return PS_EUNSUPPORTED;
    if (k > 64) return PS_EUNSUPPORTED;             // the k best keys of a query live across the 64 lanes
    if (nq == 0) return PS_OK;
    if (!qcodes || !dist || !ids || (N > 0 && !codes)) return PS_EINVAL;
    if ((reinterpret_cast<size_t>(qcodes) | reinterpret_cast<size_t>(codes)) % 4 != 0) return PS_EINVAL;
    if (N >= ((int64_t)1 << 32)) return PS_EUNSUPPORTED;
    const int words = cs / 4;
    const int s = pick_splits(nq, N, cs);
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
    const int QUERY_TILE = pick_tile(nq, N * cs);
    const int64_t waves = ((nq + QUERY_TILE - 1) / QUERY_TILE) * s;
    const unsigned grid = (unsigned)ps_cdiv(waves, 4);
    int kcap = 16;
    while (kcap < k) kcap <<= 1;
    const size_t lds = (size_t)4 * QUERY_TILE * kcap * sizeof(uint32_t);
    const int shift = key_shift(cs);
    const uint32_t *q32 = reinterpret_cast<const uint32_t *>(qcodes);
    const uint32_t *c32 = reinterpret_cast<const uint32_t *>(codes);
#define PS_LAUNCH_SCAN(WORDS_)                                                                                  \
    hipLaunchKernelGGL((hamming_scan_kernel<WORDS_, (WORDS_ >= 32 ? 2 : 4)>), dim3(grid), dim3(256), lds, st, q32, nq, \
                       c32, N, k, kcap, QUERY_TILE, s, shift, id_offset, cd, ci)
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
