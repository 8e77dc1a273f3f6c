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

// Exact selection of the kk smallest 64-bit keys (value key << 32 | id: unique) a wave sees, with ONE threshold for the whole
// wave: a key is admitted iff it is smaller than the kk-th smallest seen so far (and, in a later sweep, larger than the last key
// already emitted).  Admitted keys go to a 256-slot queue in LDS (slots by ballot + prefix count, no atomics); when fewer than
// 64 slots are free the queue is reduced to its kk smallest (kk rounds of a wave minimum over four keys per lane) and the
// threshold tightened.  A row of N elements admits ~kk ln(N / kk) keys in all.  The first version kept a sorted list per LANE:
// with 64 private thresholds some lane was inserting at nearly every element, and the wave sat in a divergent LDS insertion
// loop for the whole row (1.06 ms per 4 546 x 59 047 slab).
constexpr int SELQ = 256;
__device__ __forceinline__ uint64_t wave_min_key(uint64_t v) {
#define PS_STEP(ctrl, rows)                                                                                     \
    {                                                                                                           \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v, ctrl, rows, 0xf, false); \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v >> 32), ctrl, rows, 0xf, false); \
        const uint64_t o = ((uint64_t)hi << 32) | lo;                                                           \
        v = o < v ? o : v;                                                                                      \
    }
    PS_STEP(0xB1, 0xf) PS_STEP(0x4E, 0xf) PS_STEP(0x141, 0xf) PS_STEP(0x140, 0xf) PS_STEP(0x142, 0xa) PS_STEP(0x143, 0xc)
#undef PS_STEP
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}
struct WaveSelect {
    uint64_t *q;          // LDS, SELQ keys of this wave
    int lane, kk, cnt;    // cnt: wave-uniform
    uint64_t thr, after;  // wave-uniform: admit iff after < key < thr (after = 0 / first sweep: no lower bound)
    bool first;
    __device__ __forceinline__ void reset(int kk_) { kk = kk_; cnt = 0; thr = EMPTY_KEY; }
    // queue -> its kk smallest in q[0 .. cnt), ascending; thr = the kk-th when there are kk
    __device__ __forceinline__ void reduce() {
        ps_wave_lds_sync();
        uint64_t r[SELQ / 64];
#pragma unroll
        for (int u = 0; u < SELQ / 64; ++u) r[u] = (u * 64 + lane) < cnt ? q[u * 64 + lane] : EMPTY_KEY;
        ps_wave_lds_sync();
        int found = 0;
        uint64_t last = EMPTY_KEY;
        for (int t = 0; t < kk; ++t) {
            uint64_t m = r[0];
#pragma unroll
            for (int u = 1; u < SELQ / 64; ++u) m = r[u] < m ? r[u] : m;
            const uint64_t w = wave_min_key(m);
            if (w == EMPTY_KEY) break;
#pragma unroll
            for (int u = 0; u < SELQ / 64; ++u) if (r[u] == w) r[u] = EMPTY_KEY;      // keys are unique
            if (lane == 0) q[t] = w;
            last = w;
            ++found;
        }
        cnt = found;
        if (found == kk) thr = last;
        ps_wave_lds_sync();
    }
    __device__ __forceinline__ void offer(bool valid, uint64_t key) {         // converged call: every lane, valid or not
        const bool adm = valid && key < thr && (first || key > after);
        const uint64_t m = __ballot(adm);
        if (m != 0ull) {
            if (adm) q[cnt + __popcll(m & ((1ull << lane) - 1ull))] = key;
            cnt += __popcll(m);
            if (cnt > SELQ - 64) reduce();
        }
    }
};

// MODE 0: largest similarity first (exact inner-product search).  MODE 1: smallest L2 distance first,
// dist = |q|^2 + |x|^2 - 2 q.x, optionally restricted to the items whose inverted list (assign) is probed.
template <int MODE>
__global__ __launch_bounds__(256) void row_topk_kernel(const float *__restrict__ sims, int64_t N, int64_t rows,
                                                       const int64_t *__restrict__ qidx, int exclude_self, int k, int kcap,
                                                       const float *__restrict__ qn, const float *__restrict__ xn,
                                                       const int32_t *__restrict__ assign,
                                                       const uint32_t *__restrict__ probe, int words,
                                                       float *__restrict__ vals, int64_t *__restrict__ ids) {
    __shared__ uint64_t skeys[4 * SELQ];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    if (row >= rows) return;
    const float *s = sims + row * N;
    const int64_t self = exclude_self ? qidx[row] : -1;
    const float qnr = (MODE == 1) ? qn[row] : 0.f;
    // k > kcap: further sweeps over the row, each admitting only keys AFTER the last one emitted -- keys are unique (id in the
    // low word), so the sweeps partition the order exactly
    WaveSelect sel{skeys + wv * SELQ, lane, 0, 0, EMPTY_KEY, 0, true};
    bool dry = false;
    for (int base = 0; base < k; base += kcap) {
        const int kk = (k - base) < kcap ? (k - base) : kcap;
        sel.reset(kk);
        if (!dry)
            for (int64_t j0 = 0; j0 < N; j0 += 512) {              // eight elements per lane requested together
                float sv[8], xv[8];
                int32_t av[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t j = j0 + lane + 64 * u;
                    const bool ok = j < N;
                    sv[u] = ok ? s[j] : 0.f;
                    xv[u] = (MODE == 1 && ok) ? xn[j] : 0.f;
                    av[u] = (MODE == 1 && assign && ok) ? assign[j] : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t j = j0 + lane + 64 * u;
                    bool ok = j < N;
                    float v = sv[u];
                    if (MODE == 1) {
                        if (assign && ok && !((probe[row * words + (av[u] >> 5)] >> (av[u] & 31)) & 1u)) ok = false;   // list not probed
                        v = (qnr + xv[u]) - 2.f * v;
                    }
                    if (j == self) v = -INFINITY;
                    sel.offer(ok, ((uint64_t)(MODE == 1 ? ~desc_key(v) : desc_key(v)) << 32) | (uint32_t)j);
                }
            }
        sel.reduce();
        for (int r = lane; r < kk; r += 64) {
            const uint64_t best = r < sel.cnt ? sel.q[r] : EMPTY_KEY;
            const uint32_t kb = (uint32_t)(best >> 32);
            vals[row * k + base + r] = best != EMPTY_KEY ? key_value(MODE == 1 ? ~kb : kb) : (MODE == 1 ? 3.4028234663852886e38f : -INFINITY);
            ids[row * k + base + r] = best != EMPTY_KEY ? (int64_t)(uint32_t)best : -1;
        }
        if (sel.cnt < kk) dry = true;                                 // nothing left: the remaining slots are padding
        else sel.after = sel.q[kk - 1];
        sel.first = false;
        ps_wave_lds_sync();
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


// ---- inverted-file scan (WeakANDIndex = faiss.IndexIVFFlat over IndexFlatL2, reference utils/nearest_neighbors.py:88-93, :134) ----
// Items are stored sorted by inverted list (list l = rows [list_ptr[l], list_ptr[l + 1]) of X, item_ids = their original ids);
// a query visits its `nprobe` lists only.  List-major evaluation: the (query, list) pairs are grouped by list, every list's
// queries are gathered into consecutive rows (padded to whole 64-row blocks) and ONE grouped fp32-MFMA product multiplies each
// 64-row block with the rows of ITS list only, into a compact slab [rows of the list][items of the list]; a query's top-k
// then sweeps its nprobe slab rows.  Work = nq * nprobe * (average list length) dot products instead of nq * N.
constexpr int IVF_PAIRS_PER_BLOCK = 512;

// bc[b][l] = number of (query, probe) pairs of block b (IVF_PAIRS_PER_BLOCK consecutive pairs) that visit list l: an LDS
// histogram per block (200 000 global atomics on 100 counters took 128 us)
__global__ __launch_bounds__(256) void ivf_count_kernel(const int32_t *__restrict__ probes, int64_t npairs, int nlist, int32_t *__restrict__ bc) {
    extern __shared__ int32_t hist[];
    for (int l = threadIdx.x; l < nlist; l += 256) hist[l] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * IVF_PAIRS_PER_BLOCK;
    for (int i = threadIdx.x; i < IVF_PAIRS_PER_BLOCK; i += 256) {
        const int64_t p = p0 + i;
        if (p < npairs) {
            const int l = probes[p];
            if (l >= 0 && l < nlist) atomicAdd(&hist[l], 1);
        }
    }
    __syncthreads();
    for (int l = threadIdx.x; l < nlist; l += 256) bc[(size_t)blockIdx.x * nlist + l] = hist[l];
}

// one workgroup: bc[b][l] -> the rank base of block b inside list l (exclusive scan over the blocks, in place); then, lists in
// order, row_start[l] (first gathered row of list l, lists padded to 64 rows), slab_off[l] (first float of its slab) and the
// grouped product's descriptors (W row offset, columns, y offset) for every 64-row block up to `max_tiles`
// A list's (start, length) as every kernel of the inverted-file scan sees it: clamped to the slab geometry the host sized from
// the CALLER's `max_list` and N (ps_ivf_topk is an exported entry: a max_list below the longest list or a non-monotone list_ptr
// must not let the grouped product or the row sweep run past the slab; such input gets a truncated list, never a stray access)
__device__ __forceinline__ void ivf_list(const int64_t *__restrict__ list_ptr, int l, int64_t max_list, int64_t N, int64_t &j0,
                                         int64_t &n) {
    j0 = list_ptr[l];
    n = list_ptr[l + 1] - j0;
    if (j0 < 0) j0 = 0;
    if (j0 > N) j0 = N;
    if (n < 0) n = 0;
    if (n > max_list) n = max_list;
    if (n > N - j0) n = N - j0;
}

__global__ __launch_bounds__(1024) void ivf_layout_kernel(int32_t *__restrict__ bc, int nblocks, const int64_t *__restrict__ list_ptr, int nlist,
                                                          int64_t max_list, int64_t N, int64_t max_tiles, int32_t *__restrict__ cnt, int64_t *__restrict__ row_start,
                                                          int64_t *__restrict__ slab_off, int64_t *__restrict__ grp) {
    __shared__ int64_t tile0;
    __shared__ int64_t off0;
    // exclusive scan of bc[.][l] over the blocks, one wave per list: lane j owns a run of consecutive blocks
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
        const int per = (nblocks + 63) / 64;
        for (int l = wv; l < nlist; l += nwv) {
            const int b0 = lane * per, b1 = (b0 + per) < nblocks ? (b0 + per) : nblocks;
            int32_t sum = 0;
            for (int b = b0; b < b1; ++b) sum += bc[(size_t)b * nlist + l];
            int32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int32_t up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            int32_t run = incl - sum;
            for (int b = b0; b < b1; ++b) {
                const int32_t c = bc[(size_t)b * nlist + l];
                bc[(size_t)b * nlist + l] = run;
                run += c;
            }
            if (lane == 63) cnt[l] = incl;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { tile0 = 0; off0 = 0; }
    __syncthreads();
    for (int l = 0; l < nlist; ++l) {                                        // serial over lists (100 by default), tiles in parallel
        const int64_t t0 = tile0, o0 = off0;
        int64_t j0, n;
        ivf_list(list_ptr, l, max_list, N, j0, n);
        const int64_t tiles = n > 0 ? ((int64_t)cnt[l] + 63) / 64 : 0;      // an empty list needs no product
        for (int64_t t = threadIdx.x; t < tiles; t += blockDim.x) {
            if (t0 + t < max_tiles) {
                grp[(t0 + t) * 3 + 0] = j0;
                grp[(t0 + t) * 3 + 1] = n;
                grp[(t0 + t) * 3 + 2] = o0 + t * 64 * n;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            row_start[l] = t0 * 64;
            slab_off[l] = o0;
            tile0 = t0 + tiles;
            off0 = o0 + tiles * 64 * n;
        }
        __syncthreads();
    }
    const int64_t used = tile0;
    for (int64_t t = used + threadIdx.x; t < max_tiles; t += blockDim.x) { grp[t * 3] = 0; grp[t * 3 + 1] = 0; grp[t * 3 + 2] = 0; }
}

// blocks as in ivf_count_kernel; one thread per (query, probe) pair: the pair's rank inside its list = the block's base + an
// LDS cursor (the rank only decides WHERE the pair's slab row lives, never a result), the query's D floats copied to that
// row of the gathered matrix, the slab row's offset recorded
__global__ __launch_bounds__(256) void ivf_gather_kernel(const float *__restrict__ Q, int D, const int32_t *__restrict__ probes,
                                                         int64_t npairs, int nprobe, int nlist, const int64_t *__restrict__ list_ptr,
                                                         int64_t max_list, int64_t N, const int32_t *__restrict__ bc, const int64_t *__restrict__ row_start,
                                                         const int64_t *__restrict__ slab_off, float *__restrict__ Qg,
                                                         int64_t *__restrict__ seg_off) {
    extern __shared__ int32_t cursor[];                                       // [nlist] cursors, then [IVF_PAIRS_PER_BLOCK] gathered rows
    int64_t *rows = reinterpret_cast<int64_t *>(cursor + ((nlist + 1) & ~1));
    for (int l = threadIdx.x; l < nlist; l += 256) cursor[l] = bc[(size_t)blockIdx.x * nlist + l];
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * IVF_PAIRS_PER_BLOCK;
    for (int i = threadIdx.x; i < IVF_PAIRS_PER_BLOCK; i += 256) {            // one thread per pair: rank, row, slab offset
        const int64_t p = p0 + i;
        int64_t row = -1;
        if (p < npairs) {
            const int l = probes[p];
            int64_t n = 0, j0 = 0;
            if (l >= 0 && l < nlist) ivf_list(list_ptr, l, max_list, N, j0, n);
            if (n > 0) {
                const int rank = atomicAdd(&cursor[l], 1);
                row = row_start[l] + rank;
                seg_off[p] = slab_off[l] + (int64_t)rank * n;
            } else {
                seg_off[p] = -1;                                               // skipped probe / empty list
            }
        }
        rows[i] = row;
    }
    __syncthreads();
    // copy the queries: 16-byte pieces when D allows, all pairs of the block in flight together
    if ((D & 3) == 0) {
        const int vpr = D >> 2;                                                // float4 per row
        for (int64_t e = threadIdx.x; e < (int64_t)IVF_PAIRS_PER_BLOCK * vpr; e += 256) {
            const int i = (int)(e / vpr), v = (int)(e - (int64_t)i * vpr);
            const int64_t row = rows[i];
            if (row >= 0)
                reinterpret_cast<float4 *>(Qg + row * D)[v] = reinterpret_cast<const float4 *>(Q + ((p0 + i) / nprobe) * D)[v];
        }
    } else {
        for (int64_t e = threadIdx.x; e < (int64_t)IVF_PAIRS_PER_BLOCK * D; e += 256) {
            const int i = (int)(e / D), d = (int)(e - (int64_t)i * D);
            const int64_t row = rows[i];
            if (row >= 0) Qg[row * D + d] = Q[((p0 + i) / nprobe) * D + d];
        }
    }
}

// One wave per query: the k smallest (distance, original id) over the slab rows of its probed lists.
// dist = (|q|^2 + |x|^2) - 2 q.x with the same operations and the same k-ordered dot product as the flat / masked scans
// (row_topk_kernel<1>), so all of them agree bit for bit on every (query, item) pair they see.
// The nprobe segments (slab row, list start, length) are fetched by nprobe lanes in ONE round trip and the sweep runs over
// the flattened sequence of 256-element pieces of all segments, the loads of piece g + 1 issued before piece g is processed:
// the first version walked the segments one by one behind three dependent loads each and took 14 us per segment.
__global__ __launch_bounds__(256) void ivf_row_topk_kernel(const float *__restrict__ slab, int64_t rows, const int32_t *__restrict__ probes,
                                                           int nprobe, int nlist, const int64_t *__restrict__ list_ptr,
                                                           int64_t max_list, int64_t N, const int64_t *__restrict__ seg_off, const int64_t *__restrict__ item_ids,
                                                           int k, int kcap, const float *__restrict__ qn, const float *__restrict__ xn,
                                                           float *__restrict__ vals, int64_t *__restrict__ ids) {
    __shared__ uint64_t skeys[4 * SELQ];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    if (row >= rows) return;
    const float qnr = qn[row];
    WaveSelect sel{skeys + wv * SELQ, lane, 0, 0, EMPTY_KEY, 0, true};
    bool dry = false;
    for (int base = 0; base < k; base += kcap) {
        const int kk = (k - base) < kcap ? (k - base) : kcap;
        sel.reset(kk);
        for (int pb = 0; pb < nprobe && !dry; pb += 64) {                    // 64 segments per round (nprobe is 20 by default)
            const int np = (nprobe - pb) < 64 ? (nprobe - pb) : 64;
            // lane pi: segment pi of this round
            int64_t so = -1, j0 = 0;
            int n = 0;
            if (lane < np) {
                so = seg_off[row * nprobe + pb + lane];
                if (so >= 0) {
                    const int l = probes[row * nprobe + pb + lane];
                    int64_t n64;
                    ivf_list(list_ptr, l, max_list, N, j0, n64);
                    n = (int)n64;
                }
            }
            float sv[2][4], xv[2][4];
            int64_t iv[2][4];
            int seg = 0, off = 0;                                            // next piece to load: elements off.. of segment seg
            auto skip_empty = [&]() __attribute__((always_inline)) { while (seg < np && off >= __builtin_amdgcn_readlane(n, seg)) { ++seg; off = 0; } };
            auto load_piece = [&](int buf) __attribute__((always_inline)) {  // wave-uniform (seg, off); advances to the next piece
                const int ns = __builtin_amdgcn_readlane(n, seg);
                const int64_t sos = ((int64_t)__builtin_amdgcn_readlane((int)(so >> 32), seg) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)so, seg);
                const int64_t j0s = ((int64_t)__builtin_amdgcn_readlane((int)(j0 >> 32), seg) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)j0, seg);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = off + lane + 64 * u;
                    const bool ok = jj < ns;
                    sv[buf][u] = ok ? slab[sos + jj] : 0.f;
                    xv[buf][u] = ok ? xn[j0s + jj] : 0.f;
                    iv[buf][u] = ok ? item_ids[j0s + jj] : -1;
                }
                off += 256;
            };
            auto process = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = (qnr + xv[buf][u]) - 2.f * sv[buf][u];
                    sel.offer(iv[buf][u] >= 0, ((uint64_t)(~desc_key(v)) << 32) | (uint32_t)iv[buf][u]);
                }
            };
            skip_empty();
            if (seg < np) {
                load_piece(0);
                while (true) {                                               // buffer indices are literals: the pieces stay in registers
                    skip_empty();
                    const bool more1 = seg < np;
                    if (more1) load_piece(1);
                    process(0);
                    if (!more1) break;
                    skip_empty();
                    const bool more0 = seg < np;
                    if (more0) load_piece(0);
                    process(1);
                    if (!more0) break;
                }
            }
        }
        sel.reduce();
        for (int r = lane; r < kk; r += 64) {
            const uint64_t best = r < sel.cnt ? sel.q[r] : EMPTY_KEY;
            vals[row * k + base + r] = best != EMPTY_KEY ? key_value(~(uint32_t)(best >> 32)) : 3.4028234663852886e38f;
            ids[row * k + base + r] = best != EMPTY_KEY ? (int64_t)(uint32_t)best : -1;
        }
        if (sel.cnt < kk) dry = true;
        else sel.after = sel.q[kk - 1];
        sel.first = false;
        ps_wave_lds_sync();
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
    for (int64_t q0 = 0; q0 < nq; q0 += c) {
        const int64_t rows = (nq - q0) < c ? (nq - q0) : c;
        int64_t g = ps_cdiv(rows * D, 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, E, D, qidx + q0, rows, Q);
        PS_CHECK_LAUNCH();
        const int rc = ps_linear(Q, rows, D, E, D, nullptr, (int)N, nullptr, 0, nullptr, 0, 0, sims, stream);
        if (rc != PS_OK) return rc;
        hipLaunchKernelGGL(row_topk_kernel<0>, dim3((unsigned)ps_cdiv(rows, 4)), dim3(256), 0, st, sims, N, rows,
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
    for (int64_t q0 = 0; q0 < nq; q0 += c) {
        const int64_t rows = (nq - q0) < c ? (nq - q0) : c;
        const int rc = ps_linear(Q + q0 * D, rows, D, X, D, nullptr, (int)N, nullptr, 0, nullptr, 0, 0, sims, stream);
        if (rc != PS_OK) return rc;
        hipLaunchKernelGGL(row_topk_kernel<1>, dim3((unsigned)ps_cdiv(rows, 4)), dim3(256), 0, st, sims, N, rows,
                           (const int64_t *)nullptr, 0, k, kcap, qn + q0, xn, assign, probe ? probe + q0 * words : nullptr, words,
                           dist + q0 * k, ids + q0 * k);
        PS_CHECK_LAUNCH();
    }
    return PS_OK;
}

namespace {
// queries per chunk: the slabs of a chunk hold at most (pairs + 64 nlist) * max_list floats; keep that under 2 GiB
int64_t ivf_chunk(int64_t nq, int nprobe, int nlist, int64_t max_list) {
    const int64_t budget = ((int64_t)1 << 29) / (max_list > 0 ? max_list : 1) - 64 * (int64_t)nlist;      // <= 2 GiB of slabs
    int64_t c = budget / nprobe;
    if (c < 64) c = 64;
    return c < nq ? c : nq;
}
struct IvfLayout {
    size_t slab, qg, xn, qn, grp, seg, rowstart, slaboff, cnt, bc, total;
    int64_t chunk, max_tiles, nblocks;
};
IvfLayout ivf_layout(int64_t nq, int64_t N, int D, int nlist, int nprobe, int64_t max_list) {
    IvfLayout L{};
    L.chunk = ivf_chunk(nq, nprobe, nlist, max_list);
    L.max_tiles = (L.chunk * nprobe + 63) / 64 + nlist;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.slab = take((size_t)L.max_tiles * 64 * (size_t)(max_list > 0 ? max_list : 1) * sizeof(float));
    L.qg = take((size_t)L.max_tiles * 64 * D * sizeof(float));
    L.xn = take((size_t)(N + 64) * sizeof(float));
    L.qn = take((size_t)(nq + 64) * sizeof(float));
    L.grp = take((size_t)L.max_tiles * 3 * sizeof(int64_t));
    L.seg = take((size_t)L.chunk * nprobe * sizeof(int64_t));
    L.rowstart = take((size_t)nlist * sizeof(int64_t));
    L.slaboff = take((size_t)nlist * sizeof(int64_t));
    L.cnt = take((size_t)nlist * sizeof(int32_t));
    L.nblocks = (L.chunk * nprobe + IVF_PAIRS_PER_BLOCK - 1) / IVF_PAIRS_PER_BLOCK;
    L.bc = take((size_t)L.nblocks * nlist * sizeof(int32_t));
    L.total = off + 512;
    return L;
}
}  // namespace

extern "C" size_t ps_ivf_topk_workspace_bytes(int64_t nq, int64_t N, int D, int k, int nlist, int nprobe, int64_t max_list) {
    (void)k;
    if (nq <= 0 || N < 0 || D <= 0 || nlist <= 0 || nprobe <= 0 || max_list < 0) return 256;
    return ivf_layout(nq, N, D, nlist, nprobe, max_list).total;
}

extern "C" int ps_ivf_topk(const float *X, int64_t N, int D, const int64_t *list_ptr, int nlist, int64_t max_list,
                           const int64_t *item_ids, const float *Q, int64_t nq, const int32_t *probes, int nprobe, int k,
                           float *dist, int64_t *ids, void *workspace, size_t workspace_bytes, ps_stream_t stream) {
    if (N < 0 || D <= 0 || nq < 0 || k <= 0 || nlist <= 0 || nprobe <= 0 || max_list < 0 || max_list > N) return PS_EINVAL;
    if (N >= ((int64_t)1 << 32)) return PS_EUNSUPPORTED;
    if (nq == 0) return PS_OK;
    if (!Q || !probes || !dist || !ids || !workspace || !list_ptr) return PS_EINVAL;
    if (N > 0 && (!X || !item_ids)) return PS_EINVAL;
    const IvfLayout L = ivf_layout(nq, N, D, nlist, nprobe, max_list);
    if (workspace_bytes < L.total) return PS_EWORKSPACE;
    hipStream_t st = ps_stream(stream);
    char *base = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    float *slab = reinterpret_cast<float *>(base + L.slab), *Qg = reinterpret_cast<float *>(base + L.qg);
    float *xn = reinterpret_cast<float *>(base + L.xn), *qn = reinterpret_cast<float *>(base + L.qn);
    int64_t *grp = reinterpret_cast<int64_t *>(base + L.grp), *seg = reinterpret_cast<int64_t *>(base + L.seg);
    int64_t *row_start = reinterpret_cast<int64_t *>(base + L.rowstart), *slab_off = reinterpret_cast<int64_t *>(base + L.slaboff);
    int32_t *cnt = reinterpret_cast<int32_t *>(base + L.cnt), *bc = reinterpret_cast<int32_t *>(base + L.bc);
    const size_t hist_lds = (size_t)nlist * sizeof(int32_t);
    if (hist_lds > 48 * 1024) return PS_EUNSUPPORTED;          // > 12 288 lists
    const int kcap = k < TOPK_SWEEP ? k : TOPK_SWEEP;
    int64_t g;
    if (N > 0) {
        g = ps_cdiv(N, 4);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)g), dim3(256), 0, st, X, N, D, xn);
        PS_CHECK_LAUNCH();
    }
    g = ps_cdiv(nq, 4);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)g), dim3(256), 0, st, Q, nq, D, qn);
    PS_CHECK_LAUNCH();
    for (int64_t q0 = 0; q0 < nq; q0 += L.chunk) {
        const int64_t rows = (nq - q0) < L.chunk ? (nq - q0) : L.chunk;
        const int64_t npairs = rows * nprobe;
        const int32_t *pr = probes + q0 * nprobe;
        const int nb = (int)ps_cdiv(npairs, IVF_PAIRS_PER_BLOCK);
        hipLaunchKernelGGL(ivf_count_kernel, dim3((unsigned)nb), dim3(256), hist_lds, st, pr, npairs, nlist, bc);
        PS_CHECK_LAUNCH();
        hipLaunchKernelGGL(ivf_layout_kernel, dim3(1), dim3(1024), 0, st, bc, nb, list_ptr, nlist, max_list, N, L.max_tiles, cnt, row_start, slab_off, grp);
        PS_CHECK_LAUNCH();
        hipLaunchKernelGGL(ivf_gather_kernel, dim3((unsigned)nb), dim3(256), ((hist_lds + 7) / 8 * 8) + IVF_PAIRS_PER_BLOCK * sizeof(int64_t), st, Q + q0 * D, D, pr, npairs, nprobe, nlist, list_ptr,
                           max_list, N, bc, row_start, slab_off, Qg, seg);
        PS_CHECK_LAUNCH();
        if (N > 0 && max_list > 0) {
            const int rc = psi_linear_grouped(Qg, L.max_tiles * 64, D, X, D, slab, grp, (int)max_list, stream);
            if (rc != PS_OK) return rc;
        }
        hipLaunchKernelGGL(ivf_row_topk_kernel, dim3((unsigned)ps_cdiv(rows, 4)), dim3(256), 0, st, slab, rows, pr, nprobe, nlist,
                           list_ptr, max_list, N, seg, item_ids, k, kcap, qn + q0, xn, dist + q0 * k, ids + q0 * k);
        PS_CHECK_LAUNCH();
    }
    return PS_OK;
}
