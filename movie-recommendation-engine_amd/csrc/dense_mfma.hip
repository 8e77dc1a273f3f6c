// dense_mfma.hip -- fp32 MFMA GEMM for gfx950 with fused epilogues.
//
//   ps_linear     : y = epi( x W^T (+ x2 W2^T) + b ), epi = bias / ReLU / row L2-normalise.
//                   Replaces the nn.Linear + F.relu + torch.cat + F.normalize chain of
//                   PinSage.forward (reference model/pinsage.py:202, 235-240, 248-249).
//   ps_lsh_encode : codes = bitpack( x A^T >= 0 ), the random-hyperplane projection of
//                   faiss.IndexLSH (reference utils/nearest_neighbors.py:26,43,66) with a wavefront
//                   ballot bit-pack epilogue (LSB-first bytes, faiss fvec2bitvec).
//
// Arithmetic: v_mfma_f32_32x32x2_f32, i.e. an exact fp32 fma chain with k ascending and one
// accumulator per output (no split-K) -- bit-identical to `acc = fmaf(x[k], w[k], acc)`, which is
// what the CPU oracle evaluates, so LSH sign decisions are bit-exact.
//
// Tiling: WM x WN waves; block tile BM x BN = (WM*TM*32) x (WN*TN*32): 64 x 256 when the epilogue normalises whole
// rows, 64 x 128 otherwise, 32 x 256 for small shards (gemm_shard_kernel below when the launch is at most two workgroups per
// CU); K step BK = 32 (BK = 16 was measured 6 % slower: the 64 x 256
// tile is register limited to 2 waves per SIMD, not LDS limited).
// LDS image per operand row: [BK/8 groups of 8 k][lane-half h][4] so that the lane (row i, half h)
// fetches its four k values of one group (k = 8g + 2t + h, t = 0..3) with a single ds_read_b128;
// row stride BK+4 floats keeps the b128 reads bank-conflict free.  The k-permutation is done
// in registers while staging (two float4 global loads per row-group), global loads for the next
// K step are in flight while the current one is multiplied -- issued one after every four MFMAs, not in a burst --
// and inside a K step the LDS fragments of k-group g+1 are requested before the 16 MFMAs of group g (order
// pinned with sched_group_barrier).  Two blocks per CU (launch bound): the older wave of a SIMD owns the MFMA pipe and
// the younger runs in its gaps; MFMA and VALU instructions do not overlap on a SIMD, which is what prices the epilogue
// (tools/gemm_trace.py).
#include "ps_common.h"
#include <type_traits>

#ifndef PS_GEMM_DEBUG
#define PS_GEMM_DEBUG 0    // knock-out experiments (tools/gemm_knockout.sh): 1 no global fetch after the first K step, 2 no LDS
#endif                     // staging after it, 4 no barriers in the K loop, 8 no epilogue math, 16 fragments read once, 32 no stores

#ifndef PS_GEMM_SPREAD
#define PS_GEMM_SPREAD 1
#endif

#if PS_GEMM_DEBUG & 64      // timeline experiment (tools/gemm_trace.py): per wave HW_ID, XCC_ID and s_memtime stamps
__device__ unsigned long long ps_gemm_trace_buf[4096 * 4 * 48];
#define PS_TRACE(slot) do { if (lane == 0 && blockIdx.x < 4096 && (slot) < 48) \
    ps_gemm_trace_buf[((size_t)blockIdx.x * 4 + (wv & 3)) * 48 + (slot)] = __builtin_readcyclecounter(); } while (0)
extern "C" int ps_debug_gemm_trace(unsigned long long *host, int clear) {
    if (clear) { static unsigned long long z[4096 * 4 * 48]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(ps_gemm_trace_buf), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ps_gemm_trace_buf), sizeof(unsigned long long) * 4096 * 4 * 48);
}
#else
#define PS_TRACE(slot) do {} while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


struct GemmArgs {
    const float *x;  int64_t M; int K;  const float *W;  int ldw;
    const float *x2; int K2;            const float *W2; int ldw2;
    const float *bias; int N; int flags;
    float *y;            // EPI 0
    uint8_t *codes; int cs;  // EPI 1
    const int64_t *grp;      // optional (grouped products, psi_linear_grouped): per 64-row block (W row offset, columns, y offset)
};

// 8 consecutive k of one row (zero-filled outside [0,K) / invalid row)
__device__ __forceinline__ void load8(const float *base, bool row_ok, int k, int K, bool vec_ok, float (&v)[8]) {
    if (row_ok && vec_ok && k + 8 <= K) {
        const float4 a = *reinterpret_cast<const float4 *>(base + k);
        const float4 b = *reinterpret_cast<const float4 *>(base + k + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (row_ok && k + j < K) ? base[k + j] : 0.f;
    }
}

// Sum over the 32 lanes of each wave half, valid in lanes 16..31 / 48..63, with DPP only (gemm_dma_kernel)
#define PS_DPP_ACC(v, ctrl, rows) \
    (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rows), 0xf, true))
__device__ __forceinline__ float half_sum32(float v) {
    PS_DPP_ACC(v, 0xB1, 0xf);    // quad_perm [1,0,3,2]: lane ^ 1
    PS_DPP_ACC(v, 0x4E, 0xf);    // quad_perm [2,3,0,1]: lane ^ 2
    PS_DPP_ACC(v, 0x141, 0xf);   // row_half_mirror: the other quad of the 8
    PS_DPP_ACC(v, 0x140, 0xf);   // row_mirror: the other 8 of the 16
    PS_DPP_ACC(v, 0x142, 0xa);   // row_bcast:15 -> rows 1 and 3 add the row before them
    return v;
}
#undef PS_DPP_ACC

// Row sums of the fused L2 norm.  Every lane holds NV partial sums (one per row of its wave tile: v[idx], idx = 16 a + r)
// over ITS column; wanted: the sums over the 32 columns of the lane's wave half.  Reducing every value over 32 lanes takes
// 5 NV DPP adds; a transposing butterfly halves the number of live values at every level instead (lane L keeps the values
// whose idx bit equals a predicate of L and adds its partner's copy of the same values): 3 (NV/2 + NV/4 + ...) operations,
// and every lane ends up with the complete sum of ONE idx (bits = its predicates).  Partners: quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror (gfx9 DPP has no xor 4 / xor 8), v_permlane16_swap for the two rows of the half; a mirror complements the lower
// lane bits, so the predicates are b0^b2, b1^b2, b2^b3, b3, b4 -- each invariant under every LATER pairing, which keeps a
// lane and its partner on the same set of rows.
#define PS_DPP_ADD(v, ctrl) \
    ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true)))
template <int NV>
__device__ __forceinline__ float transpose_sum32(float (&v)[NV], int lane) {
    static_assert(NV == 16 || NV == 32, "one or two 32-row tiles per wave");
    const bool p1 = ((lane ^ (lane >> 2)) & 1) != 0, p2 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0;
    const bool p3 = (((lane >> 2) ^ (lane >> 3)) & 1) != 0, p4 = ((lane >> 3) & 1) != 0;
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) {      // idx bit 0 <- p1
        const float lo = PS_DPP_ADD(v[2 * j], 0xB1), hi = PS_DPP_ADD(v[2 * j + 1], 0xB1);      // quad_perm [1,0,3,2]
        v[j] = p1 ? hi : lo;
    }
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {      // idx bit 1 <- p2
        const float lo = PS_DPP_ADD(v[2 * j], 0x4E), hi = PS_DPP_ADD(v[2 * j + 1], 0x4E);      // quad_perm [2,3,0,1]
        v[j] = p2 ? hi : lo;
    }
#pragma unroll
    for (int j = 0; j < NV / 8; ++j) {      // idx bit 2 <- p3
        const float lo = PS_DPP_ADD(v[2 * j], 0x141), hi = PS_DPP_ADD(v[2 * j + 1], 0x141);    // row_half_mirror
        v[j] = p3 ? hi : lo;
    }
#pragma unroll
    for (int j = 0; j < NV / 16; ++j) {     // idx bit 3 <- p4
        const float lo = PS_DPP_ADD(v[2 * j], 0x140), hi = PS_DPP_ADD(v[2 * j + 1], 0x140);    // row_mirror
        v[j] = p4 ? hi : lo;
    }
    // the two 16-lane rows of the half: v_permlane16_swap(x, y) exchanges x's odd rows with y's even rows
    const float x = v[0], y = NV == 32 ? v[1] : v[0];
    const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    const unsigned s0 = sw[0], s1 = sw[1];   // ([x.row0, y.row0, ..], [x.row1, y.row1, ..]): the sum is x's total in even rows, y's in odd
    return __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);            // idx bit 4 <- b4 (NV == 32)
}
#undef PS_DPP_ADD

// One (row, 8-k group) item, registers lo = k 0..3, hi = k 4..7, into its LDS image [lane half h][4] = k 2 t + h.
// (Writing the permutation with ds_write2_b32 from two unrelated registers needs no moves at all but measured equal: its
// 8-byte writes conflict 4-way.)
template <bool PIN>
__device__ __forceinline__ void stash_item(float *d, f32x4 lo, f32x4 hi) {
    if (PIN) asm volatile("" : "+v"(lo), "+v"(hi));  // pins the moves behind the barrier (else: vmcnt waits among the MFMAs)
    *reinterpret_cast<f32x4 *>(d) = f32x4{lo[0], lo[2], hi[0], hi[2]};
    *reinterpret_cast<f32x4 *>(d + 4) = f32x4{lo[1], lo[3], hi[1], hi[3]};
}
// The same item of an operand that is STORED in image order (PS_WPERM: every 8-k group of a weight row as k 0 2 4 6 1 3 5 7,
// ps_permute_k): no register moves.  The 64 x 256 tile's K step had 40 v_mov for 64 MFMAs, 32 of them for the weights, and a
// vector instruction costs a SIMD whose matrix pipe is saturated ~13 cycles of MFMA issue (PMC: 9.2 M VALU / 3.15 M MFMA per
// launch, pipe 63 % busy): weights are static, so the permutation is done once per parameter version instead of per K step.
template <bool PIN>
__device__ __forceinline__ void stash_item_stored(float *d, f32x4 lo, f32x4 hi) {
    if (PIN) asm volatile("" : "+v"(lo), "+v"(hi));
    *reinterpret_cast<f32x4 *>(d) = lo;
    *reinterpret_cast<f32x4 *>(d + 4) = hi;
}

// FAST: every operand is 16-B aligned with K % BK == 0 -> unconditional float4 loads (rows past the end are
// clamped to the last valid row; their results are never stored), so nothing branches or waits inside the
// fetch and the loads stay in flight under the MFMAs.  The general variant predicates every element.
// The epilogue of a block tile (bias / ReLU / fused row L2 norm / stores, or the LSH sign + ballot bit-pack), shared by the one-tile
// kernel and the persistent kernel below.  C layout (32x32 tile): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
template <int WM, int WN, int TM, int TN, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, f32x16 (&acc)[TM][TN], int64_t m0, int n0, float *sRed, float *sNrm) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN, li = lane & 31, lh = lane >> 5;
    (void)BN; (void)wv;
    PS_TRACE(40);
    // ------------------------------ epilogue -------------------------------------------------
    // C layout (32x32 tile): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (EPI == 1) {
        uint32_t *codes32 = reinterpret_cast<uint32_t *>(g.codes);
        const int words = g.cs >> 2;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int colbase = n0 + (wn * TN + b) * 32;
                const bool col_ok = colbase + li < g.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t mask = __ballot(col_ok && acc[a][b][r] >= 0.f);   // bit = (xt >= 0)
                    const int64_t row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (li == 0 && row < g.M && colbase < g.N)
                        codes32[row * words + (colbase >> 5)] = lh ? (uint32_t)(mask >> 32) : (uint32_t)mask;
                }
            }
        PS_TRACE(41);
        return;
    }

    const bool relu = (g.flags & PS_RELU) && !(PS_GEMM_DEBUG & 8), l2 = (g.flags & PS_L2NORM) && !(PS_GEMM_DEBUG & 8);
    if (PS_GEMM_DEBUG & 32) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[a][b][r];
        if (s == 12345.678f) g.y[0] = s;
        PS_TRACE(41);
        return;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + (wn * TN + b) * 32 + li;
        const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][b][r] + bv;
                if (relu) v = v > 0.f ? v : 0.f;
                if (col >= g.N) v = 0.f;
                acc[a][b][r] = v;
            }
    }
    PS_TRACE(42);
    if (l2) {   // F.normalize(p=2, dim=1, eps=1e-12): x / max(||x||, eps); the block holds whole rows
        float ss[TM * 16];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = 0.f;
#pragma unroll
                for (int b = 0; b < TN; ++b) t = fmaf(acc[a][b][r], acc[a][b][r], t);
                ss[a * 16 + r] = t;
            }
        const float tot = transpose_sum32<TM * 16>(ss, lane);               // this wave's columns of ONE row per lane:
        {                                                                   // idx bits = the butterfly's predicates
            const int idx = ((lane ^ (lane >> 2)) & 1) | (((lane >> 1) ^ (lane >> 2)) & 1) << 1 | (((lane >> 2) ^ (lane >> 3)) & 1) << 2 |
                            (lane & 8) | (TM == 2 ? lane & 16 : 0);
            const int a = idx >> 4, r = idx & 15;
            const int rowl = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (TM == 2 || li < 16) sRed[rowl * WN + wn] = tot;
        }
        PS_TRACE(43);
        __syncthreads();
        // one thread per row: norm, its correctly rounded reciprocal, the fast path's guard (see below)
        for (int row = tid; row < BM; row += NT) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) t += sRed[row * WN + w];
            float nrm = sqrtf(t);
            nrm = nrm > 1e-12f ? nrm : 1e-12f;
            const bool tame = nrm >= 0x1p-40f && nrm <= 0x1p40f;
            *reinterpret_cast<f32x4 *>(sNrm + row * 4) = f32x4{nrm, 1.0f / nrm, nrm * 0x1p-60f, tame ? 0.f : 1.f};
        }
        __syncthreads();
        PS_TRACE(44);
        // x / nrm, IEEE-exact, without 64 v_div sequences per lane (11 instructions each; the epilogue was 22 % of a block's
        // life, tools/gemm_trace.py): with y = RN(1 / nrm), q0 = x y is within 1.5 ulp, q1 = q0 + (x - nrm q0) y is a faithful
        // quotient and q2 = q1 + (x - nrm q1) y is the correctly rounded one (Markstein's theorem; the remainders are exact
        // in an fma) -- as long as nothing underflows on the way: the row's norm in [2^-40, 2^40] and x = +0 or
        // |x| >= 2^-60 nrm.  A wave that sees anything else divides the ordinary way (tools/ubench/div_check.hip counts
        // mismatches of the fast path against a / b: none in 6.9e10 pairs, half of them next to rounding boundaries).
        bool wild = false;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 nr = *reinterpret_cast<const f32x4 *>(sNrm + ((wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 4);
                wild |= nr[3] != 0.f;
#pragma unroll
                for (int b = 0; b < TN; ++b)                      // -0 counts as nonzero: the corrections would return +0
                    wild |= __float_as_uint(acc[a][b][r]) != 0u && !(fabsf(acc[a][b][r]) >= nr[2]);
            }
        if (__builtin_amdgcn_ballot_w64(wild) == 0) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 nr = *reinterpret_cast<const f32x4 *>(sNrm + ((wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 4);
                    const float nrm = nr[0], y = nr[1];
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const float x = acc[a][b][r];
                        const float q0 = x * y, q1 = fmaf(fmaf(-nrm, q0, x), y, q0);
                        acc[a][b][r] = fmaf(fmaf(-nrm, q1, x), y, q1);
                    }
                }
        } else {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float nrm = sNrm[((wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 4];
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b][r] = acc[a][b][r] / nrm;
                }
        }
    }
    PS_TRACE(45);
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);     // block-uniform: no per-element guards
    if (interior) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *dst = g.y + (m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * g.N + n0 + wn * TN * 32 + li;
#pragma unroll
                for (int b = 0; b < TN; ++b) dst[b * 32] = acc[a][b][r];
            }
        PS_TRACE(41);
        return;
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= g.M) continue;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + (wn * TN + b) * 32 + li;
                if (col < g.N) g.y[row * g.N + col] = acc[a][b][r];
            }
        }
    PS_TRACE(41);
}

template <int WM, int WN, int TM, int TN, int BK, int EPI, bool FAST>
#ifndef PS_GEMM_OCC
#define PS_GEMM_OCC 2      // two blocks per CU (<= 256 VGPR + AGPR): one block's barriers and epilogue under the other's MFMAs
#endif
__global__ __launch_bounds__(WM * WN * 64, PS_GEMM_OCC) void gemm_f32_kernel(GemmArgs g_in) {
    GemmArgs g = g_in;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    // global loads of the next K step spread over the MFMAs (below): pays for the 2 x 2-tile waves (10 loads per 64 MFMAs);
    // the 1 x 2-tile blocks measured 3 % slower with it (LSH projection 0.151 -> 0.155 ms)
    constexpr bool SPREAD = FAST && PS_GEMM_SPREAD && TM * TN >= 4;
    constexpr int GRP = BK / 8;                        // 8-k groups per row
    constexpr int LDS_STRIDE = BK + 4;                 // floats; 144 B (BK 32) / 80 B (BK 16): b128 reads conflict free
    constexpr int A_ITEMS = (BM * GRP + NT - 1) / NT, B_ITEMS = (BN * GRP + NT - 1) / NT;   // (row, group) items per thread
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_STRIDE + BM * WN + BM * 4];
    float *sA = smem, *sB = smem + BM * LDS_STRIDE, *sRed = smem + (BM + BN) * LDS_STRIDE, *sNrm = sRed + BM * WN;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int li = lane & 31, lh = lane >> 5;
    // grouped products (the inverted-file scan, csrc/dot_topk.hip): row block blockIdx.x multiplies its 64 rows of x with ITS
    // OWN block of W rows -- the items of one inverted list -- and writes a compact [rows, columns] slab
    if (g.grp != nullptr) {
        const int64_t *d = g.grp + (size_t)blockIdx.x * 3;
        const int64_t w_row0 = d[0], ncols = d[1], y_off = d[2];
        if (n0 >= ncols) return;                                       // block-uniform
        g.W += w_row0 * g.ldw;
        g.N = (int)ncols;
        g.y += y_off - m0 * ncols;                                     // y[(m0 + r) * N + c] = slab[r * ncols + c]
    }

#if PS_GEMM_DEBUG & 64
    if (lane == 0 && blockIdx.x < 4096) {
        ps_gemm_trace_buf[((size_t)blockIdx.x * 4 + (wv & 3)) * 48 + 46] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);    // HW_ID
        ps_gemm_trace_buf[((size_t)blockIdx.x * 4 + (wv & 3)) * 48 + 47] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);   // XCC_ID
    }
    int trace_step = 0;
#endif
    PS_TRACE(0);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    for (int phase = 0; phase < 2; ++phase) {
        const float *X = phase == 0 ? g.x : g.x2;
        const float *Wp = phase == 0 ? g.W : g.W2;
        const int K = phase == 0 ? g.K : g.K2;
        const int ldw = phase == 0 ? g.ldw : g.ldw2;
        if (X == nullptr || K <= 0) continue;
        const bool vecA = (K % 4 == 0) && (reinterpret_cast<size_t>(X) % 16 == 0);
        const bool vecB = (ldw % 4 == 0) && (reinterpret_cast<size_t>(Wp) % 16 == 0);

        const bool wperm = (g.flags & PS_WPERM) != 0;                          // block-uniform
        // staging registers: the two 16-byte halves of every (row, 8-k group) item, as loaded
        f32x4 ra[A_ITEMS][2], rb[B_ITEMS][2];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int q = 0; q < A_ITEMS; ++q) {
                // FAST: no guards -- when the tile has fewer items than threads the surplus threads duplicate an
                // item (identical loads, identical LDS writes)
                const int it = FAST ? (tid + NT * q) % (BM * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                const int64_t m = m0 + row;
                if (FAST) {
                    const float *src = X + (m < g.M ? m : g.M - 1) * K + k0 + grp * 8;
                    ra[q][0] = *reinterpret_cast<const f32x4 *>(src);
                    ra[q][1] = *reinterpret_cast<const f32x4 *>(src + 4);
                } else if (it < BM * GRP) {
                    float v[8];
                    load8(X + m * K, m < g.M, k0 + grp * 8, K, vecA, v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ra[q][0][j] = v[j]; ra[q][1][j] = v[4 + j]; }
                }
            }
#pragma unroll
            for (int q = 0; q < B_ITEMS; ++q) {
                const int it = FAST ? (tid + NT * q) % (BN * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                const int n = n0 + row;
                if (FAST) {
                    const float *src = Wp + (int64_t)(n < g.N ? n : g.N - 1) * ldw + k0 + grp * 8;
                    rb[q][0] = *reinterpret_cast<const f32x4 *>(src);
                    rb[q][1] = *reinterpret_cast<const f32x4 *>(src + 4);
                } else if (it < BN * GRP) {
                    float v[8];
                    load8(Wp + (int64_t)n * ldw, n < g.N, k0 + grp * 8, K, vecB, v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { rb[q][0][j] = v[j]; rb[q][1][j] = v[4 + j]; }
                }
            }
        };
        // LDS image of an item: [lane half h][4] = k 2 t + h.  The empty asm pins the permuting register moves HERE, after
        // the barrier: left to itself the compiler performs them right behind the loads, i.e. waits for global memory
        // inside the MFMA stream of the previous step
        auto stash = [&]() {
#pragma unroll
            for (int q = 0; q < A_ITEMS; ++q) {
                const int it = FAST ? (tid + NT * q) % (BM * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                if (!FAST && it >= BM * GRP) continue;
                float *d = sA + row * LDS_STRIDE + grp * 8;
                stash_item<SPREAD>(d, ra[q][0], ra[q][1]);
            }
#pragma unroll
            for (int q = 0; q < B_ITEMS; ++q) {
                const int it = FAST ? (tid + NT * q) % (BN * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                if (!FAST && it >= BN * GRP) continue;
                float *d = sB + row * LDS_STRIDE + grp * 8;
                if (FAST && wperm) stash_item_stored<SPREAD>(d, rb[q][0], rb[q][1]);
                else stash_item<SPREAD>(d, rb[q][0], rb[q][1]);
            }
        };

        fetch(0);
        for (int k0 = 0; k0 < K; k0 += BK) {
#if PS_GEMM_DEBUG & 64
            PS_TRACE(2 + 2 * trace_step);                                   // end of the step's MFMA stream (issue)
#endif
            if (!(PS_GEMM_DEBUG & 4) || k0 == 0) __syncthreads();           // previous step's fragment reads are done
            if (!(PS_GEMM_DEBUG & 2) || k0 == 0) stash();
            if (!(PS_GEMM_DEBUG & 4) || k0 == 0) __syncthreads();
#if PS_GEMM_DEBUG & 64
            PS_TRACE(3 + 2 * trace_step);                                   // image published: MFMA stream starts
            ++trace_step;
#endif
            // FAST: unconditional (the last step re-reads its own slab: harmless) so that the loads, the fragment reads and
            // the MFMAs of a step are ONE basic block and the issue order below applies to all of them
            if (SPREAD && !(PS_GEMM_DEBUG & 1)) fetch(k0 + BK < K ? k0 + BK : k0);
            else if (k0 + BK < K && !(PS_GEMM_DEBUG & 1)) fetch(k0 + BK);
            // fragments of group g+1 are requested BEFORE group g's MFMAs: the two waves of a SIMD run in lockstep
            // (same barriers, round-robin issue), so an LDS round trip taken between groups idles the MFMA pipe for
            // both of them -- measured 65 % pipe use without this prefetch
            f32x4 fa[2][TM], fb[2][TN];
            auto frags = [&](int grp, int s) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    fa[s][a] = *reinterpret_cast<const f32x4 *>(sA + ((wm * TM + a) * 32 + li) * LDS_STRIDE + grp * 8 + lh * 4);
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    fb[s][b] = *reinterpret_cast<const f32x4 *>(sB + ((wn * TN + b) * 32 + li) * LDS_STRIDE + grp * 8 + lh * 4);
            };
            if (!(PS_GEMM_DEBUG & 16) || k0 == 0) frags(0, 0);
#pragma unroll
            for (int grp = 0; grp < GRP; ++grp) {
                const int s = (PS_GEMM_DEBUG & 16) ? 0 : grp & 1;
                if (grp + 1 < GRP && !(PS_GEMM_DEBUG & 16)) frags(grp + 1, s ^ 1);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][a][t], fb[s][b][t], acc[a][b], 0, 0, 0);
            }
            // issue order of the step (one basic block): the fragments of group g+1 are requested BEFORE group g's MFMAs
            // (the compiler otherwise sinks the reads behind them), and the global loads of the next K step are spread
            // over the MFMAs, one after every TM*TN of them: issued in one burst after the barrier they kept all eight
            // waves of the CU in the address path while the MFMA pipe idled (knock-out: 22 of 176 us)
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
            for (int grp = 0; grp < GRP; ++grp) {
                if (grp + 1 < GRP) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                    if (SPREAD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    }

    gemm_epilogue<WM, WN, TM, TN, EPI>(g, acc, m0, n0, sRed, sNrm);
}

// ------------------------------------------------------------------------------------------------------------
// gemm_f32_pkernel: the same block tile, LDS image, MFMA stream and epilogue as gemm_f32_kernel<..., FAST = true>, as a
// PERSISTENT kernel: a workgroup walks the tiles b, b + gridDim.x, ... and treats a tile boundary (and the boundary between
// the two operand pairs of a layer GEMM) like a K-step boundary -- the first slab of the NEXT tile is requested during the last
// MFMA step of the current one, so the global-load latency in front of a tile's first MFMA, which the one-tile kernel pays per
// block, is paid once per workgroup.  Aligned operands only (16-byte rows, K % 32 == 0).  Arithmetic unchanged: one
// accumulator per output, k ascending.
template <int WM, int WN, int TM, int TN, int BK, int EPI>
__global__ __launch_bounds__(WM * WN * 64, PS_GEMM_OCC) void gemm_f32_pkernel(GemmArgs g, int tiles_n, int ntiles) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr bool SPREAD = PS_GEMM_SPREAD && TM * TN >= 4;
    constexpr int GRP = BK / 8;
    constexpr int LDS_STRIDE = BK + 4;
    constexpr int A_ITEMS = (BM * GRP + NT - 1) / NT, B_ITEMS = (BN * GRP + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_STRIDE + BM * WN + BM * 4];
    float *sA = smem, *sB = smem + BM * LDS_STRIDE, *sRed = smem + (BM + BN) * LDS_STRIDE, *sNrm = sRed + BM * WN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int steps0 = g.K / BK, steps1 = g.x2 != nullptr ? g.K2 / BK : 0, nsteps = steps0 + steps1;
    const bool wperm = (g.flags & PS_WPERM) != 0;                              // block-uniform

    f32x4 ra[A_ITEMS][2], rb[B_ITEMS][2];
    // slab s of the tile at (m0, n0): K step s of the first operand pair, or s - steps0 of the second
    auto fetch = [&](int64_t m0, int n0, int s) __attribute__((always_inline)) {
        const bool second = s >= steps0;
        const float *X = second ? g.x2 : g.x;
        const float *Wp = second ? g.W2 : g.W;
        const int K = second ? g.K2 : g.K, ldw = second ? g.ldw2 : g.ldw;
        const int k0 = (second ? s - steps0 : s) * BK;
#pragma unroll
        for (int q = 0; q < A_ITEMS; ++q) {
            const int it = (tid + NT * q) % (BM * GRP), row = it / GRP, grp = it % GRP;
            const int64_t m = m0 + row;
            const float *src = X + (m < g.M ? m : g.M - 1) * K + k0 + grp * 8;
            ra[q][0] = *reinterpret_cast<const f32x4 *>(src);
            ra[q][1] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
#pragma unroll
        for (int q = 0; q < B_ITEMS; ++q) {
            const int it = (tid + NT * q) % (BN * GRP), row = it / GRP, grp = it % GRP;
            const int n = n0 + row;
            const float *src = Wp + (int64_t)(n < g.N ? n : g.N - 1) * ldw + k0 + grp * 8;
            rb[q][0] = *reinterpret_cast<const f32x4 *>(src);
            rb[q][1] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < A_ITEMS; ++q) {
            const int it = (tid + NT * q) % (BM * GRP), row = it / GRP, grp = it % GRP;
            stash_item<SPREAD>(sA + row * LDS_STRIDE + grp * 8, ra[q][0], ra[q][1]);
        }
#pragma unroll
        for (int q = 0; q < B_ITEMS; ++q) {
            const int it = (tid + NT * q) % (BN * GRP), row = it / GRP, grp = it % GRP;
            if (wperm) stash_item_stored<SPREAD>(sB + row * LDS_STRIDE + grp * 8, rb[q][0], rb[q][1]);
            else stash_item<SPREAD>(sB + row * LDS_STRIDE + grp * 8, rb[q][0], rb[q][1]);
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int64_t m0 = (int64_t)(tile / tiles_n) * BM;
    int n0 = (tile % tiles_n) * BN;
    fetch(m0, n0, 0);
    while (true) {
        const int next = tile + (int)gridDim.x;
        const bool more = next < ntiles;                     // block-uniform
        const int64_t m0n = more ? (int64_t)(next / tiles_n) * BM : m0;
        const int n0n = more ? (next % tiles_n) * BN : n0;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int s = 0; s < nsteps; ++s) {
            __syncthreads();                                  // previous step's fragment reads are done
            stash();
            __syncthreads();
            // the next slab: of this tile, or the first one of the workgroup's next tile, or (nothing left) this one again --
            // always a fetch, so that the loads, the fragment reads and the MFMAs of a step are ONE basic block
            {
                const bool last = s + 1 >= nsteps;
                fetch(last ? m0n : m0, last ? n0n : n0, last ? (more ? 0 : s) : s + 1);
            }
            f32x4 fa[2][TM], fb[2][TN];
            auto frags = [&](int grp, int sl) __attribute__((always_inline)) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    fa[sl][a] = *reinterpret_cast<const f32x4 *>(sA + ((wm * TM + a) * 32 + li) * LDS_STRIDE + grp * 8 + lh * 4);
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    fb[sl][b] = *reinterpret_cast<const f32x4 *>(sB + ((wn * TN + b) * 32 + li) * LDS_STRIDE + grp * 8 + lh * 4);
            };
            frags(0, 0);
#pragma unroll
            for (int grp = 0; grp < GRP; ++grp) {
                const int sl = grp & 1;
                if (grp + 1 < GRP) frags(grp + 1, sl ^ 1);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl][a][t], fb[sl][b][t], acc[a][b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
            for (int grp = 0; grp < GRP; ++grp) {
                if (grp + 1 < GRP) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                    if (SPREAD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
        gemm_epilogue<WM, WN, TM, TN, EPI>(g, acc, m0, n0, sRed, sNrm);
        if (!more) break;
        tile = next;
        m0 = m0n;
        n0 = n0n;
    }
}

// ------------------------------------------------------------------------------------------------------------
// gemm_dma_kernel: the same arithmetic (v_mfma_f32_32x32x2_f32, k ascending, one accumulator per output: bit-identical
// to the fmaf chain) with both operands brought into LDS by global_load_lds_dwordx4 -- no registers and no ds_write for
// staging, a 3-deep ring, ONE barrier per 32-deep K step (gemm_f32_kernel: two barriers and a register -> LDS write pass
// per step, the MFMA pipe 57 % busy).
// LDS image of a stage: per 64-row block and 4-k slot kk one 1 KiB piece [64 rows][4 floats] -- lane = row, source =
// 16 contiguous bytes of that row: lane-linear for the DMA and conflict-free for the fragment reads.  A lane (row i,
// k half h) reads the whole 16-byte slot of its row and picks component 2 m + h for the m-th MFMA of the slot
// (v_cndmask on the lane half), which keeps the contraction order k = 0, 1, 2, ... .
// Block = 4 waves = 64 x 256 outputs (1 x 4 waves of 64 x 64), two 40 KiB stages: two blocks per CU, so one block's
// prologue / epilogue overlaps the other's MFMAs (an 8-wave 128 x 256 block with three stages, one per CU, measured
// 80 TFLOP/s against gemm_f32_kernel's 87).
// The DMA is issued from inline asm with counted vmcnt (see csrc/hamming_mfma.hip for why).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void gemm_lds_dma16(const void *gptr, uint32_t lds_byte_offset) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte_offset) : "memory", "m0");
}
#pragma clang diagnostic pop

constexpr int DMA_BM = 64, DMA_BN = 256, DMA_BK = 32, DMA_STAGES = 2, DMA_WAVES = 4;
constexpr int DMA_STAGE_BYTES = (DMA_BM + DMA_BN) * DMA_BK * 4;          // 40 KiB: two stages = 80 KiB, two blocks per CU
constexpr int DMA_APIECES = DMA_BM / 64 * (DMA_BK / 4);                  // 8 one-KiB pieces of x per stage
constexpr int DMA_PIECES = (DMA_BM + DMA_BN) / 64 * (DMA_BK / 4);        // 40 pieces per stage
constexpr int DMA_PPW = DMA_PIECES / DMA_WAVES;                          // 10 per wave

template <int EPI>
__global__ __launch_bounds__(DMA_WAVES * 64, 2) void gemm_dma_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = 0, wn = wv;                              // 1 x 4 waves of 64 x 64 outputs
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * DMA_BM;
    const int n0 = blockIdx.y * DMA_BN;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(dsm);
    const int nk1 = g.K / DMA_BK, nk2 = (g.x2 ? g.K2 : 0) / DMA_BK, nks = nk1 + nk2;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // piece p = wv + 4 q of a stage: p < 8: x rows lane, slot p;  else W rows ((p - 8) >> 3) * 64 + lane, slot p & 7.
    // Per-lane source pointers of this wave's pieces, at k = 0 of the current operand pair (recomputed once, when the
    // contraction moves from (x, W) to (x2, W2)); a K step only adds its byte offset.
    const unsigned char *src[DMA_PPW];
    auto sources = [&](bool second) {
        const float *X = second ? g.x2 : g.x;
        const float *Wp = second ? g.W2 : g.W;
        const int K = second ? g.K2 : g.K;
        const int ldw = second ? g.ldw2 : g.ldw;
#pragma unroll
        for (int q = 0; q < DMA_PPW; ++q) {
            const int p = wv + DMA_WAVES * q;               // wave-uniform
            const int kk = p & 7;
            if (p < DMA_APIECES) {
                int64_t row = m0 + lane;
                row = row < g.M ? row : g.M - 1;            // rows past the end are clamped, their results never stored
                src[q] = reinterpret_cast<const unsigned char *>(X + row * K + kk * 4);
            } else {
                int n = n0 + ((p - DMA_APIECES) >> 3) * 64 + lane;
                n = n < g.N ? n : g.N - 1;
                src[q] = reinterpret_cast<const unsigned char *>(Wp + (int64_t)n * ldw + kk * 4);
            }
        }
    };
    auto prefetch = [&](int ks, int buf) {
        const int kbytes = (ks >= nk1 ? ks - nk1 : ks) * (DMA_BK * 4);
#pragma unroll
        for (int q = 0; q < DMA_PPW; ++q)
            gemm_lds_dma16(src[q] + kbytes, lds_base + (uint32_t)(buf * DMA_STAGE_BYTES + (wv + DMA_WAVES * q) * 1024));
    };

    sources(nk1 == 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // nothing of the compiler's in flight (vmcnt(0), builtin form)
    if (nks > 0) prefetch(0, 0);
    int buf = 0;
    for (int ks = 0; ks < nks; ++ks) {
        // stage ks has landed (this wave's pieces; after the barrier everybody's) and everybody is done with the other
        // buffer, which the next K step's DMA overwrites while this step is multiplied
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ks + 1 < nks) {
            if (ks + 1 == nk1) sources(true);
            prefetch(ks + 1, buf ^ 1);
        }
        const unsigned char *sA = dsm + buf * DMA_STAGE_BYTES + (wm * 8) * 1024 + li * 16;
        const unsigned char *sB = dsm + buf * DMA_STAGE_BYTES + (DMA_APIECES + wn * 8) * 1024 + li * 16;
        float4 fa[2][2], fb[2][2];                          // [slot parity][tile]
        auto frags = [&](int kk, int s) {
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[s][a] = *reinterpret_cast<const float4 *>(sA + kk * 1024 + a * 512);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[s][b] = *reinterpret_cast<const float4 *>(sB + kk * 1024 + b * 512);
        };
        frags(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int kk = 0; kk < DMA_BK / 4; ++kk) {
            const int s = kk & 1;
            if (kk + 1 < DMA_BK / 4) {
                frags(kk + 1, s ^ 1);                        // next slot's fragments before this slot's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float av[2], bv[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) av[a] = m == 0 ? (lh ? fa[s][a].y : fa[s][a].x) : (lh ? fa[s][a].w : fa[s][a].z);
#pragma unroll
                for (int b = 0; b < 2; ++b) bv[b] = m == 0 ? (lh ? fb[s][b].y : fb[s][b].x) : (lh ? fb[s][b].w : fb[s][b].z);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
        }
        buf ^= 1;
    }

    // ------------------------------ epilogue (as gemm_f32_kernel) ------------------------------------------------
    // C layout (32x32 tile): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (EPI == 1) {
        uint32_t *codes32 = reinterpret_cast<uint32_t *>(g.codes);
        const int words = g.cs >> 2;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int colbase = n0 + (wn * 2 + b) * 32;
                const bool col_ok = colbase + li < g.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t mask = __ballot(col_ok && acc[a][b][r] >= 0.f);   // bit = (xt >= 0)
                    const int64_t row = m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (li == 0 && row < g.M && colbase < g.N)
                        codes32[row * words + (colbase >> 5)] = lh ? (uint32_t)(mask >> 32) : (uint32_t)mask;
                }
            }
        return;
    }
    const bool relu = g.flags & PS_RELU, l2 = g.flags & PS_L2NORM;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + (wn * 2 + b) * 32 + li;
        const float bias = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][b][r] + bias;
                if (relu) v = v > 0.f ? v : 0.f;
                if (col >= g.N) v = 0.f;
                acc[a][b][r] = v;
            }
    }
    if (l2) {   // F.normalize(p=2, dim=1, eps=1e-12): the block holds whole rows (N <= 256); partial sums of the 4 column waves via LDS
        float *sRed = reinterpret_cast<float *>(dsm);         // the ring is drained: every wave passed the last barrier ...
        __syncthreads();                                      // ... and finished its fragment reads
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ss = 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) ss = fmaf(acc[a][b][r], acc[a][b][r], ss);
                ss = half_sum32(ss);
                const int rowl = (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (li == 16) sRed[rowl * 4 + wn] = ss;
            }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowl = (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float ss = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) ss += sRed[rowl * 4 + w];
                float nrm = sqrtf(ss);
                nrm = nrm > 1e-12f ? nrm : 1e-12f;
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b][r] = acc[a][b][r] / nrm;
            }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= g.M) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = n0 + (wn * 2 + b) * 32 + li;
                if (col < g.N) g.y[row * g.N + col] = acc[a][b][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------
// gemm_shard_kernel: the GEMM of ONE SHARD of a multi-GPU job (59 047 items / 8 ranks = 7 381 rows) and of every other call
// with so few rows that the launch is one or two workgroups per CU.  Such a launch is not bound by the matrix pipe but by the
// serial step of its lone workgroup: gemm_f32_kernel's 32 x 256 tile spent 2 668 cycles in a K step's 32 MFMAs (2 048 of pipe
// time), 936 in barrier / register -> LDS staging / barrier, and had nobody else on the CU to fill either gap (tools/gemm_trace.py
// 7381 with PS_GEMM_PERSIST=0).  Here a K step is ONE barrier: both operands come into a ring of STAGES LDS images by
// global_load_lds_dwordx4 (no staging registers, no ds_write), requested STAGES - 1 steps ahead with counted vmcnt, the first
// fragments of a step are read before the next stage is requested, and a step's MFMA chain never waits for memory it did not
// ask for a whole step earlier.  Tile 32 x 256, 4 waves of 32 x 64, v_mfma_f32_32x32x2_f32 with k ascending and one accumulator
// per output: the same fmaf chain, bit for bit, as every other kernel of this file.
// LDS image of a stage (36 one-KiB pieces, lane-linear for the DMA):
//   piece j < 4      : x, lane = (row r = lane & 31, 16-byte chunk 2 j + (lane >> 5)) of the step's 128-byte row slab
//   piece 4 + 8 rb + c: W rows rb * 64 + lane, chunk c
// A lane (row i, k half h) of a natural-order operand takes the two floats h and 2 + h of a chunk (one ds_read2_b32: k = 4 c + h
// and 4 c + 2 + h); of a weight stored in image order (PS_WPERM: k 0 2 4 6 | 1 3 5 7 per group of 8) it takes chunk 2 g + h whole
// (one ds_read_b128 = its four k of the group).
constexpr int SH_BM = 32, SH_BN = 256, SH_BK = 32;
constexpr int SH_STAGE_BYTES = (SH_BM + SH_BN) * SH_BK * 4;              // 36 KiB
constexpr int SH_TAIL_BYTES = (SH_BM * 4 + SH_BM * 4) * 4;               // sRed + sNrm of the epilogue

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void shard_lds_dma16(uint32_t voff, const void *sbase, uint32_t lds_byte_offset) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_offset) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int EPI, bool WP, int STAGES>
__global__ __launch_bounds__(256, STAGES == 2 ? 2 : 1) void gemm_shard_kernel(GemmArgs g) {
    static_assert(STAGES == 2 || STAGES == 3, "ring of two or three images");
    extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv;                                       // 1 x 4 waves of 32 x 64 outputs
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * SH_BM;
    const int n0 = blockIdx.y * SH_BN;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(dsm);
    const int nk1 = g.K / SH_BK, nk2 = (g.x2 ? g.K2 : 0) / SH_BK, nks = nk1 + nk2;
    float *sRed = reinterpret_cast<float *>(dsm + STAGES * SH_STAGE_BYTES), *sNrm = sRed + SH_BM * 4;

    f32x16 acc[1][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;

    // this wave's nine pieces of a stage: piece wv of x, pieces 4 + wv + 4 (q - 1), q = 1..8, of W.  Per lane the byte offset of
    // its 16 bytes from the tile's first row at k = 0 of the operand pair (rows past the end are clamped: never stored);
    // a K step moves the scalar base only.
    uint32_t voff[9];
    auto offsets = [&](bool second) {
        const int K = second ? g.K2 : g.K, ldw = second ? g.ldw2 : g.ldw;
        int64_t row = m0 + li;
        row = (row < g.M ? row : g.M - 1) - m0;
        voff[0] = (uint32_t)row * (uint32_t)K * 4u + (uint32_t)(2 * wv + lh) * 16u;
#pragma unroll
        for (int q = 1; q < 9; ++q) {
            const int idx = wv + 4 * (q - 1);                // wave-uniform
            int n = n0 + (idx >> 3) * 64 + lane;
            n = (n < g.N ? n : g.N - 1) - n0;
            voff[q] = (uint32_t)n * (uint32_t)ldw * 4u + (uint32_t)(idx & 7) * 16u;
        }
    };
    // scalar cursors: the tile's first row at the K step to request next, of the current operand pair
    const float *Xc = g.x + m0 * g.K, *Wc = g.W + (int64_t)n0 * g.ldw;
    const float *X2c = g.x2 ? g.x2 + m0 * g.K2 : nullptr, *W2c = g.x2 ? g.W2 + (int64_t)n0 * g.ldw2 : nullptr;
    uint32_t issue_lds = 0;
    auto issue_begin = [&](int ks, int buf) {
        if (ks == nk1) {                                     // block-uniform: the contraction moves on to (x2, W2)
            offsets(true);
            Xc = X2c;
            Wc = W2c;
        }
        issue_lds = lds_base + (uint32_t)(buf * SH_STAGE_BYTES + wv * 1024);
    };
    auto issue_piece = [&](int q) {                           // q = 0: this wave's piece of x; 1..8: its pieces of W
        shard_lds_dma16(voff[q], q == 0 ? Xc : Wc, issue_lds + (uint32_t)q * 4096u);
        if (q == 8) {
            Xc += SH_BK;
            Wc += SH_BK;
        }
    };
    auto issue = [&](int ks, int buf) {
        issue_begin(ks, buf);
#pragma unroll
        for (int q = 0; q < 9; ++q) issue_piece(q);
    };

#if PS_GEMM_DEBUG & 64
    // timeline (tools/gemm_shard_trace.py): sums kept in registers and stored after the loop -- a store inside it would
    // join the vmcnt FIFO the counted waits rely on
    const unsigned long long tr_start = __builtin_readcyclecounter();
    unsigned long long tr_wait = 0, tr_first = 0, tr_stream = 0, tr_a = 0, tr_b = 0, tr_c = 0;
#endif
    offsets(false);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // nothing of the compiler's in flight (vmcnt(0), builtin form)
    issue(0, 0);
    if (STAGES == 3 && nks > 1) issue(1, 1);
    int buf = 0;
    for (int ks = 0; ks < nks; ++ks) {
#if PS_GEMM_DEBUG & 64
        tr_a = __builtin_readcyclecounter();
        if (ks > 0) tr_stream += tr_a - tr_c;
#endif
        // stage ks has landed -- this wave's pieces; behind the barrier everybody's -- and everybody is done with the image of
        // stage ks - 1, which the request issued below overwrites
        if (STAGES == 3 && ks + 1 < nks) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if PS_GEMM_DEBUG & 64
        tr_b = __builtin_readcyclecounter();
        tr_wait += tr_b - tr_a;
#endif
        const unsigned char *sA = dsm + buf * SH_STAGE_BYTES + li * 16 + lh * 4;
        const unsigned char *sB = dsm + buf * SH_STAGE_BYTES + (4 + wn * 8) * 1024 + li * 16 + (WP ? lh * 1024 : lh * 4);
        float av[2][4];                                      // [slot parity][t]: k = 8 g + 2 t + lh
        f32x4 bv[2][2];                                      // [slot parity][tile]: component t
        auto frags = [&](int grp, int s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *p = reinterpret_cast<const float *>(sA + grp * 1024 + h * 512);
                av[s][2 * h] = p[0];
                av[s][2 * h + 1] = p[2];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (WP) {
                    bv[s][b] = *reinterpret_cast<const f32x4 *>(sB + 2 * grp * 1024 + b * 512);
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float *p = reinterpret_cast<const float *>(sB + (2 * grp + h) * 1024 + b * 512);
                        bv[s][b][2 * h] = p[0];
                        bv[s][b][2 * h + 1] = p[2];
                    }
                }
            }
        };
        // Issue order of the step, pinned (sched_barrier): the first group's fragments, then per group the next group's
        // fragments, and the nine requests of the stage STAGES - 1 steps ahead one at a time behind the step's first nine MFMAs --
        // issued in one burst behind the barrier they kept the matrix pipe idle for ~500 cycles of every step
        // (tools/gemm_shard_trace.py: barrier -> first fragments 687 cycles per step, 2 048 of MFMA)
        const int nx = ks + STAGES - 1;
        frags(0, 0);
        auto stream = [&](auto more_tag) __attribute__((always_inline)) {
            constexpr bool MORE = decltype(more_tag)::value;
            if (MORE) issue_begin(nx, buf == 0 ? STAGES - 1 : buf - 1);   // (buf + STAGES - 1) % STAGES: the image of stage ks - 1
            __builtin_amdgcn_sched_barrier(0);
#if PS_GEMM_DEBUG & 64
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the step's first fragments are here
            tr_c = __builtin_readcyclecounter();
            tr_first += tr_c - tr_b;
#endif
#pragma unroll
            for (int grp = 0; grp < SH_BK / 8; ++grp) {
                const int s = grp & 1;
                if (grp + 1 < SH_BK / 8) {
                    frags(grp + 1, s ^ 1);                       // next group's fragments before this group's MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][t], bv[s][b][t], acc[0][b], 0, 0, 0);
                        const int q = grp * 8 + t * 2 + b;
                        if (MORE && q < 9) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(q);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
        };
        if (nx < nks) stream(std::true_type{});              // block-uniform
        else stream(std::false_type{});
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
#if PS_GEMM_DEBUG & 64
    if (lane == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
        unsigned long long *t = ps_gemm_trace_buf + ((size_t)blockIdx.x * 4 + wv) * 48;
        t[0] = tr_start; t[1] = tr_wait; t[2] = tr_first; t[3] = tr_stream + (__builtin_readcyclecounter() - tr_c);
        t[46] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4); t[47] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);
    }
#endif
    gemm_epilogue<1, 4, 1, 2, EPI>(g, acc, m0, n0, sRed, sNrm);
}

__global__ void l2norm_rows_kernel(float *y, int64_t M, int N) {   // N > 256 only
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t m = wave; m < M; m += nw) {
        float ss = 0.f;
        for (int n = lane; n < N; n += 64) ss = fmaf(y[m * N + n], y[m * N + n], ss);
        ss = ps_wave_sum_f32(ss);
        float nrm = sqrtf(ss);
        nrm = nrm > 1e-12f ? nrm : 1e-12f;
        for (int n = lane; n < N; n += 64) y[m * N + n] = y[m * N + n] / nrm;
    }
}

// workgroups a persistent launch keeps resident: the occupancy of the instantiation x the CUs of the device.  `slots` is the
// CALLER's cache (one per kernel instantiation: every instantiation has the same function-pointer type, so a static in here
// would be shared by all of them -- r03's first version was, and gave every tile shape the grid of the first one used)
template <typename K>
int persistent_slots(K kernel, PsPerDevice &slots) {
    int dv = 0;
    if (hipGetDevice(&dv) != hipSuccess || dv < 0 || dv >= 64) return 0;
    int n = slots.get(dv);
    if (n == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dv) != hipSuccess || cus <= 0) return 0;
        n = per_cu * cus;
        slots.set(dv, n);
    }
    return n;
}

template <int WM, int WN, int TM, int TN, int EPI>
int launch_persistent(const GemmArgs &g, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int64_t tiles_m = ps_cdiv(g.M, BM);
    const int tiles_n = (int)ps_cdiv(g.N, BN);
    const int64_t ntiles = tiles_m * tiles_n;
    if (ntiles > 0x7fffffff) return PS_EUNSUPPORTED;
    static PsPerDevice slot_cache;
    const int slots = persistent_slots(gemm_f32_pkernel<WM, WN, TM, TN, 32, EPI>, slot_cache);
    if (slots <= 0) return PS_ELAUNCH;
    const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    hipLaunchKernelGGL((gemm_f32_pkernel<WM, WN, TM, TN, 32, EPI>), dim3(grid), dim3(256), 0, st, g, tiles_n, (int)ntiles);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

template <int EPI, bool FAST>
int launch_gemm_v(const GemmArgs &g, hipStream_t st) {
    // aligned operands: the persistent form of the same tiles (PS_GEMM_PERSIST=0: the one-tile-per-workgroup kernels)
    // (r03, measured and not kept: eight waves per 32 x 256 tile for shards of <= 512 row tiles -- one 32 x 32 MFMA tile per wave
    // halves a wave's MFMA chain, but the launch then waits for the next K step's operands instead, which are only requested one
    // step ahead: 7 381 rows 60.7 -> 55.8 TFLOP/s, 14 762 rows 80.4 -> 72.7, LSH projection 83 -> 69.  Nor is it the operand latency:
    // a persistent variant that requests the slabs TWO K steps ahead into two register sets was equal at 7 381 rows (61.6 / 61.2) and
    // 4-12 % slower at 14 762 / 20 000; a shard-sized tile is bound by its own serial step: two barriers, the staging write, the
    // first fragment reads and 32 MFMAs per wave with nobody else on the CU to fill the gaps.)
    if (FAST && g.grp == nullptr && g.N > 128 && (g.x2 == nullptr || g.K2 % 32 == 0)) {
        const char *pe = getenv("PS_GEMM_PERSIST");
        const bool persist = pe == nullptr || atoi(pe) != 0;
        if (persist) {
            // measured r03 (tools/gemm_probe.py --env PS_GEMM_PERSIST=0,1, M = 59 047 / 10 000): 64 x 128 tiles +2-3 % / +9 %,
            // 32 x 256 tiles (small M, fused norm) +7-9 %; the 64 x 256 fused-norm tile does not fit the registers in this form
            // (256 VGPRs, 35 spilled: 103 -> 92 TFLOP/s) and keeps the one-tile kernel
            if (!(g.flags & PS_L2NORM)) return launch_persistent<2, 2, 1, 2, EPI>(g, st);
            if (g.M < 64 * 384) return launch_persistent<1, 4, 1, 2, EPI>(g, st);
        }
    }
    if (g.N <= 64) {
        dim3 grid((unsigned)ps_cdiv(g.M, 64), 1);
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1, 1, 32, EPI, FAST>), grid, dim3(256), 0, st, g);
    } else if (g.N <= 128) {
        dim3 grid((unsigned)ps_cdiv(g.M, 64), 1);
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1, 2, 32, EPI, FAST>), grid, dim3(256), 0, st, g);
    } else {
        if (!(g.flags & PS_L2NORM)) {
            // no row reduction in the epilogue: 64 x 128 tiles (twice the resident waves) measured 2-7 % faster (r02, with the
            // spread loads in the 64 x 256 kernel: input projection equal, LSH projection still 7 % faster)
            dim3 grid((unsigned)ps_cdiv(g.M, 64), (unsigned)ps_cdiv(g.N, 128));
            hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1, 2, 32, EPI, FAST>), grid, dim3(256), 0, st, g);
        } else if (g.M < 64 * 384) {
            // a shard of a multi-GPU job (59 047 / 8 rows = 116 tiles of 64 rows) would leave half of the CUs idle:
            // 32-row tiles
            dim3 grid((unsigned)ps_cdiv(g.M, 32), (unsigned)ps_cdiv(g.N, 256));
            hipLaunchKernelGGL((gemm_f32_kernel<1, 4, 1, 2, 32, EPI, FAST>), grid, dim3(256), 0, st, g);
        } else {
            // 64 x 256 tiles (whole rows for the fused L2 norm); 32-row tiles measured 8 % slower, BK = 16 6 % slower,
            // 128 x 256 tiles / 8 waves per block / a double-buffered BK = 16 image no faster
            dim3 grid((unsigned)ps_cdiv(g.M, 64), (unsigned)ps_cdiv(g.N, 256));
            hipLaunchKernelGGL((gemm_f32_kernel<1, 4, 2, 2, 32, EPI, FAST>), grid, dim3(256), 0, st, g);
        }
    }
    PS_CHECK_LAUNCH();
    return PS_OK;
}

bool aligned_operand(const float *p, int K, int ld) {
    return p == nullptr || (reinterpret_cast<size_t>(p) % 16 == 0 && K % 32 == 0 && ld % 4 == 0);
}

template <int EPI, bool WP, int STAGES>
int launch_shard(const GemmArgs &g, hipStream_t st) {
    constexpr int lds = STAGES * SH_STAGE_BYTES + SH_TAIL_BYTES;
    static PsPerDevice attr_done;                             // per instantiation
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return PS_ELAUNCH;
    if (!attr_done.get(devid)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_shard_kernel<EPI, WP, STAGES>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return PS_ELAUNCH;
        attr_done.set(devid, 1);
    }
    dim3 grid((unsigned)ps_cdiv(g.M, SH_BM), (unsigned)ps_cdiv(g.N, SH_BN));
    hipLaunchKernelGGL((gemm_shard_kernel<EPI, WP, STAGES>), grid, dim3(256), lds, st, g);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

// Which launches go to gemm_shard_kernel (measured r04 on 7 381 / 14 762-row shards, tools/gemm_probe.py --env PS_GEMM_SHARD=0,2,3,
// image-order weights): the layer GEMM (K = 256 + 256, fused norm) 30.3 -> 26.9 us with the ring of three at 231 tiles and
// 48.2 -> 45.8 us with the ring of two at 462, the output projection (K = 256, fused norm) 18.3 -> 17.1 / 26.4 -> 26.0; launches
// WITHOUT the whole-row epilogue are no faster than the 64 x 128 tiles they already use (input projection 11.3 vs 11.1 us,
// LSH projection 23.4 vs 23.3 at 7 381 rows and 39.9 vs 46.6 at 14 762) and stay there.  The same ring with 64 x 256 tiles for
// launches of MANY rows measured 4-13 % SLOWER than the register-staged kernels (59 047 rows: layer GEMM 104.5 -> 100.0 TFLOP/s,
// LSH projection 111.4 -> 96.7; tools/experiments/r04_gemm_dma_ring_64row_tiles.patch): two workgroups per CU already hide the staging.
constexpr int64_t SHARD_TILES_RING3 = 256, SHARD_TILES_RING2 = 512;

template <int EPI>
int launch_gemm(const GemmArgs &g, hipStream_t st) {
    const bool fast = aligned_operand(g.x, g.K, g.K) && aligned_operand(g.W, g.K, g.ldw) &&
                      (g.x2 == nullptr || (aligned_operand(g.x2, g.K2, g.K2) && aligned_operand(g.W2, g.K2, g.ldw2)));
    {
        // PS_GEMM_SHARD: 0 = never, 2 / 3 = that ring depth whenever the operands allow (tests, experiments); default: by size
        const char *se = getenv("PS_GEMM_SHARD");
        const int mode = se ? atoi(se) : -1;
        const int64_t tiles = ps_cdiv(g.M, SH_BM) * ps_cdiv(g.N, SH_BN);
        const bool fits = (int64_t)g.ldw * 1024 < 0x7fffffff && (g.x2 == nullptr || (int64_t)g.ldw2 * 1024 < 0x7fffffff) &&
                          (int64_t)g.K * 128 < 0x7fffffff && (g.x2 == nullptr || (int64_t)g.K2 * 128 < 0x7fffffff);
        const bool by_size = EPI == 0 && (g.flags & PS_L2NORM) && tiles <= SHARD_TILES_RING2;
        if (fast && fits && g.grp == nullptr && g.N > 128 && mode != 0 && (mode > 0 || by_size)) {
            const bool wp = (g.flags & PS_WPERM) != 0;
            if (mode == 3 || (mode < 0 && tiles <= SHARD_TILES_RING3))
                return wp ? launch_shard<EPI, true, 3>(g, st) : launch_shard<EPI, false, 3>(g, st);
            return wp ? launch_shard<EPI, true, 2>(g, st) : launch_shard<EPI, false, 2>(g, st);
        }
    }
    // LDS-DMA kernel (opt-in, PS_GEMM_DMA=1: measured SLOWER than gemm_f32_kernel on MI355X, 82 vs 87.5 TFLOP/s for the
    // layer GEMMs and 85 vs 94 for the LSH projection -- one barrier per K step and no staging registers do not pay for
    // the doubled fragment reads, the lane-half selects and the 16-bytes-per-row DMA pieces; kept because it is tested
    // bit-identical and documents the experiment): aligned operands, whole 256-column tiles, many rows
    const bool use_dma = getenv("PS_GEMM_DMA") != nullptr;
    if (fast && use_dma && !(g.flags & PS_WPERM) && g.N % DMA_BN == 0 && g.M >= 64 * 384) {
        static PsPerDevice attr_done;
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return PS_ELAUNCH;
        if (!attr_done.get(devid)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_dma_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    DMA_STAGES * DMA_STAGE_BYTES) != hipSuccess) return PS_ELAUNCH;
            attr_done.set(devid, 1);
        }
        dim3 grid((unsigned)ps_cdiv(g.M, DMA_BM), (unsigned)(g.N / DMA_BN));
        hipLaunchKernelGGL((gemm_dma_kernel<EPI>), grid, dim3(DMA_WAVES * 64), DMA_STAGES * DMA_STAGE_BYTES, st, g);
        PS_CHECK_LAUNCH();
        return PS_OK;
    }
    if ((g.flags & PS_WPERM) && !fast) return PS_EINVAL;      // image-order weights exist for aligned operands only (K % 32 == 0)
    return fast ? launch_gemm_v<EPI, true>(g, st) : launch_gemm_v<EPI, false>(g, st);
}

}  // namespace

// Grouped x W^T: row block b (64 rows of x, M a multiple of 64) is multiplied with W rows [grp[3b], grp[3b] + grp[3b+1]) and its
// [64, grp[3b+1]] result written at y + grp[3b+2] (row stride grp[3b+1]); max_cols = the largest grp[3b+1].  Same k-ordered fp32
// chain per output as ps_linear.  Internal to the library (ps_ivf_topk).
int psi_linear_grouped(const float *x, int64_t M, int K, const float *W, int ldw, float *y, const int64_t *grp, int max_cols,
                       ps_stream_t stream) {
    if (M <= 0 || M % 64 != 0 || K <= 0 || max_cols <= 0 || !x || !W || !y || !grp || ldw < K) return PS_EINVAL;
    GemmArgs g{x, M, K, W, ldw, nullptr, 0, nullptr, 0, nullptr, max_cols, 0, y, nullptr, 0, grp};
    hipStream_t st = ps_stream(stream);
    dim3 grid((unsigned)(M / 64), (unsigned)ps_cdiv(max_cols, 128));
    if (aligned_operand(x, K, K) && aligned_operand(W, K, ldw))
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1, 2, 32, 0, true>), grid, dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1, 2, 32, 0, false>), grid, dim3(256), 0, st, g);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_linear(const float *x, int64_t M, int K, const float *W, int ldw, const float *b, int N,
                         const float *x2, int K2, const float *W2, int ldw2, int flags, float *y, ps_stream_t stream) {
    if (M < 0 || K <= 0 || N <= 0 || ldw < K) return PS_EINVAL;
    if (M == 0) return PS_OK;
    if (!x || !W || !y) return PS_EINVAL;
    if (x2 && (!W2 || K2 <= 0 || ldw2 < K2)) return PS_EINVAL;
    if (M > (int64_t)0x7fffffff * 64) return PS_EINVAL;
    hipStream_t st = ps_stream(stream);
    GemmArgs g{x, M, K, W, ldw, x2, x2 ? K2 : 0, W2, ldw2, b, N, flags, y, nullptr, 0};
    const bool fused_norm = N <= 256;
    if (!fused_norm) g.flags &= ~PS_L2NORM;
    int rc = launch_gemm<0>(g, st);
    if (rc != PS_OK) return rc;
    if ((flags & PS_L2NORM) && !fused_norm) {
        int64_t grid = ps_cdiv(M, 4);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)grid), dim3(256), 0, st, y, M, N);
        PS_CHECK_LAUNCH();
    }
    return PS_OK;
}

// out[r][8 g + j] = W[r][8 g + p(j)], p = 0 2 4 6 1 3 5 7: the order in which the GEMM's LDS image holds an 8-k group
__global__ void permute_k_kernel(const float *__restrict__ W, int64_t rows, int K, int ld, float *__restrict__ out) {
    const int64_t total = rows * (K / 8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (K / 8);
        const int g8 = (int)(i % (K / 8));
        const float *src = W + r * ld + g8 * 8;
        float *dst = out + r * K + g8 * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[j] = src[2 * j]; dst[4 + j] = src[2 * j + 1]; }
    }
}

extern "C" int ps_permute_k(const float *W, int64_t rows, int K, int ld, float *out, ps_stream_t stream) {
    if (rows < 0 || K <= 0 || K % 8 != 0 || ld < K) return PS_EINVAL;
    if (rows == 0) return PS_OK;
    if (!W || !out) return PS_EINVAL;
    int64_t grid = ps_cdiv(rows * (K / 8), 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(permute_k_kernel, dim3((unsigned)grid), dim3(256), 0, ps_stream(stream), W, rows, K, ld, out);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_lsh_encode(const float *x, int64_t N, int D, const float *A, int nbits, uint8_t *codes, int flags,
                             ps_stream_t stream) {
    if (N < 0 || D <= 0 || nbits <= 0 || (flags & ~PS_WPERM)) return PS_EINVAL;
    if (nbits % 32 != 0) return PS_EUNSUPPORTED;      // codes are written as whole 32-bit ballot words
    if (N == 0) return PS_OK;
    if (!x || !A || !codes || reinterpret_cast<size_t>(codes) % 4 != 0) return PS_EINVAL;
    GemmArgs g{x, N, D, A, D, nullptr, 0, nullptr, 0, nullptr, nbits, flags, nullptr, codes, nbits / 8};
    return launch_gemm<1>(g, ps_stream(stream));
}
