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
// rows, 64 x 128 otherwise, 32 x 256 for small shards; K step BK = 32 (BK = 16 was measured 6 % slower: the 64 x 256
// tile is register limited to 2 waves per SIMD, not LDS limited).
// LDS image per operand row: [BK/8 groups of 8 k][lane-half h][4] so that the lane (row i, half h)
// fetches its four k values of one group (k = 8g + 2t + h, t = 0..3) with a single ds_read_b128;
// row stride BK+4 floats keeps the b128 reads bank-conflict free.  The k-permutation is done
// in registers while staging (two float4 global loads per row-group), global loads for the next
// K step are in flight while the current one is multiplied, and inside a K step the LDS fragments of
// k-group g+1 are requested before the 16 MFMAs of group g (order pinned with sched_group_barrier).
#include "ps_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


struct GemmArgs {
    const float *x;  int64_t M; int K;  const float *W;  int ldw;
    const float *x2; int K2;            const float *W2; int ldw2;
    const float *bias; int N; int flags;
    float *y;            // EPI 0
    uint8_t *codes; int cs;  // EPI 1
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

// Sum over the 32 lanes of each wave half, valid in lanes 16..31 / 48..63, with DPP only (the LDS crossbar that
// __shfl_xor goes through is shared by the CU's waves: 160 shuffles per wave cost ~17 us per launch).
#define PS_DPP_ADD(v, ctrl, rows) \
    (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rows), 0xf, true))
__device__ __forceinline__ float half_sum32(float v) {
    PS_DPP_ADD(v, 0xB1, 0xf);    // quad_perm [1,0,3,2]: lane ^ 1
    PS_DPP_ADD(v, 0x4E, 0xf);    // quad_perm [2,3,0,1]: lane ^ 2
    PS_DPP_ADD(v, 0x141, 0xf);   // row_half_mirror: the other quad of the 8
    PS_DPP_ADD(v, 0x140, 0xf);   // row_mirror: the other 8 of the 16
    PS_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 -> rows 1 and 3 add the row before them
    return v;
}
#undef PS_DPP_ADD

// FAST: every operand is 16-B aligned with K % BK == 0 -> unconditional float4 loads (rows past the end are
// clamped to the last valid row; their results are never stored), so nothing branches or waits inside the
// fetch and the loads stay in flight under the MFMAs.  The general variant predicates every element.
template <int WM, int WN, int TM, int TN, int BK, int EPI, bool FAST>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr int GRP = BK / 8;                        // 8-k groups per row
    constexpr int LDS_STRIDE = BK + 4;                 // floats; 144 B (BK 32) / 80 B (BK 16): b128 reads conflict free
    constexpr int A_ITEMS = (BM * GRP + NT - 1) / NT, B_ITEMS = (BN * GRP + NT - 1) / NT;   // (row, group) items per thread
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_STRIDE + BM * WN];
    float *sA = smem, *sB = smem + BM * LDS_STRIDE, *sRed = smem + (BM + BN) * LDS_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int li = lane & 31, lh = lane >> 5;

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

        float ra[A_ITEMS][8], rb[B_ITEMS][8];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int q = 0; q < A_ITEMS; ++q) {
                // FAST: no guards -- when the tile has fewer items than threads the surplus threads duplicate an
                // item (identical loads, identical LDS writes)
                const int it = FAST ? (tid + NT * q) % (BM * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                const int64_t m = m0 + row;
                if (FAST) {
                    {
                        const float *src = X + (m < g.M ? m : g.M - 1) * K + k0 + grp * 8;
                        const float4 a = *reinterpret_cast<const float4 *>(src);
                        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
                        ra[q][0] = a.x; ra[q][1] = a.y; ra[q][2] = a.z; ra[q][3] = a.w;
                        ra[q][4] = b.x; ra[q][5] = b.y; ra[q][6] = b.z; ra[q][7] = b.w;
                    }
                } else if (it < BM * GRP) load8(X + m * K, m < g.M, k0 + grp * 8, K, vecA, ra[q]);
            }
#pragma unroll
            for (int q = 0; q < B_ITEMS; ++q) {
                const int it = FAST ? (tid + NT * q) % (BN * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                const int n = n0 + row;
                if (FAST) {
                    {
                        const float *src = Wp + (int64_t)(n < g.N ? n : g.N - 1) * ldw + k0 + grp * 8;
                        const float4 a = *reinterpret_cast<const float4 *>(src);
                        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
                        rb[q][0] = a.x; rb[q][1] = a.y; rb[q][2] = a.z; rb[q][3] = a.w;
                        rb[q][4] = b.x; rb[q][5] = b.y; rb[q][6] = b.z; rb[q][7] = b.w;
                    }
                } else if (it < BN * GRP) load8(Wp + (int64_t)n * ldw, n < g.N, k0 + grp * 8, K, vecB, rb[q]);
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int q = 0; q < A_ITEMS; ++q) {
                const int it = FAST ? (tid + NT * q) % (BM * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                if (!FAST && it >= BM * GRP) continue;
                float *d = sA + row * LDS_STRIDE + grp * 8;
                *reinterpret_cast<float4 *>(d) = make_float4(ra[q][0], ra[q][2], ra[q][4], ra[q][6]);
                *reinterpret_cast<float4 *>(d + 4) = make_float4(ra[q][1], ra[q][3], ra[q][5], ra[q][7]);
            }
#pragma unroll
            for (int q = 0; q < B_ITEMS; ++q) {
                const int it = FAST ? (tid + NT * q) % (BN * GRP) : tid + NT * q, row = it / GRP, grp = it % GRP;
                if (!FAST && it >= BN * GRP) continue;
                float *d = sB + row * LDS_STRIDE + grp * 8;
                *reinterpret_cast<float4 *>(d) = make_float4(rb[q][0], rb[q][2], rb[q][4], rb[q][6]);
                *reinterpret_cast<float4 *>(d + 4) = make_float4(rb[q][1], rb[q][3], rb[q][5], rb[q][7]);
            }
        };

        fetch(0);
        for (int k0 = 0; k0 < K; k0 += BK) {
            __syncthreads();           // previous step's fragment reads are done
            stash();
            __syncthreads();
            if (k0 + BK < K) fetch(k0 + BK);
            // fragments of group g+1 are requested BEFORE group g's MFMAs: the two waves of a SIMD run in lockstep
            // (same barriers, round-robin issue), so an LDS round trip taken between groups idles the MFMA pipe for
            // both of them -- measured 65 % pipe use without this prefetch
            float4 fa[2][TM], fb[2][TN];
            auto frags = [&](int grp, int s) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    fa[s][a] = *reinterpret_cast<const float4 *>(sA + ((wm * TM + a) * 32 + li) * LDS_STRIDE + grp * 8 + lh * 4);
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    fb[s][b] = *reinterpret_cast<const float4 *>(sB + ((wn * TN + b) * 32 + li) * LDS_STRIDE + grp * 8 + lh * 4);
            };
            frags(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);           // keep this order: DS reads ...
#pragma unroll
            for (int grp = 0; grp < GRP; ++grp) {
                const int s = grp & 1;
                if (grp + 1 < GRP) {
                    frags(grp + 1, s ^ 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   // ... next group's DS reads first,
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);   // ... then this group's MFMAs
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const float av = t == 0 ? fa[s][a].x : t == 1 ? fa[s][a].y : t == 2 ? fa[s][a].z : fa[s][a].w;
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            const float bv = t == 0 ? fb[s][b].x : t == 1 ? fb[s][b].y : t == 2 ? fb[s][b].z : fb[s][b].w;
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

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
        return;
    }

    const bool relu = g.flags & PS_RELU, l2 = g.flags & PS_L2NORM;
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
    if (l2) {   // F.normalize(p=2, dim=1, eps=1e-12): x / max(||x||, eps); the block holds whole rows
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ss = 0.f;
#pragma unroll
                for (int b = 0; b < TN; ++b) ss = fmaf(acc[a][b][r], acc[a][b][r], ss);
                ss = half_sum32(ss);                                            // over the 32 columns of the half
                const int rowl = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (li == 16) sRed[rowl * WN + wn] = ss;
            }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowl = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float ss = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) ss += sRed[rowl * WN + w];
                float nrm = sqrtf(ss);
                nrm = nrm > 1e-12f ? nrm : 1e-12f;
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b][r] = acc[a][b][r] / nrm;
            }
    }
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

template <int EPI, bool FAST>
int launch_gemm_v(const GemmArgs &g, hipStream_t st) {
    if (g.N <= 64) {
        dim3 grid((unsigned)ps_cdiv(g.M, 64), 1);
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1, 1, 32, EPI, FAST>), grid, dim3(256), 0, st, g);
    } else if (g.N <= 128) {
        dim3 grid((unsigned)ps_cdiv(g.M, 64), 1);
        hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1, 2, 32, EPI, FAST>), grid, dim3(256), 0, st, g);
    } else {
        if (!(g.flags & PS_L2NORM)) {
            // no row reduction in the epilogue: 64 x 128 tiles (twice the resident waves) measured 2-7 % faster
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

template <int EPI>
int launch_gemm(const GemmArgs &g, hipStream_t st) {
    const bool fast = aligned_operand(g.x, g.K, g.K) && aligned_operand(g.W, g.K, g.ldw) &&
                      (g.x2 == nullptr || (aligned_operand(g.x2, g.K2, g.K2) && aligned_operand(g.W2, g.K2, g.ldw2)));
    // LDS-DMA kernel (opt-in, PS_GEMM_DMA=1: measured SLOWER than gemm_f32_kernel on MI355X, 82 vs 87.5 TFLOP/s for the
    // layer GEMMs and 85 vs 94 for the LSH projection -- one barrier per K step and no staging registers do not pay for
    // the doubled fragment reads, the lane-half selects and the 16-bytes-per-row DMA pieces; kept because it is tested
    // bit-identical and documents the experiment): aligned operands, whole 256-column tiles, many rows
    const bool use_dma = getenv("PS_GEMM_DMA") != nullptr;
    if (fast && use_dma && g.N % DMA_BN == 0 && g.M >= 64 * 384) {
        static bool attr_done[64] = {};
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return PS_ELAUNCH;
        if (!attr_done[devid]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_dma_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    DMA_STAGES * DMA_STAGE_BYTES) != hipSuccess) return PS_ELAUNCH;
            attr_done[devid] = true;
        }
        dim3 grid((unsigned)ps_cdiv(g.M, DMA_BM), (unsigned)(g.N / DMA_BN));
        hipLaunchKernelGGL((gemm_dma_kernel<EPI>), grid, dim3(DMA_WAVES * 64), DMA_STAGES * DMA_STAGE_BYTES, st, g);
        PS_CHECK_LAUNCH();
        return PS_OK;
    }
    return fast ? launch_gemm_v<EPI, true>(g, st) : launch_gemm_v<EPI, false>(g, st);
}

}  // namespace

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

extern "C" int ps_lsh_encode(const float *x, int64_t N, int D, const float *A, int nbits, uint8_t *codes,
                             ps_stream_t stream) {
    if (N < 0 || D <= 0 || nbits <= 0) return PS_EINVAL;
    if (nbits % 32 != 0) return PS_EUNSUPPORTED;      // codes are written as whole 32-bit ballot words
    if (N == 0) return PS_OK;
    if (!x || !A || !codes || reinterpret_cast<size_t>(codes) % 4 != 0) return PS_EINVAL;
    GemmArgs g{x, N, D, A, D, nullptr, 0, nullptr, 0, nullptr, nbits, 0, nullptr, codes, nbits / 8};
    return launch_gemm<1>(g, ps_stream(stream));
}
