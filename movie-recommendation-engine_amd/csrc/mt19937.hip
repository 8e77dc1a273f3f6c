// mt19937.hip -- numpy's legacy global RNG stream, generated on the device.
//
// The reference draws one `random_sample()` double from the process-global MT19937 per taken walk step
// (np.random.choice at utils/random_walk.py:79).  ps_mt19937_random_sample reproduces n such doubles from a
// given state (key[624], pos), optionally after skipping `skip` doubles (item shards of a multi-GPU job), and
// returns the advanced state, so the host can `np.random.set_state` afterwards and every later consumer of
// np.random sees the stream the reference would have left behind.
//   word stream : MT19937 recurrence x_{n+624} = x_{n+397} ^ twist(x_n, x_{n+1}), then tempering
//   double      : genrand_res53: a = w0 >> 5, b = w1 >> 6, (a * 2^26 + b) / 2^53
// The recurrence is serial (parallelism 227 per 624 words), so the stream is cut into 2^17-word chunks that
// independent workgroups generate; the 624-word window at the start of every chunk comes from GF(2) jump-ahead:
// with g(t) = t^(2^m) mod phi(t) (pinsage_hip/mtjump.py), the window 2^m words ahead is the XOR of the windows
// at the offsets i with g_i = 1 -- one workgroup expands 34 blocks of the sequence to global memory, then 24
// workgroups each XOR the windows selected by 1/24 of the polynomial (the 24 partial windows are XORed by whoever
// reads the window next).  Chunk windows are produced by doubling (1 -> 2 -> 4 ... windows per round), by radix-32 rounds
// (31 multiplier polynomials per round), or -- more than 32 chunks, the walk sampler's case -- by two radix-32 rounds whose
// window products run on the matrix cores ("jump products on the matrix cores" below).
// Without polynomials (or for short requests) a single workgroup generates the stream serially.
#include "ps_common.h"

#ifndef PS_MT_DEBUG
#define PS_MT_DEBUG 0     // knock-outs (tools/mt_knockout.sh; wrong results): jump product 1 no nibble-stream build, 2 A operands
#endif                    // loaded once, 4 no parity epilogue, 8 no MFMA, 64 Hankel fragments loaded once; chunk generator 16 no stores, 32 no barrier

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int DEG = 19937;
constexpr int CHUNK_LOG2 = 17;                          // 2^16: more jumps than chunk time saved (1.09 vs 0.94 ms for 11.8 M doubles)
constexpr int64_t CHUNK = (int64_t)1 << CHUNK_LOG2;     // words per chunk

// The chunk windows a request needs, as up to four runs of window indices (window w starts at stream word 1 + w * CHUNK): slot 0
// is always window 0 (free: it comes from the state itself), slot 1 + q is window win_of(q).  A whole-stream request is one run
// 1 .. K - 1; a rank of a sharded job asks only for the words of ITS start nodes (ps_mt19937_raw_stream `ranges`).
struct WinSet { int n; long long first[4]; long long cum[5]; };
struct Wanted { int n; long long lo[4], hi[4]; };             // wanted stream words: union of [lo, hi)
// (literal indices only: a run-time index into a by-value kernel argument sends the whole struct to scratch memory -- the chunk
// kernel took 149 us instead of 40 with loops over wt.n)
__host__ __device__ __forceinline__ long long win_of(const WinSet &ws, long long q) {
    long long w = ws.first[0] + q;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < ws.n && q >= ws.cum[i]) w = ws.first[i] + (q - ws.cum[i]);
    return w;
}

__device__ __forceinline__ uint32_t twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// next 624 words from the previous 624 (o -> n), all threads of the block, ONE barrier: thread t produces n[t],
// n[t + 227] and n[t + 454] in turn -- n[i] = n[i - 227] ^ twist(o[i], o[i + 1]) for i >= 227 needs only the thread's
// own previous result and words of the OLD block; the single exception n[623] = n[396] ^ twist(o[623], n[0]) takes n[0]
// from old words too (thread 169 recomputes it).  The first version synchronised after each of the three phases:
// 161 us per 2^17-word chunk instead of ~70.
__device__ __forceinline__ void next_block(const uint32_t *o, uint32_t *n, int t) {
    if (t < 227) {
        const uint32_t a = o[t + MT_M] ^ twist(o[t], o[t + 1]);
        const uint32_t b = a ^ twist(o[227 + t], o[228 + t]);
        n[t] = a;
        n[227 + t] = b;
        if (t < 169) n[454 + t] = b ^ twist(o[454 + t], o[455 + t]);
        else if (t == 169) n[623] = b ^ twist(o[623], o[MT_M] ^ twist(o[0], o[1]));
    }
    __syncthreads();
}

// the same update with every new word also handed to emit(i, v) straight from the producing thread's registers: consumers
// that copy each block to memory (sequence expansion, chunk generation) otherwise re-read the finished block from LDS
// after the barrier -- a second LDS round trip in a serial chain of 34 / 210 blocks (chunks: 1500 -> 1100 cycles per block)
template <typename Emit>
__device__ __forceinline__ void next_block_emit(const uint32_t *o, uint32_t *n, int t, Emit emit) {
    if (t < 227) {
        // every operand is requested up front (ONE LDS round trip per block; thread 169's four extra words are broadcast reads
        // for everybody, threads 170..226 re-read a valid pair they do not use)
        const int t3 = t < 169 ? 454 + t : 622;
        const uint32_t x0 = o[t], x1 = o[t + 1], xm = o[t + MT_M], y0 = o[227 + t], y1 = o[228 + t], z0 = o[t3], z1 = o[t3 + 1];
        const uint32_t l623 = o[623], lm = o[MT_M], l0 = o[0], l1 = o[1];
        const uint32_t a = xm ^ twist(x0, x1);
        const uint32_t b = a ^ twist(y0, y1);
        const uint32_t c = b ^ (t < 169 ? twist(z0, z1) : twist(l623, lm ^ twist(l0, l1)));
        n[t] = a;
        n[227 + t] = b;
        emit(t, a);
        emit(227 + t, b);
        if (t <= 169) {
            const int i3 = t < 169 ? 454 + t : 623;
            n[i3] = c;
            emit(i3, c);
        }
    }
    if (!(PS_MT_DEBUG & 32)) __syncthreads();
}

// The recurrence run BACKWARDS: x[k+624] ^ x[k+397] = twist(x[k], x[k+1]) is invertible -- the top bit of the left side says
// whether the magic constant went in (= low bit of x[k+1]) -- and yields y(k) = (top bit of x[k]) | (low 31 bits of x[k+1]), so
// x[k] = top(y(k)) | low(y(k-1)).
__device__ __forceinline__ uint32_t untwist(uint32_t tmp) {
    const uint32_t m = (uint32_t)((int32_t)tmp >> 31);                       // all ones when the magic constant went in
    return __builtin_amdgcn_alignbit(tmp ^ (m & 0x9908b0dfu), tmp, 31);      // ((..) << 1) | (tmp >> 31): four instructions
}

// previous 624 words from a block (o = x[n .. n+623] -> p = x[n-624 .. n-1]), ONE barrier and NO dependency inside the block:
// with Y(i) = y(n - 624 + i) = untwist(o[i] ^ (i >= 227 ? o[i - 227] : p[i + 397])) and p[i] = top(Y(i)) | low(Y(i - 1)),
//   p[i], i >= 228   needs o[i], o[i-1], o[i-227], o[i-228] only;
//   p[i], i = 1..226 needs p[i+397], p[i+396] -- both of the first kind -- and o[i], o[i-1];
//   p[227] and p[0] need p[623] (Y(226) = untwist(o[226] ^ p[623]), Y(-1) = untwist(p[623] ^ p[396])), of the first kind too.
// Thread t < 227 produces p[t + 397] and p[t] (recomputing p[t + 396] for itself: 9 old words), thread t < 170 also p[227 + t];
// thread 0's two extra words are broadcast reads for everybody.  (Measured and not kept: three untwists per thread with the
// neighbours' Y values taken by DPP wave_shr:1, waves overlapping by two lanes -- bit-identical, fewer instructions, but the chunk
// launch took 54.2 instead of 50.3 us and the begin kernel 9.7 instead of 8.8 in the same process: the wave-wide shifts sit in
// the block's dependent chain.)
template <typename Emit>
__device__ __forceinline__ void prev_block_emit(const uint32_t *o, uint32_t *p, int t, Emit emit) {
    constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu;
    if (t < 227) {
        const int tm = t > 0 ? t - 1 : 0, u = t < 170 ? t : 169;               // threads 170..226 re-read a valid pair they do not use
        const uint32_t a0 = o[t + 395], a1 = o[t + 396], a2 = o[t + 397], b0 = o[t + 168], b1 = o[t + 169], b2 = o[t + 170];
        const uint32_t c0 = o[tm], c1 = o[t], d0 = o[226 + u], d1 = o[227 + u], l622 = o[622], l623 = o[623];
        const uint32_t y397 = untwist(a2 ^ b2), y396 = untwist(a1 ^ b1), y395 = untwist(a0 ^ b0);
        const uint32_t n397 = (y397 & UP) | (y396 & LO);                       // p[t + 397]
        const uint32_t n396 = (y396 & UP) | (y395 & LO);                       // p[t + 396]
        uint32_t n623 = 0;
        if (t == 0) n623 = (untwist(l623 ^ a1) & UP) | (untwist(l622 ^ a0) & LO);        // p[623] (a1 = o[396], a0 = o[395]); three of four waves skip it
        const uint32_t ylo = untwist((t > 0 ? c0 : n623) ^ n396);              // Y(t - 1)
        const uint32_t nt = (untwist(c1 ^ n397) & UP) | (ylo & LO);            // p[t]
        p[t] = nt;
        p[t + 397] = n397;
        emit(t, nt);
        emit(t + 397, n397);
        if (t < 170) {
            const uint32_t v = (untwist(d1 ^ c1) & UP) | (untwist(d0 ^ (t > 0 ? c0 : n623)) & LO);    // p[227 + t]
            p[227 + t] = v;
            emit(227 + t, v);
        }
    }
    if (!(PS_MT_DEBUG & 32)) __syncthreads();
}

// ---- serial path: one workgroup, raw (tempered) words written in place of the doubles ----------------------
__global__ __launch_bounds__(256) void mt_words_kernel(const uint32_t *state_in, int pos_in, int64_t nwords,
                                                       uint32_t *raw, uint32_t *state_out, int32_t *pos_out) {
    __shared__ uint32_t mt[2][MT_N];
    const int t = threadIdx.x;
    for (int i = t; i < MT_N; i += 256) mt[0][i] = state_in[i];
    __syncthreads();
    int cur = 0, pos = pos_in;
    int64_t done = 0;
    while (done < nwords) {
        if (pos >= MT_N) {
            next_block(mt[cur], mt[cur ^ 1], t);
            cur ^= 1;
            pos = 0;
        }
        const int64_t rem = nwords - done;
        const int take = (MT_N - pos) < rem ? (MT_N - pos) : (int)rem;
        for (int i = t; i < take; i += 256) raw[done + i] = temper(mt[cur][pos + i]);
        pos += take;
        done += take;
    }
    __syncthreads();
    for (int i = t; i < MT_N; i += 256) state_out[i] = mt[cur][i];
    if (t == 0) pos_out[0] = pos;
}

__global__ void mt_pairs_to_double_kernel(double *out, int64_t n) {      // serial path: in place
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 w = reinterpret_cast<const uint2 *>(out)[i];
        out[i] = ((double)(w.x >> 5) * 67108864.0 + (double)(w.y >> 6)) * (1.0 / 9007199254740992.0);
    }
}

// ---- parallel path --------------------------------------------------------------------------------------
// Stream word w = 0 is the next word numpy would output (key[pos], or the first word after a twist when
// pos == 624).  W1 = the 624 raw words at stream indices 1..624 is the base of all jumps (word 0's low bits are
// not part of the 19937-bit state when it has never been produced by the recurrence).
__global__ __launch_bounds__(256) void mt_prepare_kernel(const uint32_t *state_in, int pos_in, uint32_t *w1,
                                                         uint32_t *word0) {
    __shared__ uint32_t seq[3 * MT_N];
    const int t = threadIdx.x;
    for (int i = t; i < MT_N; i += 256) seq[i] = state_in[i];
    __syncthreads();
    next_block(seq, seq + MT_N, t);
    next_block(seq + MT_N, seq + 2 * MT_N, t);
    for (int i = t; i < MT_N; i += 256) w1[i] = seq[pos_in + 1 + i];
    if (t == 0) word0[0] = seq[pos_in];
}

// A window is stored as JP partial vectors whose XOR is the 624-word window (a plain window = itself + zeros).
constexpr int JP = 24;                                   // workgroups per jump
constexpr int PW = MT_N / JP;                            // polynomial words per part (78 -> 2496 bits)
constexpr int SEQ_PAD = 34 * MT_N;                       // expanded sequence per source, whole blocks
constexpr int XB = 17;                                   // one-round scheme: blocks of it that lie BEFORE the source window
static_assert(SEQ_PAD >= DEG + MT_N && JP * PW == MT_N, "jump geometry");

__device__ __forceinline__ uint32_t load_window_word(const uint32_t *parts, int nparts, int t) {
    uint32_t v = 0;
    for (int p = 0; p < nparts; ++p) v ^= parts[p * MT_N + t];
    return v;
}

// seq[b] = the 34 * 624 words that start with window b (sequential expansion, one workgroup per source)
// (source window of block b = src + b * src_stride words, stored as nparts partial vectors: JP, or 1 for a plain window)
__global__ __launch_bounds__(256) void mt_expand_kernel(const uint32_t *src, int nparts, int64_t src_stride, uint32_t *seq) {
    __shared__ uint32_t mt[2][MT_N];
    const int t = threadIdx.x;
    uint32_t *out = seq + (size_t)blockIdx.x * SEQ_PAD;
    for (int i = t; i < MT_N; i += 256) {
        const uint32_t v = load_window_word(src + (size_t)blockIdx.x * src_stride, nparts, i);
        mt[0][i] = v;
        out[i] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int blk = 1; blk < 34; ++blk) {
        uint32_t *dst = out + blk * MT_N;
        next_block_emit(mt[cur], mt[cur ^ 1], t, [&](int i, uint32_t v) { dst[i] = v; });
        cur ^= 1;
    }
}

// dst_parts[b][p] = XOR over the set bits i of poly words [p*PW, (p+1)*PW) of seq[b][i .. i+624): part p of the
// window 2^m words after source b (poly = coefficients of t^(2^m) mod phi, 624 words, LSB first).
// The set bits are wave-uniform, so the bit scan is scalar work that every wave repeats: each lane therefore owns FOUR
// output words (t, t+156, t+312, t+468 -> two ds_read2_b32 per set bit) and a part is only 832 polynomial bits, which
// keeps a block's serial chain short while few windows exist (first doubling rounds) -- 8 parts x 624 single-word lanes
// took 25 us per block and 2x the instruction issue.
constexpr int CT = MT_N / 4;                             // 156 active lanes per block
__device__ __forceinline__ void combine_part(const uint32_t *s, const uint32_t *poly_part, uint32_t *dst) {
    __shared__ uint32_t sq[PW * 32 + MT_N + 8];
    __shared__ uint32_t pl[PW];
    const int t = threadIdx.x;
    for (int i = t; i < PW * 32 + MT_N; i += 192) sq[i] = s[i];
    if (t < PW) pl[t] = poly_part[t];
    __syncthreads();
    if (t < CT) {
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int w = 0; w < PW; ++w) {
            uint32_t bits = pl[w];
            const uint32_t *p = sq + w * 32 + t;
            while (bits) {
                uint32_t v[2][4];                 // two set bits per iteration (eight measured slower: 0.97 vs 0.80 ms)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool on = bits != 0u;
                    const int bu = on ? __builtin_ctz(bits) : 0;
                    v[u][0] = on ? p[bu] : 0u;
                    v[u][1] = on ? p[bu + CT] : 0u;
                    v[u][2] = on ? p[bu + 2 * CT] : 0u;
                    v[u][3] = on ? p[bu + 3 * CT] : 0u;
                    bits &= bits - 1u;            // 0 stays 0
                }
                a0 ^= v[0][0] ^ v[1][0];
                a1 ^= v[0][1] ^ v[1][1];
                a2 ^= v[0][2] ^ v[1][2];
                a3 ^= v[0][3] ^ v[1][3];
            }
        }
        dst[t] = a0;
        dst[t + CT] = a1;
        dst[t + 2 * CT] = a2;
        dst[t + 3 * CT] = a3;
    }
}

__global__ __launch_bounds__(192) void mt_combine_kernel(const uint32_t *seq, const uint32_t *poly, uint32_t *dst_parts) {
    const int b = blockIdx.x / JP, part = blockIdx.x % JP;
    combine_part(seq + (size_t)b * SEQ_PAD + part * PW * 32, poly + part * PW, dst_parts + ((size_t)b * JP + part) * MT_N);
}

// one radix-32 round: window j * have + b = jump_{j * have chunks}(window b), j = blockIdx.y + 1, for every target < K;
// polys[j - 1] = t^(j * have * CHUNK) mod phi
__global__ __launch_bounds__(192) void mt_combine_radix_kernel(const uint32_t *seq, const uint32_t *polys, uint32_t *states,
                                                               int64_t have, int64_t K) {
    const int b = blockIdx.x / JP, part = blockIdx.x % JP;
    const int64_t tgt = (int64_t)(blockIdx.y + 1) * have + b;
    if (tgt >= K) return;                                                  // block-uniform
    combine_part(seq + (size_t)b * SEQ_PAD + part * PW * 32, polys + (size_t)blockIdx.y * MT_N + part * PW,
                 states + ((size_t)tgt * JP + part) * MT_N);
}

// ---- jump products on the matrix cores -----------------------------------------------------------------------------
// A radix round applies up to 31 jump polynomials g_m to the same source window:
//     out_m[j] = XOR_{i : g_m[i] = 1} seq[i + j],   j < 624
// (seq = the source's expanded sequence).  Bit b of out_m[j] is the parity of  sum_i g_m[i] * plane_b[i + j]  with plane_b[p] =
// bit b of seq[p]: per bit plane a (32 polynomials) x (19 937) x (624 lags) matrix product with a HANKEL right operand.
// Both operands are 0 / 1, stored as fp4 (1.0 = 0x2) and contracted with v_mfma_scale_f32_32x32x64_f8f6f4 (block scales
// 2^0): every partial sum is an integer below 2^24, exact in the f32 accumulator; the parity is its lowest bit.
// (mt_combine_radix_kernel does the same products with one ds_read per set polynomial bit and output word: 8.8 GB of LDS
// reads for the 340 windows of a two-layer walk pass, 0.21 ms; this form: 2.4 M MFMAs, ~0.06 ms.)
//
// Operand A (polynomials; row m = lane & 31, k half = lane >> 5) comes pre-expanded from global memory
// (mt_pack_polys_kernel, 1 KiB per 64-bit k step).  Operand B for (lag tile J, k step ks) is the Hankel fragment
//     lane (n, kb):  plane_b[32 (2 ks + kb + J) + n + t],  t = 0..31   =: R_d, d = 2 ks + J
// which depends on 2 ks + J only: a register window of five R_d serves the five lag tiles of a wave and slides by two per
// step, so a step is 5 MFMAs for 2 fragment loads.  A fragment is 32 consecutive nibbles of the plane's nibble stream
// starting at nibble 32 (d + kb) + n: the workgroup keeps its slice of that stream in LDS in EIGHT copies, copy c shifted
// by c nibbles, so that every lane reads four whole dwords (copy n & 7).
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

constexpr int KS_TOTAL = 320;                 // 64-bit k steps: 312 (19 968 polynomial bits) padded so that 4 .. 32 slices have a multiple of 5 steps
constexpr int PLW = SEQ_PAD / 32;             // dwords per bit plane of an expanded sequence
constexpr int JT = 20, JG = 10;               // 32-lag tiles (624 lags = 19.5 tiles), lag tiles per wave
constexpr int MF_PLANES = 2;                  // bit planes per workgroup (two waves each)
static_assert(SEQ_PAD % 32 == 0 && JT * 32 >= MT_N && JT == 2 * JG && 32 % MF_PLANES == 0, "plane geometry: two waves x JG lag tiles");

// 8 bits -> 8 fp4 nibbles (bit set -> 1.0 = 0x2), lowest bit in the lowest nibble
__device__ __forceinline__ uint32_t bits_to_fp4(uint32_t byte) {
    auto four = [](uint32_t n) { return (n | (n << 3) | (n << 6) | (n << 9)) & 0x1111u; };
    return (four(byte & 15u) | (four((byte >> 4) & 15u) << 16)) << 1;
}

// polyA[ks][lane] = the 32 coefficients 64 ks + 32 (lane >> 5) + t of polynomial lane & 31 (rows >= `rows`: zero)
__global__ __launch_bounds__(256) void mt_pack_polys_kernel(const uint32_t *__restrict__ polys, int rows, uint4 *__restrict__ polyA) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= KS_TOTAL * 64) return;
    const int ks = idx >> 6, lane = idx & 63, m = lane & 31, w = 2 * ks + (lane >> 5);
    const uint32_t word = (m < rows && w < MT_N) ? polys[(size_t)m * MT_N + w] : 0u;
    polyA[idx] = make_uint4(bits_to_fp4(word & 255u), bits_to_fp4((word >> 8) & 255u), bits_to_fp4((word >> 16) & 255u),
                            bits_to_fp4(word >> 24));
}

// First launch of the one-round scheme.  Workgroups 0 and 1: the window W1 at stream word 1 and word 0 from numpy's state
// (= mt_prepare_kernel; without a state the window is read from plain0), W1 to plain0, and the 34 blocks of the sequence AROUND it
// (XB blocks before, 33 - XB after) that the window polynomials are applied to; the other workgroups meanwhile expand `ngroups` 32-row groups of the window-polynomial table into MFMA A
// operands (= mt_pack_polys_kernel per group; product q applies table row win_of(q) - 1; q >= nrows: zero).
__global__ __launch_bounds__(256) void mt_begin_kernel(const uint32_t *__restrict__ state_in, int pos_in, uint32_t *__restrict__ plain0,
                                                       uint32_t *__restrict__ word0, uint32_t *__restrict__ seq,
                                                       const uint32_t *__restrict__ polys, int nrows, int ngroups, uint4 *__restrict__ polyA,
                                                       WinSet ws) {
    const int t = threadIdx.x;
    if (blockIdx.x < 2) {
        // workgroup 0 expands forwards from the window (blocks XB + 1 .. 33 of seq), workgroup 1 backwards (blocks XB - 1 .. 0):
        // two chains of 16 / 17 block updates instead of one of 33; the window-polynomial table carries the offset XB * 624
        const bool back = blockIdx.x == 1;
        __shared__ uint32_t s3[3 * MT_N];
        __shared__ uint32_t mt[2][MT_N];
        if (state_in != nullptr) {
            for (int i = t; i < MT_N; i += 256) s3[i] = state_in[i];
            __syncthreads();
            next_block(s3, s3 + MT_N, t);
            next_block(s3 + MT_N, s3 + 2 * MT_N, t);
        }
        for (int i = t; i < MT_N; i += 256) {
            const uint32_t v = state_in != nullptr ? s3[pos_in + 1 + i] : plain0[i];     // (no state: the window is already in plain0)
            mt[0][i] = v;
            if (!back) {
                if (state_in != nullptr) plain0[i] = v;
                seq[XB * MT_N + i] = v;
            }
        }
        if (t == 0 && !back && state_in != nullptr) word0[0] = s3[pos_in];
        __syncthreads();
        int cur = 0;
        if (!back) {
            for (int blk = XB + 1; blk < 34; ++blk) {
                uint32_t *dst = seq + blk * MT_N;
                next_block_emit(mt[cur], mt[cur ^ 1], t, [&](int i, uint32_t v) { dst[i] = v; });
                cur ^= 1;
            }
        } else {
            for (int blk = XB - 1; blk >= 0; --blk) {
                uint32_t *dst = seq + blk * MT_N;
                prev_block_emit(mt[cur], mt[cur ^ 1], t, [&](int i, uint32_t v) { dst[i] = v; });
                cur ^= 1;
            }
        }
        return;
    }
    const int64_t idx = (int64_t)(blockIdx.x - 2) * 256 + t;
    if (idx >= (int64_t)ngroups * KS_TOTAL * 64) return;
    const int g = (int)(idx / (KS_TOTAL * 64)), rem = (int)(idx % (KS_TOTAL * 64));
    const int ks = rem >> 6, lane = rem & 63, q = 32 * g + (lane & 31), w = 2 * ks + (lane >> 5);
    const uint32_t word = (q < nrows && w < MT_N) ? polys[(size_t)(win_of(ws, q) - 1) * MT_N + w] : 0u;    // product q: window win_of(q) = table row - 1
    polyA[idx] = make_uint4(bits_to_fp4(word & 255u), bits_to_fp4((word >> 8) & 255u), bits_to_fp4((word >> 16) & 255u),
                            bits_to_fp4(word >> 24));
}

// planes[src][b][g] = bit b of the 32 words seq[src][32 g .. 32 g + 31] (word i -> bit i)
__global__ __launch_bounds__(256) void mt_planes_kernel(const uint32_t *__restrict__ seq, uint32_t *__restrict__ planes) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t *s = seq + (size_t)blockIdx.y * SEQ_PAD;
    uint32_t *pl = planes + (size_t)blockIdx.y * 32 * PLW;
    for (int grp = blockIdx.x * 4 + wv; grp * 64 < SEQ_PAD; grp += gridDim.x * 4) {
        const int i = grp * 64 + lane;
        const uint32_t word = i < SEQ_PAD ? s[i] : 0u;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const uint64_t m = __ballot((word >> b) & 1u);
            if (lane == 0) {
                pl[b * PLW + 2 * grp] = (uint32_t)m;
                if (2 * grp + 1 < PLW) pl[b * PLW + 2 * grp + 1] = (uint32_t)(m >> 32);
            }
        }
    }
}

__device__ __forceinline__ v16f_t bit_mfma(const v4i_t &a, const v4i_t &b, const v16f_t &c) {
    const v8i_t A = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
    const v8i_t B = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// One workgroup = (source, TWO bit planes, slice `part` of the polynomial bits); two waves per plane take ten lag tiles each:
// a k step is 10 MFMAs for one A fragment (global, shared by the workgroup's waves through the L1) and two Hankel fragments
// (LDS).  r03: five tiles per wave / four waves per plane asked for 51 B/clk of L1 and 102 B/clk of LDS per CU at the MFMA peak
// -- 80 % of either path -- and ran at 49 % of the peak; ten tiles halve both.
// PLp[part][src][b][J][lane] = the parities of this slice as the MFMA leaves them: bit r of entry `lane` = row
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5), lag 32 J + (lane & 31) (plain stores: mt_jump_reduce_kernel XORs the slices,
// mt_jump_finish_kernel turns the bit planes into words).
// `ngroups` > 1 (one-round windows): "source" src is the pair (real source src / ngroups, polynomial group src % ngroups) -- the
// groups are 32-row blocks of one long polynomial table, all applied to the same planes.
__global__ __launch_bounds__(256, 2) void mt_jump_mfma_kernel(const uint32_t *__restrict__ planes, const uint4 *__restrict__ polyA,
                                                              uint16_t *__restrict__ PLp, int nsrc, int parts, int steps, int ngroups) {
    extern __shared__ uint32_t copies[];                     // [MF_PLANES][8][CW]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pb = wv >> 1, g = wv & 1, n = lane & 31, kb = lane >> 5;
    int item = blockIdx.x;
    const int b0 = (item % (32 / MF_PLANES)) * MF_PLANES; item /= (32 / MF_PLANES);
    const int src = item % nsrc, part = item / nsrc;
    const int ks0 = part * steps, J0 = g * JG;
    const int dmin = 2 * ks0;
    const int CW = 4 * (2 * steps + JT + 1) + 8;             // dwords per copy: fragments d = dmin .. dmin + 2 steps + JT
    // nibble streams of the planes b0 .. from nibble 32 dmin on, eight shifts each
    for (int mm = tid; mm < ((PS_MT_DEBUG & 1) ? 64 : MF_PLANES * CW); mm += 256) {
        const int q = mm / CW, m = mm % CW;
        const uint32_t *pl = planes + ((size_t)(src / ngroups) * 32 + b0 + q) * PLW;
        uint32_t *cp = copies + (size_t)q * 8 * CW;
        const int p0 = 32 * dmin + 8 * m;                    // first plane bit of dword m (copy 0)
        const uint32_t w0 = (p0 >> 5) < PLW ? pl[p0 >> 5] : 0u, w1 = ((p0 + 8) >> 5) < PLW ? pl[(p0 + 8) >> 5] : 0u;
        const uint32_t e0 = bits_to_fp4((w0 >> (p0 & 31)) & 255u), e1 = bits_to_fp4((w1 >> ((p0 + 8) & 31)) & 255u);
        cp[m] = e0;
#pragma unroll
        for (int c = 1; c < 8; ++c) cp[c * CW + m] = __builtin_amdgcn_alignbit(e1, e0, 4 * c);
    }
    __syncthreads();
    const int b = b0 + pb;
    const uint32_t *frag = copies + (size_t)pb * 8 * CW + (n & 7) * CW + 4 * (J0 + kb) + (n >> 3);       // + 4 x for R_{dmin + J0 + x}
    auto load_r = [&](int x) { const uint32_t *q = frag + 4 * x; return v4i_t{(int)q[0], (int)q[1], (int)q[2], (int)q[3]}; };
    const uint4 *pa = polyA + ((size_t)(src % ngroups) * KS_TOTAL + ks0) * 64 + lane;
    // A fragments are requested PF steps ahead by inline asm and awaited with COUNTED vmcnt (the loop holds no other vector-memory
    // instruction): the compiler's own bookkeeping put `s_waitcnt vmcnt(0)` at the loop head, i.e. waited for the load it had just
    // issued, once per PF steps
    auto load_a = [&](v4i_t &dst, int u) {
        const uint4 *ptr = pa + (size_t)u * 64;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
    };

    // accumulators start at 2^23: the sums stay integers below 2^24 (at most 19 937 ones), exact in fp32, and the mantissa of
    // 2^23 + s IS s -- the parity is bit 0 of the register, no conversion (the epilogue was 3 instructions per value)
    v16f_t acc[JG];
#pragma unroll
    for (int j = 0; j < JG; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 8388608.f;
    constexpr int PF = 5;                                    // A fragments in flight (k steps ahead; 2 PF must be a multiple of JG)
    v4i_t R[JG], A[PF];
#pragma unroll
    for (int j = 0; j < JG; ++j) R[j] = load_r(j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the prologue's loads are done: from here on the count is the A ring's
#pragma unroll
    for (int u = 0; u < PF; ++u) load_a(A[u], u < steps ? u : 0);
    // `steps` is a multiple of PF (host: 45 or 15) and the loop body has NO branch: a conditional A load became a phi with
    // register copies behind `s_waitcnt vmcnt(0)` in every step (r03: 64 us for the 361 windows of a walk pass, the load latency
    // exposed once per step) -- past the end the last fragment is loaded again and not used
    for (int ub = 0; ub < steps; ub += PF) {
#pragma unroll
        for (int uu = 0; uu < PF; ++uu) {
            // step u = ub + uu: lag tile jj uses R_{dmin + J0 + 2 u + jj} = slot (2 uu + jj) % JG (2 ub is a multiple of JG)
            static_assert(PF == 5, "vmcnt literal below = PF - 1");
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(A[uu]) : : "memory");       // PF - 1 younger requests may still be in flight
            const v4i_t a = A[uu];
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                if (!(PS_MT_DEBUG & 8)) acc[jj] = bit_mfma(a, R[(2 * uu + jj) % JG], acc[jj]);
                else acc[jj][0] += __builtin_bit_cast(float, a[0] ^ R[(2 * uu + jj) % JG][0]);
                if (jj == 1 && !(PS_MT_DEBUG & 64)) {    // slots 2 uu, 2 uu + 1 are free: the two fragments the next step adds (past the end: in range, unused)
                    R[(2 * uu) % JG] = load_r(2 * (ub + uu) + JG);
                    R[(2 * uu + 1) % JG] = load_r(2 * (ub + uu) + JG + 1);
                }
            }
            { const int nx = ub + PF + uu; load_a(A[uu], (nx < steps && !(PS_MT_DEBUG & 2)) ? nx : steps - 1); }
            __builtin_amdgcn_sched_barrier(0);               // the request stays HERE, PF steps ahead of its use
        }
    }
    // The ring's last (unused) requests are still in flight: the wait must OWN their destination registers ("+v"), or the
    // compiler -- for which A[] is dead after the last MFMA -- hands those VGPRs to the epilogue (store addresses!) and a late
    // load lands on top of them.  (r03: with a bare `s_waitcnt` here, a two-process run on one GPU, where loads take longer,
    // ended in "memory access fault: write to a read-only page" every few steps; single-process runs never showed it.)
    static_assert(PF == 5, "one operand per ring slot below");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]) : : "memory");
    // parities: C col = lane & 31 (lag), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (polynomial).  Every lane packs the
    // parities of its 16 rows into 16 bits and stores them as they lie (one coalesced 128-byte store per lag tile);
    // mt_jump_finish_kernel picks bit r of lane (lag, row half) -- 16 ballots per tile cost 6 x the instructions
    uint16_t *out = PLp + ((((size_t)part * nsrc + src) * 32 + b) * JT + J0) * 64 + lane;
#pragma unroll
    for (int jj = 0; jj < JG; ++jj) {
        uint32_t bits = 0;                                   // v_alignbit: bit 0 of each value enters at bit 31, sixteen of them end in 31..16
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float f = acc[jj][r];                      // (bit_cast of a vector ELEMENT reads element 0 in this hipcc: scalar first)
            bits = __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, f), bits, 1);
        }
        bits >>= 16;
        if (!(PS_MT_DEBUG & 4) || bits == 0x12345u) out[jj * 64] = (uint16_t)bits;
    }
}

// PLp[0][i] ^= PLp[1..parts-1][i] (i over one slice's [nsrc][32][JT][64] 16-bit entries, two per thread)
// (Folding this into mt_jump_finish_kernel -- every thread XORing its 32 entries over the 8 slices itself -- measured 42 us
// instead of 5 + 5: the finish kernel's 2-byte gathers are read by 16 threads each, eight slices of them thrash the L1.)
__global__ __launch_bounds__(256) void mt_jump_reduce_kernel(uint32_t *__restrict__ PLp, int64_t slice_dwords, int parts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= slice_dwords) return;
    uint32_t v = PLp[i];
#pragma unroll 8
    for (int part = 1; part < parts; ++part) v ^= PLp[part * slice_dwords + i];     // (unrolled: the loads of a batch in flight together)
    PLp[i] = v;
}

// plain[tgt][j]: bit b = parity of polynomial row m at lag j in plane b = bit r of PL[src][b][j >> 5][32 kb + (j & 31)] with
// m = (r & 3) + 8 (r >> 2) + 4 kb; tgt = src * src_step + (m + 1) * m_step
__global__ __launch_bounds__(256) void mt_jump_finish_kernel(const uint16_t *__restrict__ PL, int nsrc, int rows, int64_t src_step,
                                                             int64_t m_step, int64_t K, uint32_t *__restrict__ plain) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)nsrc * rows * MT_N) return;
    const int j = (int)(idx % MT_N), m = (int)((idx / MT_N) % rows), src = (int)(idx / ((int64_t)MT_N * rows));
    const int64_t tgt = src * src_step + (m + 1) * m_step;
    if (tgt >= K) return;
    const int r = (m & 3) | ((m >> 3) << 2), kb = (m >> 2) & 1;
    const uint16_t *p = PL + (((size_t)src * 32) * JT + (j >> 5)) * 64 + 32 * kb + (j & 31);
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 32; ++b) w |= (((uint32_t)p[(size_t)b * JT * 64] >> r) & 1u) << b;
    plain[tgt * MT_N + j] = w;
}

__global__ __launch_bounds__(256) void mt_fold_kernel(const uint32_t *parts, uint32_t *plain) {      // JP parts -> plain window
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= MT_N) return;
    uint32_t v = 0;
#pragma unroll
    for (int p = 0; p < JP; ++p) v ^= parts[p * MT_N + i];
    plain[i] = v;
}

// Window c = states[c] = stream words [1 + (c0 + c) * CHUNK, + 624).  Its chunk is generated in BOTH directions by two
// workgroups (blockIdx.y): forwards the words [wbase, wbase + HALF), backwards the words [wbase - HALF, wbase) -- a serial chain
// of 105 block updates each (LDS round trip + barrier) instead of 210 for the same number of windows (r02: 72 us per chain).
// raw[w - w_lo] for w in [w_lo, w_hi).  The loops are unrolled by two so that the two LDS images have constant addresses, a
// block's position against the wanted range is scalar work, and a chain ends where the wanted words end.
// (Several chunks per workgroup behind common barriers were slower: 109 / 142 us for 2 / 4 against 99.)
constexpr int64_t HALF = CHUNK / 2;
// The 624 words from key_w on are also the state numpy is left in: stored to state_out by whoever produces them (key_w < 1:
// none here, mt_final_state_kernel copies them).
// Slot blockIdx.x holds window (slot == 0 ? 0 : win_of(slot - 1)) + c0; only words of `wt` are stored.  RANGED = false: one
// wanted run (the whole-stream request): the hull [lo, hi) IS the wanted set (the general tests cost the 722-workgroup launch 6 us).
template <bool RANGED>
__global__ __launch_bounds__(256) void mt_chunk_kernel(const uint32_t *states, int nparts, int64_t c0, WinSet ws, Wanted wt, int64_t w_lo,
                                                       uint32_t *raw, int64_t key_w, uint32_t *state_out, int pos, int32_t *pos_out) {
    __shared__ uint32_t mt[2][MT_N];
    const int t = threadIdx.x;
    const bool back = blockIdx.y != 0;
    if (blockIdx.x == 0 && !back && t == 0 && key_w >= 1) pos_out[0] = pos;
    const int64_t wbase = 1 + (c0 + (blockIdx.x == 0 ? 0 : win_of(ws, blockIdx.x - 1))) * CHUNK;
    // wanted words of this workgroup: the part of wt inside its half chunk [h_lo, h_hi); [lo, hi) = the hull of that part
    int64_t h_lo = back ? wbase - HALF : wbase, h_hi = back ? wbase : wbase + HALF;
    if (h_lo < 1) h_lo = 1;
    int64_t lo = h_hi, hi = h_lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < wt.n) {
            const int64_t a = wt.lo[i] > h_lo ? wt.lo[i] : h_lo, b = wt.hi[i] < h_hi ? wt.hi[i] : h_hi;
            if (a < b) { lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
        }
    }
    if (lo >= hi) return;                                      // block-uniform
    auto wanted = [&](int64_t w) {
        if (!RANGED) return w >= lo && w < hi;
        bool in = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) in = in || (i < wt.n && w >= wt.lo[i] && w < wt.hi[i]);
        return in && w >= h_lo && w < h_hi;
    };
    auto put = [&](int64_t w0, int i, uint32_t v) {            // word i of the block that starts at stream word w0
        const int64_t w = w0 + i;
        if (wanted(w)) {
            raw[w - w_lo] = v;
            if (w >= key_w && w < key_w + MT_N) state_out[w - key_w] = v;
        }
    };
    for (int i = t; i < MT_N; i += 256) {
        const uint32_t v = load_window_word(states + (size_t)blockIdx.x * nparts * MT_N, nparts, i);
        mt[0][i] = v;
        if (!back) put(wbase, i, v);
    }
    __syncthreads();
    auto step = [&](const uint32_t *o, uint32_t *n, int64_t w0) {
        bool whole = !RANGED;                                  // whole block inside ONE wanted run and this half chunk, no state word in it (scalar test)
        if (RANGED) {
#pragma unroll
            for (int i = 0; i < 4; ++i) whole = whole || (i < wt.n && w0 >= wt.lo[i] && w0 + MT_N <= wt.hi[i]);
        }
        if (whole && w0 >= (RANGED ? h_lo : lo) && w0 + MT_N <= (RANGED ? h_hi : hi) && (w0 + MT_N <= key_w || w0 >= key_w + MT_N)) {   // no per-word tests
            uint32_t *dst = raw + (w0 - w_lo);
            if (back) prev_block_emit(o, n, t, [&](int i, uint32_t v) { if (!(PS_MT_DEBUG & 16) || v == 0x12345u) dst[i] = v; });
            else next_block_emit(o, n, t, [&](int i, uint32_t v) { if (!(PS_MT_DEBUG & 16) || v == 0x12345u) dst[i] = v; });
        } else {
            bool any = !RANGED;                                // a block on the way to the wanted words stores nothing (scalar test too:
            if (RANGED) {                                      // the per-word tests made such a chain 2.5 x slower than a stored one)
#pragma unroll
                for (int i = 0; i < 4; ++i) any = any || (i < wt.n && w0 < wt.hi[i] && w0 + MT_N > wt.lo[i]);
            }
            if (!any) {
                if (back) prev_block_emit(o, n, t, [&](int, uint32_t) {});
                else next_block_emit(o, n, t, [&](int, uint32_t) {});
            } else {
                if (back) prev_block_emit(o, n, t, [&](int i, uint32_t v) { put(w0, i, v); });
                else next_block_emit(o, n, t, [&](int i, uint32_t v) { put(w0, i, v); });
            }
        }
    };
    if (!back) {
        int64_t w0 = wbase + MT_N;
        for (; w0 + MT_N < hi; w0 += 2 * MT_N) {
            step(mt[0], mt[1], w0);
            step(mt[1], mt[0], w0 + MT_N);
        }
        if (w0 < hi) step(mt[0], mt[1], w0);
    } else {
        int64_t w0 = wbase - MT_N;                             // first word of the block being produced
        for (; w0 > lo; w0 -= 2 * MT_N) {                      // a second block is wanted after this one
            step(mt[0], mt[1], w0);
            step(mt[1], mt[0], w0 - MT_N);
        }
        if (w0 + MT_N > lo) step(mt[0], mt[1], w0);
    }
}

__global__ void mt_raw_to_double_kernel(const uint32_t *raw, int64_t w_lo, int64_t skip, int64_t n, double *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = 2 * (skip + i) - w_lo;
        const uint32_t a = temper(raw[w]), b = temper(raw[w + 1]);
        out[i] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
    }
}

__global__ void mt_set_pos_kernel(int pos, int32_t *pos_out) { pos_out[0] = pos; }

__global__ void mt_final_state_kernel(const uint32_t *raw, int64_t w_lo, int64_t key_w, int pos, uint32_t *state_out,
                                      int32_t *pos_out) {
    for (int i = threadIdx.x; i < MT_N; i += blockDim.x) state_out[i] = raw[key_w - w_lo + i];
    if (threadIdx.x == 0) pos_out[0] = pos;
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct Plan {
    int64_t wa, wb;          // needed output words [wa, wb)
    int64_t key_w;           // stream index of the final key's first word (may be < 0: key unchanged)
    int pos_out;
    int64_t w_lo, w_hi;      // raw words kept
    int64_t c0, c1;          // chunks (relative to W1) covering [max(w_lo, 1), w_hi)
};

Plan make_plan(int pos_in, int64_t skip, int64_t n) {
    Plan p;
    p.wa = 2 * skip;
    p.wb = 2 * (skip + n);
    const int64_t consumed = pos_in + p.wb;             // words from the start of the current block
    int64_t q = consumed / MT_N, r = consumed % MT_N;
    if (r == 0 && p.wb > 0) { q -= 1; r = MT_N; }
    p.pos_out = (int)r;
    p.key_w = q * MT_N - pos_in;
    p.w_lo = p.wa;
    if (p.key_w >= 0 && p.key_w < p.w_lo) p.w_lo = p.key_w;
    p.w_hi = p.wb;
    if (p.key_w >= 0 && p.key_w + MT_N > p.w_hi) p.w_hi = p.key_w + MT_N;
    // window c serves the words [1 + c * CHUNK - HALF, 1 + c * CHUNK + HALF) (mt_chunk_kernel: both directions)
    const int64_t first = p.w_lo < 1 ? 1 : p.w_lo, last = p.w_hi - 1;
    p.c0 = (first - 1 + HALF) / CHUNK;
    p.c1 = last >= 1 ? (last - 1 + HALF) / CHUNK : 0;
    if (p.c1 < p.c0) p.c1 = p.c0;
    return p;
}

}  // namespace

extern "C" int ps_mt19937_chunk_log2(void) { return CHUNK_LOG2; }
extern "C" int ps_mt19937_window_shift(void) { return XB * MT_N; }

static int mt_generate(const uint32_t *state_in, int pos_in, int64_t skip, int64_t n, double *out, uint32_t *raw_out,
                       uint32_t *state_out, int32_t *pos_out, const uint32_t *jump_polys, int jump_levels,
                       const uint32_t *radix_polys, int radix_levels, const uint32_t *window_polys, int n_window,
                       const int64_t *ranges_host, int n_ranges, void *workspace, size_t workspace_bytes, ps_stream_t stream);

extern "C" int ps_mt19937_random_sample(const uint32_t *state_in, int pos_in, int64_t skip, int64_t n, double *out,
                                        uint32_t *state_out, int32_t *pos_out, const uint32_t *jump_polys,
                                        int jump_levels, const uint32_t *radix_polys, int radix_levels,
                                        const uint32_t *window_polys, int n_window, void *workspace,
                                        size_t workspace_bytes, ps_stream_t stream) {
    return mt_generate(state_in, pos_in, skip, n, out, nullptr, state_out, pos_out, jump_polys, jump_levels, radix_polys,
                       radix_levels, window_polys, n_window, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int ps_mt19937_raw_stream(const uint32_t *state_in, int pos_in, int64_t n, uint32_t *raw, uint32_t *state_out,
                                     int32_t *pos_out, const uint32_t *jump_polys, int jump_levels,
                                     const uint32_t *radix_polys, int radix_levels, const uint32_t *window_polys,
                                     int n_window, const int64_t *ranges_host, int n_ranges, void *workspace,
                                     size_t workspace_bytes, ps_stream_t stream) {
    if (!raw || !jump_polys || !workspace || n < (1 << 17) || n_ranges < 0 || (n_ranges > 0 && !ranges_host)) return PS_EINVAL;
    return mt_generate(state_in, pos_in, 0, n, nullptr, raw, state_out, pos_out, jump_polys, jump_levels, radix_polys,
                       radix_levels, window_polys, n_window, ranges_host, n_ranges, workspace, workspace_bytes, stream);
}

extern "C" size_t ps_mt19937_workspace_bytes(int64_t skip, int64_t n) {
    if (n < 0 || skip < 0) return 0;
    const Plan p = make_plan(MT_N, skip, n);            // pos only shifts the plan by < 624 words
    int64_t K = p.c1 - p.c0 + 3;
    if (K < 128) K = 128;                               // room for the one-round products of short (ranged) requests
    const size_t states = align256((size_t)(K + 4) * JP * MT_N * 4);
    const size_t seqs = align256((size_t)(K / 2 + 2) * SEQ_PAD * 4);
    return states + seqs + align256((size_t)(p.w_hi - p.w_lo + 2 * MT_N + CHUNK) * 4) + 4096;
}

static int mt_generate(const uint32_t *state_in, int pos_in, int64_t skip, int64_t n, double *out, uint32_t *raw_out,
                       uint32_t *state_out, int32_t *pos_out, const uint32_t *jump_polys,
                       int jump_levels, const uint32_t *radix_polys, int radix_levels, const uint32_t *window_polys,
                       int n_window, const int64_t *ranges_host, int n_ranges, void *workspace, size_t workspace_bytes,
                       ps_stream_t stream) {
    if (!state_in || !state_out || !pos_out || n < 0 || skip < 0 || pos_in < 0 || pos_in > MT_N) return PS_EINVAL;
    if (n > 0 && !out && !raw_out) return PS_EINVAL;
    hipStream_t st = ps_stream(stream);
    if (n == 0 && skip == 0) {                          // nothing consumed: state unchanged
        if (hipMemcpyAsync(state_out, state_in, MT_N * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return PS_ELAUNCH;
        hipLaunchKernelGGL(mt_set_pos_kernel, dim3(1), dim3(1), 0, st, pos_in, pos_out);
        PS_CHECK_LAUNCH();
        return PS_OK;
    }
    const bool parallel = jump_polys != nullptr && workspace != nullptr && (skip > 0 || n >= (1 << 17));
    if (!parallel) {
        if (skip != 0) return PS_EUNSUPPORTED;          // skipping needs the jump polynomials
        hipLaunchKernelGGL(mt_words_kernel, dim3(1), dim3(256), 0, st, state_in, pos_in, 2 * n,
                           reinterpret_cast<uint32_t *>(out), state_out, pos_out);
        PS_CHECK_LAUNCH();
        if (n > 0) {
            int64_t grid = ps_cdiv(n, 256);
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL(mt_pairs_to_double_kernel, dim3((unsigned)grid), dim3(256), 0, st, out, n);
            PS_CHECK_LAUNCH();
        }
        return PS_OK;
    }
    const Plan p = make_plan(pos_in, skip, n);
    const int64_t K = p.c1 - p.c0 + 1;
    if (workspace_bytes < ps_mt19937_workspace_bytes(skip, n)) return PS_EWORKSPACE;
    if ((p.c1 >> (jump_levels - CHUNK_LOG2)) != 0) return PS_EUNSUPPORTED;   // offset beyond the polynomial table
    char *base = reinterpret_cast<char *>(align256(reinterpret_cast<size_t>(workspace)));
    const size_t WSZ = (size_t)JP * MT_N;                                    // words per stored window
    uint32_t *states = reinterpret_cast<uint32_t *>(base);                   // [K] chunk windows (JP parts each)
    const int64_t Kc = K > 126 ? K : 126;                                    // the regions are carved for at least 126 windows (workspace_bytes: 128)
    uint32_t *tmpA = states + (size_t)(Kc + 1) * WSZ;                        // ping-pong for the base jump
    uint32_t *tmpB = tmpA + WSZ;
    uint32_t *word0 = tmpB + WSZ;
    uint32_t *seqs = reinterpret_cast<uint32_t *>(base + align256((size_t)(Kc + 4) * JP * MT_N * 4));
    uint32_t *raw = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(seqs) + align256((size_t)(Kc / 2 + 2) * SEQ_PAD * 4));
    if (raw_out) {                                       // raw mode: the chunk generators write straight into the caller's buffer
        if (p.w_lo != 0) return PS_EUNSUPPORTED;         // (skip = 0: word 2i is the first word of uniform i)
        raw = raw_out;
    }
    // One-round windows (window_polys: row j - 1 = t^(j * CHUNK) mod phi, at least K - 1 rows): every chunk window is ONE jump
    // from the first one, all K - 1 products in a single round on the matrix cores -- launches: begin (window 0, its expansion,
    // A operands), planes, products, reduce, finish, chunks, instead of the 16 of the two-round scheme (r03: 184 -> ... us for
    // the 361 windows of a two-layer walk pass; the second round's source windows had to wait for a whole first round)
    // Wanted words and window slots.  Default: everything in [w_lo, w_hi), windows 0 .. K - 1.  `ranges` (raw mode; a rank of a
    // sharded job): only the words of the given runs of uniforms plus the 624 words of the state hand-back, and only the windows
    // whose half chunks hold such words -- products, chunk chains and stores shrink with the share of the stream a rank consumes
    // (words outside the runs are left unwritten).  Needs the one-round table; otherwise the whole stream is generated.
    WinSet ws;
    Wanted wt;
    wt.n = 1; wt.lo[0] = p.w_lo; wt.hi[0] = p.w_hi;
    ws.n = 1; ws.first[0] = 1; ws.cum[0] = 0; ws.cum[1] = K - 1;
    int64_t Kw = K;                                      // window slots (slot 0 = window 0)
    bool ranged = false;
    if (ranges_host && n_ranges > 0 && n_ranges <= 3 && raw_out && skip == 0 && p.c0 == 0 && window_polys) {
        long long a[4], b[4];
        int m = 0;
        for (int i = 0; i < n_ranges; ++i) {
            const long long lo = ranges_host[2 * i] < 0 ? 0 : ranges_host[2 * i], hi = ranges_host[2 * i + 1] > n ? n : ranges_host[2 * i + 1];
            if (lo < hi) { a[m] = 2 * lo; b[m] = 2 * hi; ++m; }
        }
        if (p.key_w >= 0) { a[m] = p.key_w; b[m] = p.key_w + MT_N; ++m; }
        for (int i = 1; i < m; ++i)                      // sort by start, then merge runs that touch
            for (int j = i; j > 0 && a[j] < a[j - 1]; --j) { long long t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; t = b[j]; b[j] = b[j - 1]; b[j - 1] = t; }
        int mm = 0;
        for (int i = 0; i < m; ++i) {
            if (mm > 0 && a[i] <= b[mm - 1]) { if (b[i] > b[mm - 1]) b[mm - 1] = b[i]; }
            else { a[mm] = a[i]; b[mm] = b[i]; ++mm; }
        }
        long long f[4], l[4], maxwin = 0;
        int nw = 0;
        for (int i = 0; i < mm; ++i) {                   // window c serves the words [1 + c * CHUNK - HALF, 1 + c * CHUNK + HALF)
            const long long a1 = a[i] < 1 ? 1 : a[i];
            if (b[i] <= a1) continue;
            long long clo = (a1 - 1 + HALF) / CHUNK;
            const long long chi = (b[i] - 2 + HALF) / CHUNK;
            if (clo == 0) clo = 1;                       // window 0 is slot 0
            if (chi < clo) continue;
            if (nw > 0 && clo <= l[nw - 1] + 1) { if (chi > l[nw - 1]) l[nw - 1] = chi; }
            else { f[nw] = clo; l[nw] = chi; ++nw; }
            maxwin = l[nw - 1];
        }
        if (mm > 0 && maxwin <= n_window) {
            ranged = true;
            wt.n = mm;
            for (int i = 0; i < mm; ++i) { wt.lo[i] = a[i]; wt.hi[i] = b[i]; }
            ws.n = nw > 0 ? nw : 1;
            ws.cum[0] = 0; ws.cum[1] = 0; ws.first[0] = 1;
            for (int i = 0; i < nw; ++i) { ws.first[i] = f[i]; ws.cum[i + 1] = ws.cum[i] + (l[i] - f[i] + 1); }
            Kw = 1 + ws.cum[ws.n];
        }
    }
    const int64_t nprod = Kw - 1;                        // windows that take a product
    bool plain_states = false;
    uint32_t *plainS = nullptr;
    const int ngroups = (int)((nprod + 31) / 32);
    bool one_round = window_polys != nullptr && (ranged || (K > 32 && K - 1 <= n_window));
    uint32_t *seqM = nullptr, *planes1 = nullptr;
    uint4 *polyAw = nullptr;
    uint16_t *PLw = reinterpret_cast<uint16_t *>(states + WSZ);
    // slices of the polynomial bits per (group, plane pair): the launch runs in rounds of 512 workgroups (two per CU), so pick the
    // slicing whose last round is fullest (361 windows: 12 x 16 x 8 = 1536 = three whole rounds; 7 slices were 2.6 rounds = three)
    int parts_w = 32;
    {
        double best = 1e30;
        for (int parts = 4; parts <= 32; parts *= 2) {
            const int64_t wgs = (int64_t)ngroups * (32 / MF_PLANES) * parts;
            const double cost = (double)((wgs + 511) / 512) * (KS_TOTAL / parts + 4);      // rounds x (steps + a workgroup's fixed part)
            if (wgs >= 256 && cost < best) { best = cost; parts_w = parts; }
        }
    }
    if (one_round) {
        char *q = reinterpret_cast<char *>(seqs);
        const char *q_end = q + align256((size_t)(Kc / 2 + 2) * SEQ_PAD * 4);
        seqM = reinterpret_cast<uint32_t *>(q);       q += align256((size_t)SEQ_PAD * 4);
        planes1 = reinterpret_cast<uint32_t *>(q);    q += align256((size_t)32 * PLW * 4);
        polyAw = reinterpret_cast<uint4 *>(q);        q += align256((size_t)ngroups * KS_TOTAL * 64 * 16);
        plainS = reinterpret_cast<uint32_t *>(q);     q += align256((size_t)Kw * MT_N * 4);
        one_round = q <= q_end && (size_t)parts_w * ngroups * 32 * JT * 32 <= (size_t)(Kc - 1) * WSZ;
        if (!one_round && ranged) return PS_EWORKSPACE;  // (cannot happen with ps_mt19937_workspace_bytes: 126 windows' room at least)
    }
    const unsigned pack_blocks = (unsigned)(((int64_t)ngroups * KS_TOTAL * 64 + 255) / 256);
    if (one_round && p.c0 == 0) {
        hipLaunchKernelGGL(mt_begin_kernel, dim3(2 + pack_blocks), dim3(256), 0, st, state_in, pos_in, plainS, word0, seqM, window_polys,
                           (int)nprod, ngroups, polyAw, ws);
        PS_CHECK_LAUNCH();
    } else {
        // 1. W1 (part 0 of tmpA, the other parts zero) and word 0
        if (hipMemsetAsync(tmpA, 0, WSZ * 4, st) != hipSuccess) return PS_ELAUNCH;
        hipLaunchKernelGGL(mt_prepare_kernel, dim3(1), dim3(256), 0, st, state_in, pos_in, tmpA, word0);
        PS_CHECK_LAUNCH();
        // 2. base jump to chunk c0: offset c0 * CHUNK words = set bits of c0 at levels CHUNK_LOG2 + b
        uint32_t *cur = tmpA, *nxt = tmpB;
        for (int b = 0; (p.c0 >> b) != 0; ++b) {
            if (!((p.c0 >> b) & 1)) continue;
            hipLaunchKernelGGL(mt_expand_kernel, dim3(1), dim3(256), 0, st, cur, JP, (int64_t)WSZ, seqs);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_combine_kernel, dim3(JP), dim3(192), 0, st, seqs, jump_polys + (size_t)(CHUNK_LOG2 + b) * MT_N, nxt);
            PS_CHECK_LAUNCH();
            uint32_t *t = cur; cur = nxt; nxt = t;
        }
        if (hipMemcpyAsync(states, cur, WSZ * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return PS_ELAUNCH;
        if (one_round) {                                // after a skipped prefix: window c0 is in `states`, JP parts
            hipLaunchKernelGGL(mt_fold_kernel, dim3(3), dim3(256), 0, st, states, plainS);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_begin_kernel, dim3(2 + pack_blocks), dim3(256), 0, st, (const uint32_t *)nullptr, 0, plainS, word0, seqM,
                               window_polys, (int)nprod, ngroups, polyAw, ws);
            PS_CHECK_LAUNCH();
        }
    }
    if (one_round && nprod == 0) plain_states = true;    // a ranged request inside window 0's half chunk
    if (one_round && nprod > 0) {
        hipLaunchKernelGGL(mt_planes_kernel, dim3((SEQ_PAD / 64 + 4) / 4, 1), dim3(256), 0, st, seqM, planes1);
        PS_CHECK_LAUNCH();
        const int steps = KS_TOTAL / parts_w;
        const size_t lds = (size_t)MF_PLANES * 8 * (4 * (2 * steps + JT + 1) + 8) * 4;
        hipLaunchKernelGGL(mt_jump_mfma_kernel, dim3((unsigned)(ngroups * (32 / MF_PLANES) * parts_w)), dim3(256), lds, st, planes1, polyAw, PLw, ngroups,
                           parts_w, steps, ngroups);
        PS_CHECK_LAUNCH();
        const int64_t slice = (int64_t)ngroups * 32 * JT * 32;
        hipLaunchKernelGGL(mt_jump_reduce_kernel, dim3((unsigned)ps_cdiv(slice, 256)), dim3(256), 0, st, reinterpret_cast<uint32_t *>(PLw),
                           slice, parts_w);
        PS_CHECK_LAUNCH();
        hipLaunchKernelGGL(mt_jump_finish_kernel, dim3((unsigned)ps_cdiv((int64_t)ngroups * 32 * MT_N, 256)), dim3(256), 0, st, PLw, ngroups,
                           32, (int64_t)32, (int64_t)1, Kw, plainS);
        PS_CHECK_LAUNCH();
        plain_states = true;
    }
    // 3. chunk windows: radix-32 rounds (window j * have + r = jump_{j * have chunks}(window r), j = 1..31) when the
    //    multiplier polynomials are given -- two rounds instead of eight serial expansions for 180 chunks -- else doubling
    if (!plain_states && radix_polys && radix_levels >= 2 && K > 32 && K <= 1024) {
        // two radix-32 rounds on the matrix cores, big strides first: round A makes the windows 32 j (j = 1..JA) from window 0
        // (polynomials of level 1), round B the windows 32 a + j (j = 1..31) from the windows 32 a, a = 0..JA (level 0): every
        // product of round B shares its source with 30 others, which is what fills the 32 rows of the MFMA
        const int JA = (int)((K - 1) / 32), nsrcB = JA + 1;
        char *q = reinterpret_cast<char *>(seqs);
        const char *q_end = q + align256((size_t)(K / 2 + 2) * SEQ_PAD * 4);
        uint32_t *seqM = reinterpret_cast<uint32_t *>(q);      q += align256((size_t)nsrcB * SEQ_PAD * 4);
        uint32_t *planes = reinterpret_cast<uint32_t *>(q);    q += align256((size_t)nsrcB * 32 * PLW * 4);
        uint4 *polyA0 = reinterpret_cast<uint4 *>(q);          q += align256((size_t)KS_TOTAL * 64 * 16);
        uint4 *polyA1 = reinterpret_cast<uint4 *>(q);          q += align256((size_t)KS_TOTAL * 64 * 16);
        plainS = reinterpret_cast<uint32_t *>(q);              q += align256((size_t)K * MT_N * 4);
        // parity planes of every polynomial slice: in the JP-part window store, unused (but for window 0) in this mode
        constexpr int PARTS_A = 32, PARTS_B = 8;
        uint16_t *PLp = reinterpret_cast<uint16_t *>(states + WSZ);
        const bool pl_fits = (size_t)PARTS_B * nsrcB * 32 * JT * 32 <= (size_t)(K - 1) * WSZ && (size_t)PARTS_A * 32 * JT * 32 <= (size_t)(K - 1) * WSZ;
        if (q <= q_end && pl_fits) {
            const unsigned pk = (KS_TOTAL * 64 + 255) / 256;
            hipLaunchKernelGGL(mt_pack_polys_kernel, dim3(pk), dim3(256), 0, st, radix_polys + (size_t)1 * 31 * MT_N, JA, polyA1);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_pack_polys_kernel, dim3(pk), dim3(256), 0, st, radix_polys, 31, polyA0);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_fold_kernel, dim3(3), dim3(256), 0, st, states, plainS);
            PS_CHECK_LAUNCH();
            auto round = [&](int nsrc, const uint4 *polyA, int rows, int parts, int64_t src_step, int64_t m_step) -> int {
                // sources: the plain windows src * 32 (round A: window 0 only)
                hipLaunchKernelGGL(mt_expand_kernel, dim3((unsigned)nsrc), dim3(256), 0, st, plainS, 1, (int64_t)32 * MT_N, seqM);
                PS_CHECK_LAUNCH();
                hipLaunchKernelGGL(mt_planes_kernel, dim3((SEQ_PAD / 64 + 4) / 4, (unsigned)nsrc), dim3(256), 0, st, seqM, planes);
                PS_CHECK_LAUNCH();
                const int steps = KS_TOTAL / parts;
                const size_t lds = (size_t)MF_PLANES * 8 * (4 * (2 * steps + JT + 1) + 8) * 4;
                hipLaunchKernelGGL(mt_jump_mfma_kernel, dim3((unsigned)(nsrc * (32 / MF_PLANES) * parts)), dim3(256), lds, st, planes, polyA, PLp, nsrc,
                                   parts, steps, 1);
                PS_CHECK_LAUNCH();
                const int64_t total = (int64_t)nsrc * rows * MT_N;
                const int64_t slice = (int64_t)nsrc * 32 * JT * 32;                     // dwords (two 16-bit entries each)
                hipLaunchKernelGGL(mt_jump_reduce_kernel, dim3((unsigned)ps_cdiv(slice, 256)), dim3(256), 0, st,
                                   reinterpret_cast<uint32_t *>(PLp), slice, parts);
                PS_CHECK_LAUNCH();
                hipLaunchKernelGGL(mt_jump_finish_kernel, dim3((unsigned)ps_cdiv(total, 256)), dim3(256), 0, st, PLp, nsrc, rows, src_step,
                                   m_step, K, plainS);
                PS_CHECK_LAUNCH();
                return PS_OK;
            };
            int rc = round(1, polyA1, JA, PARTS_A, 0, 32);       // 672 short workgroups: the round is one item deep
            if (rc != PS_OK) return rc;
            rc = round(nsrcB, polyA0, 31, PARTS_B, 32, 1);
            if (rc != PS_OK) return rc;
            plain_states = true;
        }
    }
    if (plain_states) {
    } else if (radix_polys && radix_levels > 0) {
        int lvl = 0;
        for (int64_t have = 1; have < K; have *= 32, ++lvl) {
            if (lvl >= radix_levels) return PS_EUNSUPPORTED;
            const int64_t nsrc = (K - have) < have ? (K - have) : have;
            const int64_t jmax = ((K - 1) / have) < 31 ? ((K - 1) / have) : 31;
            hipLaunchKernelGGL(mt_expand_kernel, dim3((unsigned)nsrc), dim3(256), 0, st, states, JP, (int64_t)WSZ, seqs);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_combine_radix_kernel, dim3((unsigned)(nsrc * JP), (unsigned)jmax), dim3(192), 0, st, seqs,
                               radix_polys + (size_t)lvl * 31 * MT_N, states, have, K);
            PS_CHECK_LAUNCH();
        }
    } else {
        for (int m = 0; ((int64_t)1 << m) < K; ++m) {
            const int64_t have = (int64_t)1 << m;
            const int64_t make = (K - have) < have ? (K - have) : have;
            hipLaunchKernelGGL(mt_expand_kernel, dim3((unsigned)make), dim3(256), 0, st, states, JP, (int64_t)WSZ, seqs);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_combine_kernel, dim3((unsigned)(make * JP)), dim3(192), 0, st, seqs,
                               jump_polys + (size_t)(CHUNK_LOG2 + m) * MT_N, states + (size_t)have * WSZ);
            PS_CHECK_LAUNCH();
        }
    }
    // 4. chunks -> raw words; word 0 separately
    const int64_t key_fold = p.key_w >= 1 ? p.key_w : -(int64_t)4 * MT_N;         // no stream word lies in the folded range then
    if (ranged)
        hipLaunchKernelGGL(mt_chunk_kernel<true>, dim3((unsigned)Kw, 2), dim3(256), 0, st, plain_states ? plainS : states, plain_states ? 1 : JP,
                           p.c0, ws, wt, p.w_lo, raw, key_fold, state_out, p.pos_out, pos_out);
    else
        hipLaunchKernelGGL(mt_chunk_kernel<false>, dim3((unsigned)Kw, 2), dim3(256), 0, st, plain_states ? plainS : states, plain_states ? 1 : JP,
                           p.c0, ws, wt, p.w_lo, raw, key_fold, state_out, p.pos_out, pos_out);
    PS_CHECK_LAUNCH();
    if (p.w_lo == 0)
        if (hipMemcpyAsync(raw, word0, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return PS_ELAUNCH;
    // 5. doubles (raw mode: the consumer tempers and combines)
    if (n > 0 && !raw_out) {
        int64_t grid = ps_cdiv(n, 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(mt_raw_to_double_kernel, dim3((unsigned)grid), dim3(256), 0, st, raw, p.w_lo, skip, n, out);
        PS_CHECK_LAUNCH();
    }
    // 6. the state numpy would be left in
    if (p.key_w >= 1) {
        // stored by the chunk generators
    } else if (p.key_w == 0) {
        hipLaunchKernelGGL(mt_final_state_kernel, dim3(1), dim3(640), 0, st, raw, p.w_lo, p.key_w, p.pos_out, state_out, pos_out);
        PS_CHECK_LAUNCH();
    } else {
        if (hipMemcpyAsync(state_out, state_in, MT_N * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return PS_ELAUNCH;
        hipLaunchKernelGGL(mt_set_pos_kernel, dim3(1), dim3(1), 0, st, p.pos_out, pos_out);
        PS_CHECK_LAUNCH();
    }
    return PS_OK;
}
