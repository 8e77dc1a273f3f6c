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
// reads the window next).  Chunk windows are produced by doubling (1 -> 2 -> 4 ... windows per round).
// Without polynomials (or for short requests) a single workgroup generates the stream serially.
#include "ps_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int DEG = 19937;
constexpr int CHUNK_LOG2 = 17;                          // 2^16: more jumps than chunk time saved (1.09 vs 0.94 ms for 11.8 M doubles)
constexpr int64_t CHUNK = (int64_t)1 << CHUNK_LOG2;     // words per chunk

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
static_assert(SEQ_PAD >= DEG + MT_N && JP * PW == MT_N, "jump geometry");

__device__ __forceinline__ uint32_t load_window_word(const uint32_t *parts, int t) {
    uint32_t v = 0;
#pragma unroll
    for (int p = 0; p < JP; ++p) v ^= parts[p * MT_N + t];
    return v;
}

// seq[b] = the 34 * 624 words that start with window b (sequential expansion, one workgroup per source)
__global__ __launch_bounds__(256) void mt_expand_kernel(const uint32_t *src_parts, uint32_t *seq) {
    __shared__ uint32_t mt[2][MT_N];
    const int t = threadIdx.x;
    uint32_t *out = seq + (size_t)blockIdx.x * SEQ_PAD;
    for (int i = t; i < MT_N; i += 256) {
        const uint32_t v = load_window_word(src_parts + (size_t)blockIdx.x * JP * MT_N, i);
        mt[0][i] = v;
        out[i] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int blk = 1; blk < 34; ++blk) {
        next_block(mt[cur], mt[cur ^ 1], t);
        cur ^= 1;
        for (int i = t; i < MT_N; i += 256) out[blk * MT_N + i] = mt[cur][i];
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

// chunk c (window = states[c]) holds stream words [1 + (c0 + c) * CHUNK, +CHUNK); raw[w - w_lo] for w in [w_lo, w_hi)
__global__ __launch_bounds__(256) void mt_chunk_kernel(const uint32_t *states, int64_t c0, int64_t w_lo, int64_t w_hi,
                                                       uint32_t *raw) {
    __shared__ uint32_t mt[2][MT_N];
    const int t = threadIdx.x;
    const int64_t wbase = 1 + (c0 + blockIdx.x) * CHUNK;
    for (int i = t; i < MT_N; i += 256) mt[0][i] = load_window_word(states + (size_t)blockIdx.x * JP * MT_N, i);
    __syncthreads();
    int cur = 0;
    for (int64_t off = 0; off < CHUNK; off += MT_N) {
        const int take = (CHUNK - off) < MT_N ? (int)(CHUNK - off) : MT_N;
        for (int i = t; i < take; i += 256) {
            const int64_t w = wbase + off + i;
            if (w >= w_lo && w < w_hi) raw[w - w_lo] = mt[cur][i];
        }
        if (off + MT_N < CHUNK) {
            next_block(mt[cur], mt[cur ^ 1], t);
            cur ^= 1;
        }
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
    const int64_t first = p.w_lo < 1 ? 1 : p.w_lo;
    p.c0 = (first - 1) / CHUNK;
    p.c1 = (p.w_hi - 2) / CHUNK;
    if (p.c1 < p.c0) p.c1 = p.c0;
    return p;
}

}  // namespace

extern "C" int ps_mt19937_chunk_log2(void) { return CHUNK_LOG2; }

static int mt_generate(const uint32_t *state_in, int pos_in, int64_t skip, int64_t n, double *out, uint32_t *raw_out,
                       uint32_t *state_out, int32_t *pos_out, const uint32_t *jump_polys, int jump_levels,
                       const uint32_t *radix_polys, int radix_levels, void *workspace, size_t workspace_bytes,
                       ps_stream_t stream);

extern "C" int ps_mt19937_random_sample(const uint32_t *state_in, int pos_in, int64_t skip, int64_t n, double *out,
                                        uint32_t *state_out, int32_t *pos_out, const uint32_t *jump_polys,
                                        int jump_levels, const uint32_t *radix_polys, int radix_levels, void *workspace,
                                        size_t workspace_bytes, ps_stream_t stream) {
    return mt_generate(state_in, pos_in, skip, n, out, nullptr, state_out, pos_out, jump_polys, jump_levels, radix_polys,
                       radix_levels, workspace, workspace_bytes, stream);
}

extern "C" int ps_mt19937_raw_stream(const uint32_t *state_in, int pos_in, int64_t n, uint32_t *raw, uint32_t *state_out,
                                     int32_t *pos_out, const uint32_t *jump_polys, int jump_levels,
                                     const uint32_t *radix_polys, int radix_levels, void *workspace, size_t workspace_bytes,
                                     ps_stream_t stream) {
    if (!raw || !jump_polys || !workspace || n < (1 << 17)) return PS_EINVAL;
    return mt_generate(state_in, pos_in, 0, n, nullptr, raw, state_out, pos_out, jump_polys, jump_levels, radix_polys,
                       radix_levels, workspace, workspace_bytes, stream);
}

extern "C" size_t ps_mt19937_workspace_bytes(int64_t skip, int64_t n) {
    if (n < 0 || skip < 0) return 0;
    const Plan p = make_plan(MT_N, skip, n);            // pos only shifts the plan by < 624 words
    const int64_t K = p.c1 - p.c0 + 2;
    const size_t states = align256((size_t)(K + 4) * JP * MT_N * 4);
    const size_t seqs = align256((size_t)(K / 2 + 2) * SEQ_PAD * 4);
    return states + seqs + align256((size_t)(p.w_hi - p.w_lo + 2 * MT_N + CHUNK) * 4) + 4096;
}

static int mt_generate(const uint32_t *state_in, int pos_in, int64_t skip, int64_t n, double *out, uint32_t *raw_out,
                       uint32_t *state_out, int32_t *pos_out, const uint32_t *jump_polys,
                       int jump_levels, const uint32_t *radix_polys, int radix_levels, void *workspace,
                       size_t workspace_bytes, ps_stream_t stream) {
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
    uint32_t *tmpA = states + (size_t)(K + 1) * WSZ;                         // ping-pong for the base jump
    uint32_t *tmpB = tmpA + WSZ;
    uint32_t *word0 = tmpB + WSZ;
    uint32_t *seqs = reinterpret_cast<uint32_t *>(base + align256((size_t)(K + 4) * JP * MT_N * 4));
    uint32_t *raw = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(seqs) + align256((size_t)(K / 2 + 2) * SEQ_PAD * 4));
    if (raw_out) {                                       // raw mode: the chunk generators write straight into the caller's buffer
        if (p.w_lo != 0) return PS_EUNSUPPORTED;         // (skip = 0: word 2i is the first word of uniform i)
        raw = raw_out;
    }
    // 1. W1 (part 0 of tmpA, the other parts zero) and word 0
    if (hipMemsetAsync(tmpA, 0, WSZ * 4, st) != hipSuccess) return PS_ELAUNCH;
    hipLaunchKernelGGL(mt_prepare_kernel, dim3(1), dim3(256), 0, st, state_in, pos_in, tmpA, word0);
    PS_CHECK_LAUNCH();
    // 2. base jump to chunk c0: offset c0 * CHUNK words = set bits of c0 at levels CHUNK_LOG2 + b
    uint32_t *cur = tmpA, *nxt = tmpB;
    for (int b = 0; (p.c0 >> b) != 0; ++b) {
        if (!((p.c0 >> b) & 1)) continue;
        hipLaunchKernelGGL(mt_expand_kernel, dim3(1), dim3(256), 0, st, cur, seqs);
        PS_CHECK_LAUNCH();
        hipLaunchKernelGGL(mt_combine_kernel, dim3(JP), dim3(192), 0, st, seqs, jump_polys + (size_t)(CHUNK_LOG2 + b) * MT_N, nxt);
        PS_CHECK_LAUNCH();
        uint32_t *t = cur; cur = nxt; nxt = t;
    }
    if (hipMemcpyAsync(states, cur, WSZ * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return PS_ELAUNCH;
    // 3. chunk windows: radix-32 rounds (window j * have + r = jump_{j * have chunks}(window r), j = 1..31) when the
    //    multiplier polynomials are given -- two rounds instead of eight serial expansions for 180 chunks -- else doubling
    if (radix_polys && radix_levels > 0) {
        int lvl = 0;
        for (int64_t have = 1; have < K; have *= 32, ++lvl) {
            if (lvl >= radix_levels) return PS_EUNSUPPORTED;
            const int64_t nsrc = (K - have) < have ? (K - have) : have;
            const int64_t jmax = ((K - 1) / have) < 31 ? ((K - 1) / have) : 31;
            hipLaunchKernelGGL(mt_expand_kernel, dim3((unsigned)nsrc), dim3(256), 0, st, states, seqs);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_combine_radix_kernel, dim3((unsigned)(nsrc * JP), (unsigned)jmax), dim3(192), 0, st, seqs,
                               radix_polys + (size_t)lvl * 31 * MT_N, states, have, K);
            PS_CHECK_LAUNCH();
        }
    } else {
        for (int m = 0; ((int64_t)1 << m) < K; ++m) {
            const int64_t have = (int64_t)1 << m;
            const int64_t make = (K - have) < have ? (K - have) : have;
            hipLaunchKernelGGL(mt_expand_kernel, dim3((unsigned)make), dim3(256), 0, st, states, seqs);
            PS_CHECK_LAUNCH();
            hipLaunchKernelGGL(mt_combine_kernel, dim3((unsigned)(make * JP)), dim3(192), 0, st, seqs,
                               jump_polys + (size_t)(CHUNK_LOG2 + m) * MT_N, states + (size_t)have * WSZ);
            PS_CHECK_LAUNCH();
        }
    }
    // 4. chunks -> raw words; word 0 separately
    hipLaunchKernelGGL(mt_chunk_kernel, dim3((unsigned)K), dim3(256), 0, st, states, p.c0, p.w_lo, p.w_hi, raw);
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
    if (p.key_w >= 0) {
        hipLaunchKernelGGL(mt_final_state_kernel, dim3(1), dim3(256), 0, st, raw, p.w_lo, p.key_w, p.pos_out, state_out, pos_out);
        PS_CHECK_LAUNCH();
    } else {
        if (hipMemcpyAsync(state_out, state_in, MT_N * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return PS_ELAUNCH;
        hipLaunchKernelGGL(mt_set_pos_kernel, dim3(1), dim3(1), 0, st, p.pos_out, pos_out);
        PS_CHECK_LAUNCH();
    }
    return PS_OK;
}
