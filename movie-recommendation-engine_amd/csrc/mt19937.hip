// mt19937.hip -- numpy's legacy global RNG stream, generated on the device.
//
// The reference draws one `random_sample()` double from the process-global MT19937 per taken walk
// step (np.random.choice at utils/random_walk.py:79).  ps_mt19937_random_sample reproduces n such
// doubles from a given state (key[624], pos) and returns the advanced state, so the host can
// `np.random.set_state` afterwards and every later consumer of np.random sees the stream the
// reference would have left behind.
//   word stream : standard MT19937 twist (3 dependency phases per 624 words, LDS resident) + tempering
//   double      : genrand_res53: a = w0 >> 5, b = w1 >> 6, (a * 2^26 + b) / 2^53
// The twist is a serial recurrence (parallelism 227), so the generator is one workgroup; the pair
// -> double conversion is a separate fully parallel pass done in place.
#include "ps_common.h"

namespace {

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

// raw: 2n tempered words written into `out` (reinterpreted), state/pos advanced.
__global__ __launch_bounds__(256) void mt_words_kernel(const uint32_t *state_in, int pos_in, int64_t nwords,
                                                       uint32_t *raw, uint32_t *state_out, int32_t *pos_out) {
    __shared__ uint32_t mt[2][624];
    const int t = threadIdx.x;
    for (int i = t; i < 624; i += 256) mt[0][i] = state_in[i];
    __syncthreads();
    int cur = 0, pos = pos_in;
    int64_t done = 0;
    while (done < nwords) {
        if (pos >= 624) {   // regenerate the whole block: new = mt[cur^1]
            uint32_t *o = mt[cur], *n = mt[cur ^ 1];
            if (t < 227) n[t] = o[t + 397] ^ twist(o[t], o[t + 1]);
            __syncthreads();
            if (t < 227) { const int i = 227 + t; n[i] = n[i - 227] ^ twist(o[i], o[i + 1]); }
            __syncthreads();
            if (t < 169) { const int i = 454 + t; n[i] = n[i - 227] ^ twist(o[i], o[i + 1]); }
            __syncthreads();
            if (t == 0) n[623] = n[396] ^ twist(o[623], n[0]);
            __syncthreads();
            cur ^= 1;
            pos = 0;
        }
        const int64_t rem = nwords - done;
        const int take = (624 - pos) < rem ? (624 - pos) : (int)rem;
        for (int i = t; i < take; i += 256) raw[done + i] = temper(mt[cur][pos + i]);
        pos += take;
        done += take;
    }
    __syncthreads();
    for (int i = t; i < 624; i += 256) state_out[i] = mt[cur][i];
    if (t == 0) pos_out[0] = pos;
}

__global__ void mt_pairs_to_double_kernel(double *out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 w = reinterpret_cast<const uint2 *>(out)[i];
        out[i] = ((double)(w.x >> 5) * 67108864.0 + (double)(w.y >> 6)) * (1.0 / 9007199254740992.0);
    }
}

}  // namespace

extern "C" int ps_mt19937_random_sample(const uint32_t *state_in, int pos_in, int64_t n, double *out,
                                        uint32_t *state_out, int32_t *pos_out, ps_stream_t stream) {
    if (!state_in || !state_out || !pos_out || n < 0 || pos_in < 0 || pos_in > 624) return PS_EINVAL;
    if (n > 0 && !out) return PS_EINVAL;
    hipStream_t st = ps_stream(stream);
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
