// Shared helpers for the gfx950 kernels of libpinsage_hip.so (CDNA4 only: wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pinsage_hip.h"

#define PS_WAVE 64

#define PS_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return PS_ELAUNCH;            \
    } while (0)

static inline hipStream_t ps_stream(ps_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ps_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ int ps_lane() { return threadIdx.x & 63; }

// All LDS traffic of a wave is issued in order; this makes earlier LDS writes/atomics of the
// wave visible to all of its lanes (waits lgkmcnt) and stops the compiler reordering across it.
__device__ __forceinline__ void ps_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int ps_wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ps_wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
