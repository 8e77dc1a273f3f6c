// Shared helpers for the gfx950 kernels of libpinsage_hip.so (CDNA4 only: wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/pinsage_hip.h"

#define PS_WAVE 64

#define PS_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return PS_ELAUNCH;            \
    } while (0)

static inline hipStream_t ps_stream(ps_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Per-device results of one-time host queries (function attributes set, resident-workgroup counts).  The work behind them is
// idempotent, so two host threads that make a first call at the same time merely both do it; the slots are atomics so that this
// repeat is the ONLY consequence (no torn or stale read): the library keeps no state a caller could observe
// (include/pinsage_hip.h).  Zero-initialised statics; 0 = not known yet.
struct PsPerDevice {
    std::atomic<int> v[64];
    int get(int dev) const { return v[dev].load(std::memory_order_acquire); }
    void set(int dev, int x) { v[dev].store(x, std::memory_order_release); }
};

static inline int64_t ps_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// internal (not part of the C ABI): grouped x W^T of csrc/dense_mfma.hip, used by the inverted-file scan
int psi_linear_grouped(const float *x, int64_t M, int K, const float *W, int ldw, float *y, const int64_t *grp, int max_cols,
                       ps_stream_t stream);

__device__ __forceinline__ int ps_lane() { return threadIdx.x & 63; }

// All LDS traffic of a wave is issued in order; this makes earlier LDS writes/atomics of the
// wave visible to all of its lanes (waits lgkmcnt) and stops the compiler reordering across it.
__device__ __forceinline__ void ps_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide sums, result in every lane.  DPP only (quad_perm, row_half_mirror, row_mirror, row_bcast:15 / :31, then a
// v_readlane of lane 63): __shfl_xor goes through the LDS crossbar, which all waves of a CU share.
#define PS_DPP_MOV(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rows), 0xf, true)
__device__ __forceinline__ int ps_wave_sum_i32(int v) {
    v += PS_DPP_MOV(v, 0xB1, 0xf);
    v += PS_DPP_MOV(v, 0x4E, 0xf);
    v += PS_DPP_MOV(v, 0x141, 0xf);
    v += PS_DPP_MOV(v, 0x140, 0xf);
    v += PS_DPP_MOV(v, 0x142, 0xa);
    v += PS_DPP_MOV(v, 0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float ps_wave_sum_f32(float v) {
    v += __builtin_bit_cast(float, PS_DPP_MOV(__builtin_bit_cast(int, v), 0xB1, 0xf));
    v += __builtin_bit_cast(float, PS_DPP_MOV(__builtin_bit_cast(int, v), 0x4E, 0xf));
    v += __builtin_bit_cast(float, PS_DPP_MOV(__builtin_bit_cast(int, v), 0x141, 0xf));
    v += __builtin_bit_cast(float, PS_DPP_MOV(__builtin_bit_cast(int, v), 0x140, 0xf));
    v += __builtin_bit_cast(float, PS_DPP_MOV(__builtin_bit_cast(int, v), 0x142, 0xa));
    v += __builtin_bit_cast(float, PS_DPP_MOV(__builtin_bit_cast(int, v), 0x143, 0xc));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef PS_DPP_MOV
