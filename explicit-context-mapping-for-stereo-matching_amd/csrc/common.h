// Shared helpers for the gfx950 kernels of libecm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ecm_hip.h"

#define ECM_CHECK_ARG(cond) do { if (!(cond)) return ECM_EINVAL; } while (0)
#define ECM_LAUNCH_RESULT() ((int)hipGetLastError())

static inline hipStream_t ecm_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// 64-lane wave reductions (CDNA wave = 64).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.01f * x; }   // nn.LeakyReLU default slope
