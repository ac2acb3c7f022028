// Shared helpers for the gfx950 kernels of libecm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include "../../include/ecm_hip.h"

#define ECM_CHECK_ARG(cond) do { if (!(cond)) return ECM_EINVAL; } while (0)
#define ECM_LAUNCH_RESULT() ((int)hipGetLastError())

static inline hipStream_t ecm_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Raise a kernel's dynamic-LDS limit, once per (kernel, device): function attributes are per device and one process
// may drive several GPUs from several threads (the reference wraps the model in nn.DataParallel, train.py:78-79).
static inline hipError_t ecm_allow_lds(const void* kern, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, unsigned long long> done;      // kernel -> bit mask of configured devices
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lock(mu);
    unsigned long long& mask = done[kern];
    if (mask & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) mask |= bit;
    return e;
}

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
