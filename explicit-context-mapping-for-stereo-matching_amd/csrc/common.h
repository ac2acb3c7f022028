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

// XCD-aware tile order.  Workgroup ids are dealt round-robin over the 8 XCDs (id % 8), and each XCD has its own L2, so
// handing tile `id` to workgroup `id` puts spatial neighbours (which share halo voxels) on eight different L2s.  This
// maps workgroup id -> tile so that every XCD works through ONE contiguous run of the n tiles: XCD x gets
// q + (x < r) tiles (q = n / 8, r = n % 8) starting at x*q + min(x, r), and its workgroups x, x+8, x+16, ... take them
// in order.  A bijection on [0, n).
__device__ __forceinline__ int ecm_xcd_tile(int id, int n) {
    const int x = id & 7, q = n >> 3, r = n & 7;
    return x * q + (x < r ? x : r) + (id >> 3);
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

// Streaming 16-byte store of a result that nothing in THIS kernel reads again (non-temporal hint: `global_store_dwordx4 ... nt`).
// Measured in round 4 on the GroupNorm kernels (profiles/r04_gn_store_policy.txt): 4-18 % faster launches with the hint on the
// stores, nothing from hinting the loads; the line still stays in the XCD's L2 for the next kernel (MI355X_MICROARCH.md).
__device__ __forceinline__ void ecm_st_stream(float* p, const float4& v) {
#ifdef ECM_NO_STREAM_STORES
    *reinterpret_cast<float4*>(p) = v;
#else
    typedef float f32x4_st __attribute__((ext_vector_type(4)));
    f32x4_st t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_st*>(p));
#endif
}
