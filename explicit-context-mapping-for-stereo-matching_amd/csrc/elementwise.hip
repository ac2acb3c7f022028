// n-ary elementwise sum: out = ((a + b) + c) + d in that fixed order (c, d optional).  This is the gradient accumulation of a
// tensor with several consumers (cost0 feeds the first hourglass and the residual adds of all three, cmfsm.py:686-693) done
// in ONE pass -- n reads + 1 write -- instead of autograd's n-1 binary adds (3 passes each).  HBM-bound: 16-byte lanes.
#include "common.h"

namespace {

template <int N>
__global__ __launch_bounds__(256) void sum_n_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ c, const float* __restrict__ d,
                                                    float* __restrict__ out, long long n4, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(a)[i];
        const float4 w = reinterpret_cast<const float4*>(b)[i];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        if (N >= 3) { const float4 u = reinterpret_cast<const float4*>(c)[i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        if (N >= 4) { const float4 u = reinterpret_cast<const float4*>(d)[i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        ecm_st_stream(out + (size_t)i * 4, v);
    }
    // tail (n % 4 elements) by the first threads of the grid
    const long long t = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float v = a[t] + b[t];
        if (N >= 3) v += c[t];
        if (N >= 4) v += d[t];
        out[t] = v;
    }
}

// The shape that streams fastest on this part (round 4, profiles/r04_sum_n_ab.txt): one short-lived workgroup per contiguous
// 16 KB chunk, four float4 per thread and operand issued up front -- 0.885 -> 0.745 ms for the sum of four 849 MB tensors
// (5.7 TB/s), 0.511 -> 0.432 for two (5.9 TB/s, ATen's `add`: 6.0), against the grid-stride loop of 4096 persistent workgroups
// above, which is kept for lengths that are not a multiple of four.  Streaming (nt) or plain stores: no difference here.
template <int N, bool NT>
__global__ __launch_bounds__(256) void sum_n_chunk_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ c, const float* __restrict__ d,
                                                          float* __restrict__ out, long long n4) {
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256;
        if (i < n4) {
            v[k] = reinterpret_cast<const float4*>(a)[i];
            const float4 w = reinterpret_cast<const float4*>(b)[i];
            v[k].x += w.x; v[k].y += w.y; v[k].z += w.z; v[k].w += w.w;
            if (N >= 3) { const float4 u = reinterpret_cast<const float4*>(c)[i]; v[k].x += u.x; v[k].y += u.y; v[k].z += u.z; v[k].w += u.w; }
            if (N >= 4) { const float4 u = reinterpret_cast<const float4*>(d)[i]; v[k].x += u.x; v[k].y += u.y; v[k].z += u.z; v[k].w += u.w; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256;
        if (i < n4) { if (NT) ecm_st_stream(out + (size_t)i * 4, v[k]); else reinterpret_cast<float4*>(out)[i] = v[k]; }
    }
}

// out[b,c,y,x] = (y, x both even) ? small[b,c,y/2,x/2] : 0 -- the data gradient of a 1x1 / stride-2 projection (the encoder's
// downsample layers, cmfsm.py:72-75 via _make_layer) after W^T gy has been formed on the coarse grid: one write pass
__global__ __launch_bounds__(256) void zero_insert2_kernel(const float* __restrict__ small, float* __restrict__ out, long long planes,
                                                           int H, int W, int Hs, int Ws) {
    const long long n = planes * H * W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int x = (int)(i % W);
        const long long r = i / W;
        const int y = (int)(r % H);
        const long long p = r / H;
        float v = 0.f;
        if (((x | y) & 1) == 0 && (y >> 1) < Hs && (x >> 1) < Ws) v = small[(p * Hs + (y >> 1)) * Ws + (x >> 1)];
        out[i] = v;
    }
}

}  // namespace

extern "C" int ecm_zero_insert2d(const float* small, float* out, long long planes, int H, int W, int Hs, int Ws, void* stream) {
    ECM_CHECK_ARG(small && out && planes > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0);
    const long long n = planes * H * W;
    const long long want = (n + 255) / 256;
    hipLaunchKernelGGL(zero_insert2_kernel, dim3((unsigned)(want > 256 * 16 ? 256 * 16 : want)), dim3(256), 0, ecm_stream(stream), small,
                       out, planes, H, W, Hs, Ws);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_sum_n(const float* a, const float* b, const float* c, const float* d, float* out, long long n, void* stream) {
    ECM_CHECK_ARG(a && b && out && n > 0 && (c || !d));
    if ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(c) | reinterpret_cast<size_t>(d) |
         reinterpret_cast<size_t>(out)) & 15)
        return ECM_EUNSUP;                                   // 16-byte aligned operands (torch allocations are)
    hipStream_t st = ecm_stream(stream);
    const long long n4 = n / 4;
    const long long want = (n4 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : want > 256 * 16 ? 256 * 16 : want);
    if (n % 4 == 0 && (n4 + 1023) / 1024 < 0x7fffffffLL) {
        constexpr bool NT = true;
        const unsigned cb = (unsigned)((n4 + 1023) / 1024);
        if (d) hipLaunchKernelGGL((sum_n_chunk_kernel<4, NT>), dim3(cb), dim3(256), 0, st, a, b, c, d, out, n4);
        else if (c) hipLaunchKernelGGL((sum_n_chunk_kernel<3, NT>), dim3(cb), dim3(256), 0, st, a, b, c, d, out, n4);
        else hipLaunchKernelGGL((sum_n_chunk_kernel<2, NT>), dim3(cb), dim3(256), 0, st, a, b, c, d, out, n4);
        return ECM_LAUNCH_RESULT();
    }
    if (d) hipLaunchKernelGGL(sum_n_kernel<4>, dim3(blocks), dim3(256), 0, st, a, b, c, d, out, n4, n);
    else if (c) hipLaunchKernelGGL(sum_n_kernel<3>, dim3(blocks), dim3(256), 0, st, a, b, c, d, out, n4, n);
    else hipLaunchKernelGGL(sum_n_kernel<2>, dim3(blocks), dim3(256), 0, st, a, b, c, d, out, n4, n);
    return ECM_LAUNCH_RESULT();
}
