// a8 + a9: disparity heads.  soft-argmin over D' for the three heads (cmfsm.py:703-706, 725-728,
// 748-753; disparityregression 111-123) and NN-upsample + 9-neighbour weighted sum
// (cmfsm.py:709-723, 730-744, 755-769).  All HBM/L2-bound, one thread per pixel column / HR pixel.
#include "common.h"

namespace {

// neighbour order of the reference's return tuple (cmfsm.py:551,585-593): c,l,r,t,b,lt,rt,lb,rb
__constant__ int kDy[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
__constant__ int kDx[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};

// ---- soft-argmin, heads fused: logits_k = c_0 + ... + c_k ----------------------------------
template <int NH>
__global__ __launch_bounds__(256) void softargmin_fwd(const float* __restrict__ c0, long long hs,
                                                      float* __restrict__ disp, int B, int D, int hw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // b*hw + p
    if (i >= (long long)B * hw) return;
    const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
    const float* base = c0 + (size_t)b * D * hw + p;
    float m[NH], s[NH], t[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { m[k] = -INFINITY; }
    // pass 1: maxima (torch's softmax: exp(x - max) / sum); pass 2 re-reads from L2
    for (int d = 0; d < D; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            acc += base[(size_t)k * hs + (size_t)d * hw];
            m[k] = fmaxf(m[k], acc);
        }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) { s[k] = 0.f; t[k] = 0.f; }
    for (int d = 0; d < D; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            acc += base[(size_t)k * hs + (size_t)d * hw];
            const float e = expf(acc - m[k]);
            s[k] += e;
            t[k] += e * (float)d;
        }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) disp[(size_t)k * B * hw + i] = t[k] / s[k];
}

// d disp_k / d logit_k[d] = p_k[d] (d - disp_k);  g c_j = sum_{k>=j} g logit_k
template <int NH>
__global__ __launch_bounds__(256) void softargmin_bwd(const float* __restrict__ c0, long long hs,
                                                      const float* __restrict__ gdisp, float* __restrict__ gc0,
                                                      int B, int D, int hw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * hw) return;
    const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
    const size_t off = (size_t)b * D * hw + p;
    const float* base = c0 + off;
    float m[NH], s[NH], t[NH], g[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { m[k] = -INFINITY; }
    for (int d = 0; d < D; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            acc += base[(size_t)k * hs + (size_t)d * hw];
            m[k] = fmaxf(m[k], acc);
        }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) { s[k] = 0.f; t[k] = 0.f; g[k] = gdisp[(size_t)k * B * hw + i]; }
    for (int d = 0; d < D; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            acc += base[(size_t)k * hs + (size_t)d * hw];
            const float e = expf(acc - m[k]);
            s[k] += e;
            t[k] += e * (float)d;
        }
    }
    float inv[NH], dk[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { inv[k] = 1.f / s[k]; dk[k] = t[k] * inv[k]; }
    for (int d = 0; d < D; ++d) {
        float acc = 0.f, gl[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            acc += base[(size_t)k * hs + (size_t)d * hw];
            gl[k] = expf(acc - m[k]) * inv[k] * ((float)d - dk[k]) * g[k];
        }
        float run = 0.f;
#pragma unroll
        for (int k = NH - 1; k >= 0; --k) {
            run += gl[k];
            gc0[off + (size_t)k * hs + (size_t)d * hw] = run;
        }
    }
}

__global__ __launch_bounds__(256) void dispreg_fwd(const float* __restrict__ x, float* __restrict__ out, int B, int D,
                                                   int hw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * hw) return;
    const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
    const float* base = x + (size_t)b * D * hw + p;
    float t = 0.f;
    for (int d = 0; d < D; ++d) t += base[(size_t)d * hw] * (float)d;
    out[i] = t;
}

// ---- 9-neighbour aggregation ----------------------------------------------------------------
// One thread per HR pixel; consecutive threads walk X so w9/out accesses are coalesced and the 3x3
// LR patch reads are broadcast within a cell (L1/L2 hits: d is 138 KB per head).
template <int NH>
__global__ __launch_bounds__(256) void aggregate9_fwd(const float* __restrict__ d, const float* __restrict__ w9,
                                                      float* __restrict__ out, int B, int h, int w, int s) {
    const int H = h * s, W = w * s;
    const long long HW = (long long)H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // b*HW + Y*W + X
    if (i >= B * HW) return;
    const int b = (int)(i / HW);
    const int r = (int)(i - b * HW);
    const int Y = r / W, X = r - Y * W;
    const int cy = Y / s, cx = X / s;
    float acc[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) acc[k] = 0.f;
    const float fs = (float)s;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        const int yy = cy + kDy[n], xx = cx + kDx[n];
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
        const float wn = w9[((size_t)b * 9 + n) * HW + r];
#pragma unroll
        for (int k = 0; k < NH; ++k)
            acc[k] += (fs * d[((size_t)k * B + b) * h * w + (size_t)yy * w + xx]) * wn;
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) out[((size_t)k * B + b) * HW + r] = acc[k];
}

// gw9[b,n,Y,X] = sum_k gout[k,b,Y,X] * s * d[k,b,cell+n]   (0 where the neighbour is outside)
template <int NH>
__global__ __launch_bounds__(256) void aggregate9_bwd_w(const float* __restrict__ d, const float* __restrict__ gout,
                                                        float* __restrict__ gw9, int B, int h, int w, int s) {
    const int H = h * s, W = w * s;
    const long long HW = (long long)H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW) return;
    const int b = (int)(i / HW);
    const int r = (int)(i - b * HW);
    const int Y = r / W, X = r - Y * W;
    const int cy = Y / s, cx = X / s;
    float g[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) g[k] = gout[((size_t)k * B + b) * HW + r] * (float)s;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        const int yy = cy + kDy[n], xx = cx + kDx[n];
        float v = 0.f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
#pragma unroll
            for (int k = 0; k < NH; ++k) v += g[k] * d[((size_t)k * B + b) * h * w + (size_t)yy * w + xx];
        }
        gw9[((size_t)b * 9 + n) * HW + r] = v;
    }
}

// gd[k,b,cell] = s * sum_n sum_{pixels p of cell-n} w9[b,n,p] * gout[k,b,p]   (gather form, deterministic).
// One wave per (b, LR cell): lanes cover the 9 source cells' pixels.
template <int NH>
__global__ __launch_bounds__(256) void aggregate9_bwd_d(const float* __restrict__ w9, const float* __restrict__ gout,
                                                        float* __restrict__ gd, int B, int h, int w, int s) {
    const int H = h * s, W = w * s;
    const long long HW = (long long)H * W;
    const int lane = threadIdx.x & 63;
    const long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // b*h*w + cy*w + cx
    if (cell >= (long long)B * h * w) return;      // wave-uniform
    const int b = (int)(cell / (h * w));
    const int c = (int)(cell - (long long)b * h * w);
    const int cy = c / w, cx = c - cy * w;
    float acc[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) acc[k] = 0.f;
    const int ss = s * s;
    for (int j = lane; j < 9 * ss; j += 64) {
        const int n = j / ss, q = j - n * ss;
        const int sy = cy - kDy[n], sx = cx - kDx[n];          // the HR cell whose plane-n weight points at (cy,cx)
        if (sy < 0 || sy >= h || sx < 0 || sx >= w) continue;
        const int Y = sy * s + q / s, X = sx * s + q % s;
        const size_t r = (size_t)Y * W + X;
        const float wn = w9[((size_t)b * 9 + n) * HW + r];
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[k] += wn * gout[((size_t)k * B + b) * HW + r];
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) gd[((size_t)k * B + b) * h * w + c] = v * (float)s;
    }
}

}  // namespace

#define DISPATCH_NH(KERNEL, ...)                                                       \
    switch (nheads) {                                                                  \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                     \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                     \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                     \
        default: return ECM_EUNSUP;                                                    \
    }

extern "C" int ecm_softargmin_heads_fwd(const float* c0, long long head_stride, float* disp, int nheads, int B, int D,
                                        int hw, void* stream) {
    ECM_CHECK_ARG(c0 && disp && B > 0 && D > 0 && hw > 0);
    const long long n = (long long)B * hw;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(softargmin_fwd, grid, block, 0, ecm_stream(stream), c0, head_stride, disp, B, D, hw)
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_softargmin_heads_bwd(const float* c0, long long head_stride, const float* gdisp, float* gc0,
                                        int nheads, int B, int D, int hw, void* stream) {
    ECM_CHECK_ARG(c0 && gdisp && gc0 && B > 0 && D > 0 && hw > 0);
    const long long n = (long long)B * hw;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(softargmin_bwd, grid, block, 0, ecm_stream(stream), c0, head_stride, gdisp, gc0, B, D, hw)
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_disparity_regression_fwd(const float* x, float* out, int B, int D, int hw, void* stream) {
    ECM_CHECK_ARG(x && out && B > 0 && D > 0 && hw > 0);
    const long long n = (long long)B * hw;
    hipLaunchKernelGGL(dispreg_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ecm_stream(stream), x, out, B, D,
                       hw);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_aggregate9_fwd(const float* d, const float* w9, float* out, int nheads, int B, int h, int w, int s,
                                  void* stream) {
    ECM_CHECK_ARG(d && w9 && out && B > 0 && h > 0 && w > 0 && s > 0);
    const long long n = (long long)B * h * s * w * s;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(aggregate9_fwd, grid, block, 0, ecm_stream(stream), d, w9, out, B, h, w, s)
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_aggregate9_bwd(const float* d, const float* w9, const float* gout, float* gd, float* gw9,
                                  int nheads, int B, int h, int w, int s, void* stream) {
    ECM_CHECK_ARG(d && w9 && gout && gd && gw9 && B > 0 && h > 0 && w > 0 && s > 0);
    const long long n = (long long)B * h * s * w * s;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(aggregate9_bwd_w, grid, block, 0, ecm_stream(stream), d, gout, gw9, B, h, w, s)
    const long long cells = (long long)B * h * w;
    dim3 grid2((unsigned)((cells + 3) / 4));
    DISPATCH_NH(aggregate9_bwd_d, grid2, block, 0, ecm_stream(stream), w9, gout, gd, B, h, w, s)
    return ECM_LAUNCH_RESULT();
}
