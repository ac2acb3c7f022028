// First aggregation conv on the concat cost volume without the reference-image half (cmfsm.py:667-684:
// `cost[:, :C, i, :, i:] = L[..., i:]` followed by dres0's Conv3d(64, 32, 3, pad 1)).
//
// The reference-image half of the volume is costL[c,d,h,w] = L[c,h,w]*[w >= d]: constant along d where it is not zero.
// Its contribution to the 3x3x3 convolution at (d,w) therefore depends on d only through which taps pass the wedge
// test  w+kw-1 >= d+kd-1  <=>  kw-kd >= d-w  and the depth-padding test 0 <= d+kd-1 < D, i.e. through
//     class(d,w) = (clamp(d-w, -2, 2), first / interior / last d)              (15 classes; d-w >= 3: no tap passes)
// and equals P[class][co,h,w] = conv2d(L, sum_{kd passing} W[:, :C, kd])[co,h,w].  So
//     conv3d(cost, W) = conv3d(costR, W[:, C:]) + P[class(d,w)]
// costs a 32->32 3-D conv + fifteen 32->32 2-D convs on the feature map instead of a 64->32 3-D conv, and the
// reference-image half of the 4-D volume is never written or read.  These kernels do the class gather (forward, in
// place on the 3-D conv output) and its adjoint (backward: sums of gy over the d of each class).  Needs D >= 2.
#include "common.h"

namespace {

constexpr int NCLS = 15;

__device__ __forceinline__ int edge_of(int d, int D) { return d == 0 ? 0 : (d == D - 1 ? 2 : 1); }

// y[b,co,d,h,w] += P[b,cls(d,w),co,h,w]      y: [B,Co,D,h,w] in place;  P: [B,15,Co,h,w]
template <bool VEC4>
__global__ __launch_bounds__(256) void class_gather_add(float* __restrict__ y, const float* __restrict__ P, int Co, int D,
                                                        int h, int w, long long nvec) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    constexpr int V = VEC4 ? 4 : 1;
    const int wv = w / V;
    const int x0 = (int)(i % wv) * V;
    long long r = i / wv;
    const int yy = (int)(r % h); r /= h;
    const int d = (int)(r % D); r /= D;
    const int co = (int)(r % Co);
    const int b = (int)(r / Co);
    if (d - (x0 + V - 1) >= 3) return;                       // every lane element is deep in the wedge: nothing to add
    const int e = edge_of(d, D);
    const size_t hw = (size_t)h * w;
    const float* Pb = P + (((size_t)b * NCLS) * Co + co) * hw + (size_t)yy * w;
    float* yp = y + i * V;
    float v[V];
    if (VEC4) { const float4 t = *reinterpret_cast<const float4*>(yp); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else v[0] = *yp;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int x = x0 + k, delta = d - x;
        if (delta < 3) {
            const int dc = (delta < -2 ? -2 : delta) + 2;
            v[k] += Pb[(size_t)(dc * 3 + e) * Co * hw + x];
        }
    }
    if (VEC4) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
    else *yp = v[0];
}

// gP[b,cls,co,h,w] = sum over d with cls(d,w) == cls of gy[b,co,d,h,w]      (one thread per (b,co,h,w), w fastest)
__global__ __launch_bounds__(256) void class_gather_bwd(const float* __restrict__ gy, float* __restrict__ gP, int Co, int D,
                                                        int h, int w, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % w);
    long long r = i / w;
    const int yy = (int)(r % h); r /= h;
    const int co = (int)(r % Co);
    const int b = (int)(r / Co);
    const size_t hw = (size_t)h * w;
    const float* g = gy + (((size_t)b * Co + co) * D) * hw + (size_t)yy * w + x;
    float out[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) out[c] = 0.f;
    // d - x <= -2: every tap passes the wedge test; split by depth edge
    const int da = x - 2 < D - 1 ? x - 2 : D - 1;           // last d of this region
    if (da >= 0) out[0] = g[0];
    {
        float s = 0.f;
        const int hi = da < D - 2 ? da : D - 2;
        for (int d = 1; d <= hi; ++d) s += g[(size_t)d * hw];
        out[1] = s;
    }
    if (da >= D - 1) out[2] = g[(size_t)(D - 1) * hw];
    // d - x = -1, 0, 1, 2: one plane each
#pragma unroll
    for (int k = 1; k <= 4; ++k) {
        const int d = x - 2 + k;
        if (d >= 0 && d < D) {
            const float v = g[(size_t)d * hw];
            const int e = edge_of(d, D);
            out[k * 3 + 0] = e == 0 ? v : 0.f;
            out[k * 3 + 1] = e == 1 ? v : 0.f;
            out[k * 3 + 2] = e == 2 ? v : 0.f;
        }
    }
    float* o = gP + (((size_t)b * NCLS) * Co + co) * hw + (size_t)yy * w + x;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) o[(size_t)c * Co * hw] = out[c];
}

}  // namespace

extern "C" int ecm_costvol_class_add_fwd(float* y, const float* P, int B, int Co, int D, int h, int w, void* stream) {
    ECM_CHECK_ARG(y && P && B > 0 && Co > 0 && h > 0 && w > 0);
    if (D < 2) return ECM_EUNSUP;
    const long long n = (long long)B * Co * D * h * w;
    hipStream_t st = ecm_stream(stream);
    if (w % 4 == 0) {
        const long long nv = n / 4;
        hipLaunchKernelGGL(class_gather_add<true>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, y, P, Co, D, h, w, nv);
    } else {
        hipLaunchKernelGGL(class_gather_add<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, P, Co, D, h, w, n);
    }
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_costvol_class_add_bwd(const float* gy, float* gP, int B, int Co, int D, int h, int w, void* stream) {
    ECM_CHECK_ARG(gy && gP && B > 0 && Co > 0 && h > 0 && w > 0);
    if (D < 2) return ECM_EUNSUP;
    const long long total = (long long)B * Co * h * w;
    hipLaunchKernelGGL(class_gather_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ecm_stream(stream), gy, gP, Co,
                       D, h, w, total);
    return ECM_LAUNCH_RESULT();
}
