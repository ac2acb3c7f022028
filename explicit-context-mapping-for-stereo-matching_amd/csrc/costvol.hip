// a1: shift-and-concat cost volume (reference cmfsm.py:667-682), fwd + bwd.  HBM-bound.
//
// Layout: L,R [B,C,h,w] -> cost [B,2C,D,h,w] (NCDHW, as the reference holds it).
// One workgroup owns a 1024-float segment of one (b,c) feature plane and streams it to all D
// disparity planes of both halves: the segment is read once (L to registers, R to LDS with a
// D-float halo on the left) and written 2*D times with 16-byte coalesced stores, so HBM sees
// the algorithmic traffic only (2C*D*h*w*4 written + 2*C*h*w*4 read).
// The shift x-d is wave-uniform per iteration: d = 4q+r, so the R operand is two ALIGNED LDS
// float4s (one new ds_read_b128 per four disparities) recombined with a compile-time r.
#include "common.h"

namespace {

constexpr int SEG = 1024;      // floats per workgroup segment (256 threads x float4)
constexpr int THREADS = 256;

__device__ __forceinline__ float4 shift_combine(const float4& lo, const float4& hi, int r) {
    // elements [4-r .. 7-r] of the 8-vector {lo, hi}
    switch (r) {
        case 0: return hi;
        case 1: return make_float4(lo.w, hi.x, hi.y, hi.z);
        case 2: return make_float4(lo.z, lo.w, hi.x, hi.y);
        default: return make_float4(lo.y, lo.z, lo.w, hi.x);
    }
}

template <int R>
__device__ __forceinline__ void emit_one(float* __restrict__ outL, float* __restrict__ outR, size_t plane,
                                         int d, int x, const float4& Lv, const float4& lo, const float4& hi, int D) {
    if (d >= D) return;
    float4 l = Lv, r = shift_combine(lo, hi, R);
    if (x + 0 < d) { l.x = 0.f; r.x = 0.f; }
    if (x + 1 < d) { l.y = 0.f; r.y = 0.f; }
    if (x + 2 < d) { l.z = 0.f; r.z = 0.f; }
    if (x + 3 < d) { l.w = 0.f; r.w = 0.f; }
    ecm_st_stream(outL + (size_t)d * plane, l);
    ecm_st_stream(outR + (size_t)d * plane, r);
}

// w % 4 == 0 (so a float4 never straddles a row) and hw % 4 == 0.
__global__ __launch_bounds__(THREADS) void costvol_fwd_v4(const float* __restrict__ L, const float* __restrict__ R,
                                                          float* __restrict__ cost, int C, int hw, int w, int D,
                                                          int dpad) {
    extern __shared__ __attribute__((aligned(16))) float Rs[];   // [dpad + SEG]
    const int bc = blockIdx.y;               // b*C + c
    const int b = bc / C, c = bc - b * C;
    const int f0 = blockIdx.x * SEG;
    const float* Lp = L + (size_t)bc * hw;
    const float* Rp = R + (size_t)bc * hw;
    // stage R[f0-dpad, f0+SEG)
    for (int i = threadIdx.x; i < (dpad + SEG) / 4; i += THREADS) {
        int f = f0 - dpad + 4 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f >= 0 && f < hw) v = *reinterpret_cast<const float4*>(Rp + f);
        reinterpret_cast<float4*>(Rs)[i] = v;
    }
    __syncthreads();
    const int f = f0 + threadIdx.x * 4;
    if (f >= hw) return;
    const int x = f % w;
    const float4 Lv = *reinterpret_cast<const float4*>(Lp + f);
    const size_t plane = (size_t)hw;
    float* outL = cost + ((size_t)(b * 2 * C + c) * D) * plane + f;
    float* outR = cost + ((size_t)(b * 2 * C + C + c) * D) * plane + f;
    const float4* Rs4 = reinterpret_cast<const float4*>(Rs) + dpad / 4 + threadIdx.x;   // block holding x..x+3
    float4 hi = Rs4[0];
    for (int q = 0; 4 * q < D; ++q) {
        const float4 lo = Rs4[-(q + 1)];       // x-4q-4 .. x-4q-1  (index >= 0 because dpad >= D+3 rounded)
        const int d = 4 * q;
        emit_one<0>(outL, outR, plane, d + 0, x, Lv, lo, hi, D);
        emit_one<1>(outL, outR, plane, d + 1, x, Lv, lo, hi, D);
        emit_one<2>(outL, outR, plane, d + 2, x, Lv, lo, hi, D);
        emit_one<3>(outL, outR, plane, d + 3, x, Lv, lo, hi, D);
        hi = lo;
    }
}

// Generic scalar fallback (any w): one thread per (b, c2, y, x), loops over d.
__global__ void costvol_fwd_scalar(const float* __restrict__ L, const float* __restrict__ R, float* __restrict__ cost,
                                   int C, int hw, int w, int D, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int f = (int)(i % hw);
    int bc = (int)(i / hw);
    int b = bc / C, c = bc - b * C;
    int x = f % w;
    float lv = L[i], rv;
    float* outL = cost + ((size_t)(b * 2 * C + c) * D) * hw + f;
    float* outR = cost + ((size_t)(b * 2 * C + C + c) * D) * hw + f;
    for (int d = 0; d < D; ++d) {
        bool ok = x >= d;
        rv = ok ? R[i - d] : 0.f;
        outL[(size_t)d * hw] = ok ? lv : 0.f;
        outR[(size_t)d * hw] = rv;
    }
}

// ---- backward -------------------------------------------------------------------------------
// gL[x] = sum_{d<=x} gcost[c,d,x];  gR[x'] = sum_{d: x'+d<w} gcost[C+c,d,x'+d].
// Thread owns the float4 at flat f (x..x+3) of one (b,c) plane.  The diagonal gather for gR uses
// two ALIGNED float4 loads per disparity (x+4q and x+4q+4; the second is the neighbour lane's first
// and hits L1), recombined with the wave-uniform r = d%4.
template <int R>
__device__ __forceinline__ void acc_one(const float* __restrict__ gLp, const float* __restrict__ gRp, size_t plane,
                                        int d, int x, int w, int D, float4& aL, float4& aR, const float4& lo,
                                        const float4& hi) {
    if (d >= D) return;
    const float4 g = *reinterpret_cast<const float4*>(gLp + (size_t)d * plane);
    if (x + 0 >= d) aL.x += g.x;
    if (x + 1 >= d) aL.y += g.y;
    if (x + 2 >= d) aL.z += g.z;
    if (x + 3 >= d) aL.w += g.w;
    // elements x+d .. x+d+3 of the gR row = elements [R .. R+3] of {lo (x+4q..), hi (x+4q+4..)}
    float4 v;
    switch (R) {
        case 0: v = lo; break;
        case 1: v = make_float4(lo.y, lo.z, lo.w, hi.x); break;
        case 2: v = make_float4(lo.z, lo.w, hi.x, hi.y); break;
        default: v = make_float4(lo.w, hi.x, hi.y, hi.z); break;
    }
    if (x + 0 + d < w) aR.x += v.x;
    if (x + 1 + d < w) aR.y += v.y;
    if (x + 2 + d < w) aR.z += v.z;
    if (x + 3 + d < w) aR.w += v.w;
}

__device__ __forceinline__ float4 ld4_row(const float* p, int x, int w) {   // zero beyond the row end
    if (x < w) return *reinterpret_cast<const float4*>(p);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(THREADS) void costvol_bwd_v4(const float* __restrict__ gcost, float* __restrict__ gL,
                                                          float* __restrict__ gR, int C, int hw, int w, int D) {
    const int bc = blockIdx.y;
    const int b = bc / C, c = bc - b * C;
    const int f = blockIdx.x * SEG + threadIdx.x * 4;
    if (f >= hw) return;
    const int x = f % w;
    const size_t plane = (size_t)hw;
    const float* gLp = gcost + ((size_t)(b * 2 * C + c) * D) * plane + f;
    const float* gRp = gcost + ((size_t)(b * 2 * C + C + c) * D) * plane + f;
    float4 aL = make_float4(0.f, 0.f, 0.f, 0.f), aR = aL;
    for (int q = 0; 4 * q < D; ++q) {
        const int d = 4 * q;
        // rows of plane d+r shifted by 4q (+4); each r has its own plane, so load per r.
#define ECM_STEP(RR)                                                                                     \
        if (d + RR < D) {                                                                                \
            const float* p = gRp + (size_t)(d + RR) * plane + 4 * q;                                     \
            const float4 lo = ld4_row(p, x + 4 * q, w);                                                  \
            const float4 hi = ld4_row(p + 4, x + 4 * q + 4, w);                                          \
            acc_one<RR>(gLp, gRp, plane, d + RR, x, w, D, aL, aR, lo, hi);                               \
        }
        ECM_STEP(0) ECM_STEP(1) ECM_STEP(2) ECM_STEP(3)
#undef ECM_STEP
    }
    *reinterpret_cast<float4*>(gL + (size_t)bc * hw + f) = aL;
    *reinterpret_cast<float4*>(gR + (size_t)bc * hw + f) = aR;
}

__global__ void costvol_bwd_scalar(const float* __restrict__ gcost, float* __restrict__ gL, float* __restrict__ gR,
                                   int C, int hw, int w, int D, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int f = (int)(i % hw);
    int bc = (int)(i / hw);
    int b = bc / C, c = bc - b * C;
    int x = f % w;
    const float* gLp = gcost + ((size_t)(b * 2 * C + c) * D) * hw + f;
    const float* gRp = gcost + ((size_t)(b * 2 * C + C + c) * D) * hw + f;
    float aL = 0.f, aR = 0.f;
    for (int d = 0; d < D; ++d) {
        if (x >= d) aL += gLp[(size_t)d * hw];
        if (x + d < w) aR += gRp[(size_t)d * hw + d];
    }
    gL[i] = aL;
    gR[i] = aR;
}

}  // namespace

extern "C" int ecm_costvol_concat_fwd(const float* L, const float* R, float* cost, int B, int C, int h, int w, int D,
                                      void* stream) {
    ECM_CHECK_ARG(L && R && cost && B > 0 && C > 0 && h > 0 && w > 0 && D > 0);
    const int hw = h * w;
    if (w % 4 == 0 && (long long)B * C <= 65535) {
        const int dpad = ((D + 3) / 4 + 1) * 4;
        dim3 grid((hw + SEG - 1) / SEG, B * C);
        hipLaunchKernelGGL(costvol_fwd_v4, grid, dim3(THREADS), (dpad + SEG) * sizeof(float), ecm_stream(stream), L, R,
                           cost, C, hw, w, D, dpad);
    } else {
        long long total = (long long)B * C * hw;
        hipLaunchKernelGGL(costvol_fwd_scalar, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ecm_stream(stream),
                           L, R, cost, C, hw, w, D, total);
    }
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_costvol_concat_bwd(const float* gcost, float* gL, float* gR, int B, int C, int h, int w, int D,
                                      void* stream) {
    ECM_CHECK_ARG(gcost && gL && gR && B > 0 && C > 0 && h > 0 && w > 0 && D > 0);
    const int hw = h * w;
    if (w % 4 == 0 && (long long)B * C <= 65535) {
        dim3 grid((hw + SEG - 1) / SEG, B * C);
        hipLaunchKernelGGL(costvol_bwd_v4, grid, dim3(THREADS), 0, ecm_stream(stream), gcost, gL, gR, C, hw, w, D);
    } else {
        long long total = (long long)B * C * hw;
        hipLaunchKernelGGL(costvol_bwd_scalar, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ecm_stream(stream),
                           gcost, gL, gR, C, hw, w, D, total);
    }
    return ECM_LAUNCH_RESULT();
}
