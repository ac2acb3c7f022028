// Cost volume + dres0's first Conv3d (cmfsm.py:667-684) WITHOUT the 4-D volume.
//
// The concat volume has two halves that are both constant along a line in (d, x):
//   reference half   costL[c,d,y,x] = L[c,y,x]   * [x >= d]        constant along d
//   target half      costR[c,d,y,x] = R[c,y,x-d] * [x >= d]        constant along d at fixed u = x - d
// so a 3x3x3 convolution over them collapses to 2-D convolutions of the feature maps:
//   reference half:  tap (kd,kh,kw) reads L[y+kh-1, x+kw-1] if kw-kd >= d-x (wedge at the tap) and 0 <= d+kd-1 < D;
//                    the set of passing kd depends on (d,x) only through
//                        classP(d,x) = (clamp(d-x,-2,2), first/interior/last d)            15 classes, d-x >= 3: none
//                    => contribution P[classP][co,y,x],  P[c] = conv2d_3x3(L, sum_{kd passing} W[:, :C, kd]).
//   target half:     tap (kd,kh,kw) reads R[y+kh-1, (x-d) + (kw-kd)]: a 3x5 kernel over (kh, ku = kw-kd) evaluated at
//                    u = x-d, with taps dropped when d+kd-1 is outside [0,D) or x+kw-1 == w (right border); zero for
//                    u+ku < 0 is the wedge.  classQ(d,x) = (first/interior/last d, x == w-1)         6 classes
//                    => contribution Q[classQ][co,y,u],  Q[c] = conv2d_3x5(R, sum_{kd,kw: kw-kd=ku, passing} W[:, C:]).
//   y[b,co,d,y,x] = P[b,classP(d,x),co,y,x] + Q[b,classQ(d,x),co,y,x-d]           (x-d >= -2; below that y = 0)
// The 2-D convolutions (32 -> 15*32 and 32 -> 6*32 channels on the 1/4-resolution map, ~16 GFLOP per pair instead of the
// 183 GFLOP of the 64->32 3-D convolution) run in the host layer; these kernels assemble the output volume and, in
// backward, reduce gy over d into gP / gQ (the adjoint of the assembly).  Q is stored with a 2-column left apron:
// Qp[..., j] = Q[..., u = j - 2], j in [0, w+2).  Needs D >= 2.
#include "common.h"

namespace {

constexpr int NCP = 15, NCQ = 6;

__device__ __forceinline__ int edge_of(int d, int D) { return d == 0 ? 0 : (d == D - 1 ? 2 : 1); }

template <bool VEC4>
__global__ __launch_bounds__(256) void costvol_conv_assemble(const float* __restrict__ P, const float* __restrict__ Qp,
                                                             float* __restrict__ y, int Co, int D, int h, int w,
                                                             long long nvec) {
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
    const int e = edge_of(d, D);
    const size_t hw = (size_t)h * w, hq = (size_t)h * (w + 2);
    const float* Pb = P + (((size_t)b * NCP) * Co + co) * hw + (size_t)yy * w;
    const float* Qb = Qp + (((size_t)b * NCQ) * Co + co) * hq + (size_t)yy * (w + 2);
    float v[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int x = x0 + k, delta = d - x;
        float acc = 0.f;
        if (delta < 3) {
            const int dc = (delta < -2 ? -2 : delta) + 2;
            acc = Pb[(size_t)(dc * 3 + e) * Co * hw + x] + Qb[(size_t)(e * 2 + (x == w - 1 ? 1 : 0)) * Co * hq + (x - d + 2)];
        }
        v[k] = acc;
    }
    if (VEC4) *reinterpret_cast<float4*>(y + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    else y[i] = v[0];
}

// gP[b,cls,co,y,x] = sum over d with classP(d,x) == cls of gy[b,co,d,y,x]      (one thread per (b,co,y,x), x fastest)
__global__ __launch_bounds__(256) void costvol_conv_gp(const float* __restrict__ gy, float* __restrict__ gP, int Co, int D,
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
    float out[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) out[c] = 0.f;
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
    float* o = gP + (((size_t)b * NCP) * Co + co) * hw + (size_t)yy * w + x;
#pragma unroll
    for (int c = 0; c < NCP; ++c) o[(size_t)c * Co * hw] = out[c];
}

// gQp[b,cls,co,y,j] = sum over d with x = (j-2)+d in [0,w) and classQ(d,x) == cls of gy[b,co,d,y,x]
// (one thread per (b,co,y,j), j fastest: for a fixed d consecutive lanes read consecutive x)
__global__ __launch_bounds__(256) void costvol_conv_gq(const float* __restrict__ gy, float* __restrict__ gQp, int Co, int D,
                                                       int h, int w, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int wq = w + 2;
    const int j = (int)(i % wq);
    long long r = i / wq;
    const int yy = (int)(r % h); r /= h;
    const int co = (int)(r % Co);
    const int b = (int)(r / Co);
    const size_t hw = (size_t)h * w;
    const float* g = gy + (((size_t)b * Co + co) * D) * hw + (size_t)yy * w;
    const int u = j - 2;
    float first = 0.f, mid = 0.f, last = 0.f, firstr = 0.f, midr = 0.f, lastr = 0.f;      // (edge, right-border) sums
    const int d0 = u < 0 ? -u : 0;                          // x = u + d >= 0
    const int d1 = w - 1 - u < D - 1 ? w - 1 - u : D - 1;   // x <= w - 1
    for (int d = d0; d <= d1; ++d) {
        const int x = u + d;
        const float v = g[(size_t)d * hw + x];
        const bool rb = x == w - 1;
        if (d == 0) { if (rb) firstr += v; else first += v; }
        else if (d == D - 1) { if (rb) lastr += v; else last += v; }
        else { if (rb) midr += v; else mid += v; }
    }
    const size_t hq = (size_t)h * wq;
    float* o = gQp + (((size_t)b * NCQ) * Co + co) * hq + (size_t)yy * wq + j;
    o[0 * Co * hq] = first; o[1 * Co * hq] = firstr;
    o[2 * Co * hq] = mid;   o[3 * Co * hq] = midr;
    o[4 * Co * hq] = last;  o[5 * Co * hq] = lastr;
}


// ---- row-staged forms (round 4) -----------------------------------------------------------------------------------------
// The element-wise kernels above issue two 4-byte gathers per output (forward, 2.4 TB/s of unique traffic) and read gy twice
// (backward: once per class family, 3.7 TB/s).  Here one workgroup owns one (b, co, y) row set: forward stages the 15 P rows
// and 6 Qp rows (21 w + 12 floats) in LDS with 16-byte coalesced loads and writes the D x w outputs as 16-byte streaming
// stores; backward stages the D x w gradients once and produces both class families from LDS.  Same per-element arithmetic and
// the same order of additions as the element-wise kernels (bit-identical results); those stay as the path for rows that do not
// fit (LDS) or are not a multiple of 4 wide.
__global__ __launch_bounds__(256) void costvol_conv_assemble_rows(const float* __restrict__ P, const float* __restrict__ Qp,
                                                                  float* __restrict__ y, int Co, int D, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wq = w + 2;
    float* Ps = smem;                        // [15][w]
    float* Qs = smem + NCP * w;              // [6][wq]
    int r = blockIdx.x;
    const int yy = r % h; r /= h;
    const int co = r % Co;
    const int b = r / Co;
    const size_t hw = (size_t)h * w, hq = (size_t)h * wq;
    const float* Pb = P + (((size_t)b * NCP) * Co + co) * hw + (size_t)yy * w;
    const float* Qb = Qp + (((size_t)b * NCQ) * Co + co) * hq + (size_t)yy * wq;
    const int w4 = w >> 2;
    for (int i = threadIdx.x; i < NCP * w4; i += 256) {
        const int c = i / w4, x4 = i - c * w4;
        reinterpret_cast<float4*>(Ps + c * w)[x4] = reinterpret_cast<const float4*>(Pb + (size_t)c * Co * hw)[x4];
    }
    for (int i = threadIdx.x; i < NCQ * wq; i += 256) {          // rows of w + 2: 8-byte aligned at best, plain loads
        const int c = i / wq, j = i - c * wq;
        Qs[i] = Qb[(size_t)c * Co * hq + j];
    }
    __syncthreads();
    float* yb = y + (((size_t)b * Co + co) * D) * hw + (size_t)yy * w;
    for (int i = threadIdx.x; i < D * w4; i += 256) {
        const int d = i / w4, x0 = (i - d * w4) * 4;
        const int e = edge_of(d, D);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k, delta = d - x;
            float acc = 0.f;
            if (delta < 3) {
                const int dc = (delta < -2 ? -2 : delta) + 2;
                acc = Ps[(dc * 3 + e) * w + x] + Qs[(e * 2 + (x == w - 1 ? 1 : 0)) * wq + (x - d + 2)];
            }
            v[k] = acc;
        }
        ecm_st_stream(yb + (size_t)d * hw + x0, make_float4(v[0], v[1], v[2], v[3]));
    }
}

__global__ __launch_bounds__(256) void costvol_conv_grad_rows(const float* __restrict__ gy, float* __restrict__ gP,
                                                              float* __restrict__ gQp, int Co, int D, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* G = smem;                         // [D][w]
    const int wq = w + 2;
    int r = blockIdx.x;
    const int yy = r % h; r /= h;
    const int co = r % Co;
    const int b = r / Co;
    const size_t hw = (size_t)h * w, hq = (size_t)h * wq;
    const float* gb = gy + (((size_t)b * Co + co) * D) * hw + (size_t)yy * w;
    const int w4 = w >> 2;
    for (int i = threadIdx.x; i < D * w4; i += 256) {
        const int d = i / w4, x4 = i - d * w4;
        reinterpret_cast<float4*>(G + d * w)[x4] = reinterpret_cast<const float4*>(gb + (size_t)d * hw)[x4];
    }
    __syncthreads();
    // reference half: the adjoint of the 15 wedge / depth-edge classes (same order of additions as costvol_conv_gp)
    for (int x = threadIdx.x; x < w; x += 256) {
        float out[NCP];
#pragma unroll
        for (int c = 0; c < NCP; ++c) out[c] = 0.f;
        const int da = x - 2 < D - 1 ? x - 2 : D - 1;
        if (da >= 0) out[0] = G[x];
        {
            float s = 0.f;
            const int hi = da < D - 2 ? da : D - 2;
            for (int d = 1; d <= hi; ++d) s += G[d * w + x];
            out[1] = s;
        }
        if (da >= D - 1) out[2] = G[(D - 1) * w + x];
#pragma unroll
        for (int k = 1; k <= 4; ++k) {
            const int d = x - 2 + k;
            if (d >= 0 && d < D) {
                const float v = G[d * w + x];
                const int e = edge_of(d, D);
                out[k * 3 + 0] = e == 0 ? v : 0.f;
                out[k * 3 + 1] = e == 1 ? v : 0.f;
                out[k * 3 + 2] = e == 2 ? v : 0.f;
            }
        }
        float* o = gP + (((size_t)b * NCP) * Co + co) * hw + (size_t)yy * w + x;
#pragma unroll
        for (int c = 0; c < NCP; ++c) o[(size_t)c * Co * hw] = out[c];
    }
    // target half: sums along the diagonals x = u + d (same order as costvol_conv_gq)
    for (int j = threadIdx.x; j < wq; j += 256) {
        const int u = j - 2;
        float first = 0.f, mid = 0.f, last = 0.f, firstr = 0.f, midr = 0.f, lastr = 0.f;
        const int d0 = u < 0 ? -u : 0;
        const int d1 = w - 1 - u < D - 1 ? w - 1 - u : D - 1;
        for (int d = d0; d <= d1; ++d) {
            const int x = u + d;
            const float v = G[d * w + x];
            const bool rb = x == w - 1;
            if (d == 0) { if (rb) firstr += v; else first += v; }
            else if (d == D - 1) { if (rb) lastr += v; else last += v; }
            else { if (rb) midr += v; else mid += v; }
        }
        float* o = gQp + (((size_t)b * NCQ) * Co + co) * hq + (size_t)yy * wq + j;
        o[0 * Co * hq] = first; o[1 * Co * hq] = firstr;
        o[2 * Co * hq] = mid;   o[3 * Co * hq] = midr;
        o[4 * Co * hq] = last;  o[5 * Co * hq] = lastr;
    }
}

// ---- class-indexed 2-D kernels of the collapsed first convolution (what ops.costvol_conv3d builds every step) ----------------
// wP[x][co][ci][kh][kw]    = sum_kd mP[x][kd][kw]        w[co][ci][kd][kh][kw]         x = (clamp(d-x,-2,2)+2)*3 + edge, 15 classes
// wQ[x][co][ci][kh][ku]    = sum_{kd,kw} mQ[x][kd][kw][ku] w[co][C+ci][kd][kh][kw]     x = edge*2 + (column == w-1), 6 classes
// with the 0/1 tap masks of csrc/costvol_conv.hip's header (a tap passes the wedge `x >= d` / the depth padding; a passing
// target-half tap lands on column ku = kw - kd + 2 of the sheared 3x5 kernel).  Backward: the transposed sums.
__device__ __forceinline__ bool cls_depth_ok(int e, int kd) { return !((e == 0 && kd == 0) || (e == 2 && kd == 2)); }
__device__ __forceinline__ bool cls_mp(int x, int kd, int kw) { return cls_depth_ok(x % 3, kd) && kw - kd >= x / 3 - 2; }
__device__ __forceinline__ bool cls_mq(int x, int kd, int kw, int ku) {
    return cls_depth_ok(x / 2, kd) && ku == kw - kd + 2 && ((x & 1) == 0 || kw != 2);
}

__global__ __launch_bounds__(256) void class_weights_fwd(const float* __restrict__ w, float* __restrict__ wP, float* __restrict__ wQ,
                                                         int Co, int C) {
    const int nP = 15 * Co * C * 9, nQ = 6 * Co * C * 15;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nP) {
        int r = idx;
        const int kw = r % 3; r /= 3;
        const int kh = r % 3; r /= 3;
        const int ci = r % C; r /= C;
        const int co = r % Co;
        const int x = r / Co;
        float s = 0.f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
            if (cls_mp(x, kd, kw)) s += w[(((size_t)co * 2 * C + ci) * 3 + kd) * 9 + kh * 3 + kw];
        wP[idx] = s;
    } else if (idx < nP + nQ) {
        int r = idx - nP;
        const int ku = r % 5; r /= 5;
        const int kh = r % 3; r /= 3;
        const int ci = r % C; r /= C;
        const int co = r % Co;
        const int x = r / Co;
        float s = 0.f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                if (cls_mq(x, kd, kw, ku)) s += w[(((size_t)co * 2 * C + C + ci) * 3 + kd) * 9 + kh * 3 + kw];
        wQ[idx - nP] = s;
    }
}

__global__ __launch_bounds__(256) void class_weights_bwd(const float* __restrict__ gwP, const float* __restrict__ gwQ,
                                                         float* __restrict__ gw, int Co, int C) {
    const int n = Co * 2 * C * 27;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int r = idx;
    const int kw = r % 3; r /= 3;
    const int kh = r % 3; r /= 3;
    const int kd = r % 3; r /= 3;
    const int cc = r % (2 * C);
    const int co = r / (2 * C);
    float s = 0.f;
    if (cc < C) {
#pragma unroll
        for (int x = 0; x < 15; ++x)
            if (cls_mp(x, kd, kw)) s += gwP[((((size_t)x * Co + co) * C + cc) * 3 + kh) * 3 + kw];
    } else {
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int ku = 0; ku < 5; ++ku)
                if (cls_mq(x, kd, kw, ku)) s += gwQ[((((size_t)x * Co + co) * C + (cc - C)) * 3 + kh) * 5 + ku];
    }
    gw[idx] = s;
}

}  // namespace

extern "C" int ecm_costvol_conv_assemble_fwd(const float* P, const float* Qp, float* y, int B, int Co, int D, int h, int w,
                                             void* stream) {
    ECM_CHECK_ARG(P && Qp && y && B > 0 && Co > 0 && h > 0 && w > 0);
    if (D < 2) return ECM_EUNSUP;
    const long long n = (long long)B * Co * D * h * w;
    hipStream_t st = ecm_stream(stream);
    {
        const long long rows = (long long)B * Co * h;
        const size_t lds = (size_t)(NCP * w + NCQ * (w + 2)) * sizeof(float);
        const bool aligned = (reinterpret_cast<size_t>(P) | reinterpret_cast<size_t>(y)) % 16 == 0;
        if (w % 4 == 0 && aligned && lds <= 64 * 1024 && rows <= 0x7fffffffLL) {
            hipLaunchKernelGGL(costvol_conv_assemble_rows, dim3((unsigned)rows), dim3(256), lds, st, P, Qp, y, Co, D, h, w);
            return ECM_LAUNCH_RESULT();
        }
    }
    if (w % 4 == 0) {
        const long long nv = n / 4;
        hipLaunchKernelGGL(costvol_conv_assemble<true>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, P, Qp, y, Co, D, h,
                           w, nv);
    } else {
        hipLaunchKernelGGL(costvol_conv_assemble<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, Qp, y, Co, D, h,
                           w, n);
    }
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_costvol_conv_assemble_bwd(const float* gy, float* gP, float* gQp, int B, int Co, int D, int h, int w,
                                             void* stream) {
    ECM_CHECK_ARG(gy && gP && gQp && B > 0 && Co > 0 && h > 0 && w > 0);
    if (D < 2) return ECM_EUNSUP;
    hipStream_t st = ecm_stream(stream);
    {
        const long long rows = (long long)B * Co * h;
        const size_t lds = (size_t)D * w * sizeof(float);
        if (w % 4 == 0 && reinterpret_cast<size_t>(gy) % 16 == 0 && lds <= 64 * 1024 && rows <= 0x7fffffffLL) {
            hipLaunchKernelGGL(costvol_conv_grad_rows, dim3((unsigned)rows), dim3(256), lds, st, gy, gP, gQp, Co, D, h, w);
            return ECM_LAUNCH_RESULT();
        }
    }
    const long long tp = (long long)B * Co * h * w, tq = (long long)B * Co * h * (w + 2);
    hipLaunchKernelGGL(costvol_conv_gp, dim3((unsigned)((tp + 255) / 256)), dim3(256), 0, st, gy, gP, Co, D, h, w, tp);
    hipLaunchKernelGGL(costvol_conv_gq, dim3((unsigned)((tq + 255) / 256)), dim3(256), 0, st, gy, gQp, Co, D, h, w, tq);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_costvol_class_weights_fwd(const float* w, float* wP, float* wQ, int Co, int C, void* stream) {
    ECM_CHECK_ARG(w && wP && wQ && Co > 0 && C > 0);
    const long long n = (long long)Co * C * (15 * 9 + 6 * 15);
    if (n > 0x7fffffffLL) return ECM_EUNSUP;
    hipLaunchKernelGGL(class_weights_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ecm_stream(stream), w, wP, wQ, Co, C);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_costvol_class_weights_bwd(const float* gwP, const float* gwQ, float* gw, int Co, int C, void* stream) {
    ECM_CHECK_ARG(gwP && gwQ && gw && Co > 0 && C > 0);
    const long long n = (long long)Co * 2 * C * 27;
    if (n > 0x7fffffffLL) return ECM_EUNSUP;
    hipLaunchKernelGGL(class_weights_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ecm_stream(stream), gwP, gwQ, gw, Co, C);
    return ECM_LAUNCH_RESULT();
}
