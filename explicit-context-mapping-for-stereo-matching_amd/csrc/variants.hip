// Heads of the reference's other registered architectures, each fused into ONE pass that never materialises the
// full-resolution 192 x H x W volume(s) the reference builds (8 x 424.7 MB at 576x960):
//   a10  volume mapping  (cmfsm_sub_16.py:767-801 [+ heads 804-848], cm_sub_8.py:765-800, cm_sub_4/16 alike):
//        NN-upsample the LR logits in D,H,W -> 5-neighbour spatial fuse -> 3 target-image weight volumes shifted by the
//        disparity (python loop over 192 in the reference) -> +-scale fuse along D -> softmax(192) -> regression.
//   a11  trilinear head  (bilinear_cmf.py:447-471): F.interpolate(trilinear, align_corners=False) to [192,H,W] ->
//        softmax(192) -> regression.
// One thread per output pixel, lanes along X (coalesced rows); the LR column is walked with a rolling window and the
// softmax over D is the online (running-max) form.  HBM/L2-bound: inputs are ~20 MB of weight planes + LR logits.
#include "common.h"

namespace {

struct Online {            // online softmax-weighted mean of D
    float m, s, t;
    __device__ __forceinline__ void init() { m = -INFINITY; s = 0.f; t = 0.f; }
    __device__ __forceinline__ void push(float v, float d) {
        if (v > m) { const float sc = expf(m - v); s *= sc; t *= sc; m = v; }
        const float e = expf(v - m);
        s += e;
        t += e * d;
    }
    __device__ __forceinline__ float result() const { return t / s; }
};

// logits of head k at LR voxel: cumulative sum of the raw classifier outputs c_0..c_k (cmfsm_sub_16.py:811,829)
template <int NH>
__device__ __forceinline__ void load_cum(const float* __restrict__ c, long long hs, size_t idx, float (&v)[NH]) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NH; ++k) { acc += c[(size_t)k * hs + idx]; v[k] = acc; }
}

// 5-neighbour fuse of the LR column at depth j: F_k = sum_n m[n] * C_k[j, cell + n], n = c, r, l, t, b (0 outside)
template <int NH>
__device__ __forceinline__ void fuse5(const float* __restrict__ c, long long hs, size_t bbase, int j, int h, int w, int cy,
                                      int cx, const float (&m5)[5], float (&F)[NH]) {
    const int dy[5] = {0, 0, 0, -1, 1}, dx[5] = {0, 1, -1, 0, 0};
#pragma unroll
    for (int k = 0; k < NH; ++k) F[k] = 0.f;
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const int yy = cy + dy[n], xx = cx + dx[n];
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
        float v[NH];
        load_cum<NH>(c, hs, bbase + ((size_t)j * h + yy) * w + xx, v);
#pragma unroll
        for (int k = 0; k < NH; ++k) F[k] = fmaf(m5[n], v[k], F[k]);
    }
}

template <int NH>
__global__ __launch_bounds__(256) void volume_mapping_fwd(const float* __restrict__ c, long long hs,
                                                          const float* __restrict__ m5p, const float* __restrict__ mt3p,
                                                          float* __restrict__ out, int B, int Dl, int h, int w, int s) {
    const int H = h * s, W = w * s;
    const long long HW = (long long)H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW) return;
    const int b = (int)(i / HW);
    const int r = (int)(i - b * HW);
    const int Y = r / W, X = r - Y * W;
    const int cy = Y / s, cx = X / s;
    float m5[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) m5[n] = m5p[((size_t)b * 5 + n) * HW + r];
    const float* mt = mt3p + (size_t)b * 3 * HW + (size_t)Y * W;      // rows of the three target planes (c, r, l)
    const size_t bbase = (size_t)b * Dl * h * w;
    float Fp[NH], Fc[NH], Fn[NH];
    Online acc[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { Fp[k] = 0.f; acc[k].init(); }
    fuse5<NH>(c, hs, bbase, 0, h, w, cy, cx, m5, Fc);
    for (int j = 0; j < Dl; ++j) {
        const bool has_next = j + 1 < Dl;
        if (has_next) fuse5<NH>(c, hs, bbase, j + 1, h, w, cy, cx, m5, Fn);
        for (int q = 0; q < s; ++q) {
            const int D = j * s + q;
            float t0 = 1.f, tr = 1.f, tl = 1.f;                        // ones_like initialisation (cmfsm_sub_16.py:782-784)
            if (X >= D) { t0 = mt[X - D]; tr = mt[HW + X - D]; tl = mt[2 * HW + X - D]; }
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                float v = Fc[k] * t0;
                if (has_next) v = fmaf(Fn[k], tl, v);                  // [:, :-s] += fused[:, s:] * T_l[:, :-s]   (:797)
                if (j > 0) v = fmaf(Fp[k], tr, v);                     // [:, s:]  += fused[:, :-s] * T_r[:, s:]   (:798)
                acc[k].push(v, (float)D);
            }
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) { Fp[k] = Fc[k]; Fc[k] = Fn[k]; }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) out[((size_t)k * B + b) * HW + r] = acc[k].result();
}

// PyTorch upsample_trilinear3d source index, align_corners=False
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <int NH>
__global__ __launch_bounds__(256) void trilinear_softargmin_fwd(const float* __restrict__ c, long long hs,
                                                                float* __restrict__ out, int B, int Dl, int h, int w,
                                                                int Do, int H, int W) {
    const long long HW = (long long)H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW) return;
    const int b = (int)(i / HW);
    const int r = (int)(i - b * HW);
    const int Y = r / W, X = r - Y * W;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(Y, (float)h / (float)H, h, y0, y1, ly);
    src_index(X, (float)w / (float)W, w, x0, x1, lx);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const size_t bbase = (size_t)b * Dl * h * w;
    auto plane = [&](int j, float (&G)[NH]) {
        float a[NH], bq[NH], cq[NH], d[NH];
        const size_t pj = bbase + (size_t)j * h * w;
        load_cum<NH>(c, hs, pj + (size_t)y0 * w + x0, a);
        load_cum<NH>(c, hs, pj + (size_t)y0 * w + x1, bq);
        load_cum<NH>(c, hs, pj + (size_t)y1 * w + x0, cq);
        load_cum<NH>(c, hs, pj + (size_t)y1 * w + x1, d);
#pragma unroll
        for (int k = 0; k < NH; ++k) G[k] = w00 * a[k] + w01 * bq[k] + w10 * cq[k] + w11 * d[k];
    };
    float GA[NH], GB[NH];
    int jA = -1, jB = -1;
    Online acc[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) acc[k].init();
    const float dscale = (float)Dl / (float)Do;
    for (int D = 0; D < Do; ++D) {
        int d0, d1;
        float ld;
        src_index(D, dscale, Dl, d0, d1, ld);
        if (d0 != jA) {
            if (d0 == jB) {
#pragma unroll
                for (int k = 0; k < NH; ++k) GA[k] = GB[k];
            } else {
                plane(d0, GA);
            }
            jA = d0;
        }
        if (d1 != jB) {
            if (d1 == jA) {
#pragma unroll
                for (int k = 0; k < NH; ++k) GB[k] = GA[k];
            } else {
                plane(d1, GB);
            }
            jB = d1;
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[k].push((1.f - ld) * GA[k] + ld * GB[k], (float)D);
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) out[((size_t)k * B + b) * HW + r] = acc[k].result();
}


// ---------------------------------------------------------------------------------------------------------------
// Backward of the volume-mapping head.  One workgroup = 256 consecutive X of one image row.  Per pixel: one forward
// sweep (online softmax statistics), one backward sweep that recomputes v[D], forms gv[D] = p[D] (D - pred) g and
//   * accumulates gF[j] (rolling j-1, j, j+1), from which g_m5 (registers) and the LR-logit gradient follow
//     (reduced over the s lanes of a cell with shuffles, then one float atomic per cell, neighbour, depth and head);
//   * accumulates the three target-plane gradients along the X-D diagonal in LDS (ds atomics), flushed with one
//     global atomic per touched element.
// Float atomics => the LR-logit and target-plane gradients are summed in a non-deterministic order (last-bit noise).
template <int NH>
__global__ __launch_bounds__(256) void volume_mapping_bwd(const float* __restrict__ c, long long hs,
                                                          const float* __restrict__ m5p, const float* __restrict__ mt3p,
                                                          const float* __restrict__ gout, float* __restrict__ gc,
                                                          float* __restrict__ gm5, float* __restrict__ gmt3, int B, int Dl,
                                                          int h, int w, int s, int xblocks) {
    extern __shared__ float lds[];                      // [3][256 + Dmax]
    const int H = h * s, W = w * s, Dmax = Dl * s;
    const long long HW = (long long)H * W;
    int bid = blockIdx.x;
    const int xb = bid % xblocks; bid /= xblocks;
    const int Y = bid % H;
    const int b = bid / H;
    const int X0 = xb * 256, X = X0 + threadIdx.x;
    const int span = 256 + Dmax;
    for (int e = threadIdx.x; e < 3 * span; e += 256) lds[e] = 0.f;
    __syncthreads();
    const bool valid = X < W;
    const int Xc = valid ? X : W - 1;
    const int r = Y * W + Xc;
    const int cy = Y / s, cx = Xc / s;
    float m5[5], gm[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) { m5[n] = m5p[((size_t)b * 5 + n) * HW + r]; gm[n] = 0.f; }
    const float* mt = mt3p + (size_t)b * 3 * HW + (size_t)Y * W;
    const size_t bbase = (size_t)b * Dl * h * w;
    const int dy[5] = {0, 0, 0, -1, 1}, dx[5] = {0, 1, -1, 0, 0};
    float Fp[NH], Fc[NH], Fn[NH];
    // ---- forward sweep: statistics --------------------------------------------------------------------------
    Online acc[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { Fp[k] = 0.f; acc[k].init(); }
    fuse5<NH>(c, hs, bbase, 0, h, w, cy, cx, m5, Fc);
    for (int j = 0; j < Dl; ++j) {
        const bool has_next = j + 1 < Dl;
        if (has_next) fuse5<NH>(c, hs, bbase, j + 1, h, w, cy, cx, m5, Fn);
        for (int q = 0; q < s; ++q) {
            const int D = j * s + q;
            float t0 = 1.f, tr = 1.f, tl = 1.f;
            if (Xc >= D) { t0 = mt[Xc - D]; tr = mt[HW + Xc - D]; tl = mt[2 * HW + Xc - D]; }
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                float v = Fc[k] * t0;
                if (has_next) v = fmaf(Fn[k], tl, v);
                if (j > 0) v = fmaf(Fp[k], tr, v);
                acc[k].push(v, (float)D);
            }
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) { Fp[k] = Fc[k]; Fc[k] = Fn[k]; }
    }
    float mx[NH], inv[NH], pred[NH], g[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        mx[k] = acc[k].m; inv[k] = 1.f / acc[k].s; pred[k] = acc[k].t * inv[k];
        g[k] = valid ? gout[((size_t)k * B + b) * HW + r] : 0.f;
    }
    // ---- backward sweep -------------------------------------------------------------------------------------
    float gFp[NH], gFc[NH], gFn[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { Fp[k] = 0.f; gFp[k] = gFc[k] = gFn[k] = 0.f; }
    fuse5<NH>(c, hs, bbase, 0, h, w, cy, cx, m5, Fc);
    // emit the finished gF of depth jj: g_m5 and the scatter to the LR logits
    auto emit = [&](int jj, const float (&gF)[NH]) {
        float suf[NH];                                   // gradient w.r.t. the RAW head outputs: suffix sums over heads
        float run = 0.f;
#pragma unroll
        for (int k = NH - 1; k >= 0; --k) { run += gF[k]; suf[k] = run; }
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int yy = cy + dy[n], xx = cx + dx[n];
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            const size_t idx = bbase + ((size_t)jj * h + yy) * w + xx;
            float v[NH];
            load_cum<NH>(c, hs, idx, v);
#pragma unroll
            for (int k = 0; k < NH; ++k) gm[n] = fmaf(gF[k], v[k], gm[n]);
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                float t = m5[n] * suf[k];
                for (int off = 1; off < s && off < 64; off <<= 1) t += __shfl_xor(t, off, 64);   // lanes of one cell
                if ((threadIdx.x & (s - 1)) == 0 || s > 64) atomicAdd(gc + (size_t)k * hs + idx, t);
            }
        }
    };
    for (int j = 0; j < Dl; ++j) {
        const bool has_next = j + 1 < Dl;
        if (has_next) fuse5<NH>(c, hs, bbase, j + 1, h, w, cy, cx, m5, Fn);
        for (int q = 0; q < s; ++q) {
            const int D = j * s + q;
            const bool in = Xc >= D;
            float t0 = 1.f, tr = 1.f, tl = 1.f;
            if (in) { t0 = mt[Xc - D]; tr = mt[HW + Xc - D]; tl = mt[2 * HW + Xc - D]; }
            float a0 = 0.f, ar = 0.f, al = 0.f;
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                float v = Fc[k] * t0;
                if (has_next) v = fmaf(Fn[k], tl, v);
                if (j > 0) v = fmaf(Fp[k], tr, v);
                const float gv = expf(v - mx[k]) * inv[k] * ((float)D - pred[k]) * g[k];
                gFc[k] = fmaf(gv, t0, gFc[k]);
                a0 = fmaf(gv, Fc[k], a0);
                if (has_next) { gFn[k] = fmaf(gv, tl, gFn[k]); al = fmaf(gv, Fn[k], al); }
                if (j > 0) { gFp[k] = fmaf(gv, tr, gFp[k]); ar = fmaf(gv, Fp[k], ar); }
            }
            if (in && valid) {
                const int li = (Xc - D) - (X0 - Dmax);          // >= 0
                atomicAdd(&lds[li], a0);
                atomicAdd(&lds[span + li], ar);
                atomicAdd(&lds[2 * span + li], al);
            }
        }
        if (j > 0) emit(j - 1, gFp);
#pragma unroll
        for (int k = 0; k < NH; ++k) { Fp[k] = Fc[k]; Fc[k] = Fn[k]; gFp[k] = gFc[k]; gFc[k] = gFn[k]; gFn[k] = 0.f; }
    }
    emit(Dl - 1, gFp);
    if (valid) {
#pragma unroll
        for (int n = 0; n < 5; ++n) gm5[((size_t)b * 5 + n) * HW + r] = gm[n];
    }
    __syncthreads();
    float* gt = gmt3 + (size_t)b * 3 * HW + (size_t)Y * W;
    for (int e = threadIdx.x; e < 3 * span; e += 256) {
        const int pl = e / span, li = e - pl * span;
        const int xg = X0 - Dmax + li;
        const float v = lds[e];
        if (xg >= 0 && xg < W && v != 0.f) atomicAdd(gt + (size_t)pl * HW + xg, v);
    }
}

// Backward of the trilinear head: forward sweep for the softmax statistics, backward sweep distributing
// gv[D] to the two depth planes and, when a plane index retires, to its four bilinear corners (float atomics).
template <int NH>
__global__ __launch_bounds__(256) void trilinear_softargmin_bwd(const float* __restrict__ c, long long hs,
                                                                const float* __restrict__ gout, float* __restrict__ gc,
                                                                int B, int Dl, int h, int w, int Do, int H, int W) {
    const long long HW = (long long)H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * HW) return;
    const int b = (int)(i / HW);
    const int r = (int)(i - b * HW);
    const int Y = r / W, X = r - Y * W;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(Y, (float)h / (float)H, h, y0, y1, ly);
    src_index(X, (float)w / (float)W, w, x0, x1, lx);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const size_t bbase = (size_t)b * Dl * h * w;
    auto plane = [&](int j, float (&G)[NH]) {
        float a[NH], bq[NH], cq[NH], d[NH];
        const size_t pj = bbase + (size_t)j * h * w;
        load_cum<NH>(c, hs, pj + (size_t)y0 * w + x0, a);
        load_cum<NH>(c, hs, pj + (size_t)y0 * w + x1, bq);
        load_cum<NH>(c, hs, pj + (size_t)y1 * w + x0, cq);
        load_cum<NH>(c, hs, pj + (size_t)y1 * w + x1, d);
#pragma unroll
        for (int k = 0; k < NH; ++k) G[k] = w00 * a[k] + w01 * bq[k] + w10 * cq[k] + w11 * d[k];
    };
    const float dscale = (float)Dl / (float)Do;
    float mx[NH], inv[NH], pred[NH], g[NH];
    {
        Online acc[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[k].init();
        float GA[NH], GB[NH];
        int jA = -1, jB = -1;
        for (int D = 0; D < Do; ++D) {
            int d0, d1; float ld;
            src_index(D, dscale, Dl, d0, d1, ld);
            if (d0 != jA) { plane(d0, GA); jA = d0; }
            if (d1 != jB) { plane(d1, GB); jB = d1; }
#pragma unroll
            for (int k = 0; k < NH; ++k) acc[k].push((1.f - ld) * GA[k] + ld * GB[k], (float)D);
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            mx[k] = acc[k].m; inv[k] = 1.f / acc[k].s; pred[k] = acc[k].t * inv[k];
            g[k] = gout[((size_t)k * B + b) * HW + r];
        }
    }
    // gG accumulators per LR depth plane; a plane retires once the sweep has moved past it
    auto retire = [&](int j, const float (&gG)[NH]) {
        float run = 0.f, suf[NH];
#pragma unroll
        for (int k = NH - 1; k >= 0; --k) { run += gG[k]; suf[k] = run; }
        const size_t pj = bbase + (size_t)j * h * w;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            float* gk = gc + (size_t)k * hs + pj;
            atomicAdd(gk + (size_t)y0 * w + x0, w00 * suf[k]);
            atomicAdd(gk + (size_t)y0 * w + x1, w01 * suf[k]);
            atomicAdd(gk + (size_t)y1 * w + x0, w10 * suf[k]);
            atomicAdd(gk + (size_t)y1 * w + x1, w11 * suf[k]);
        }
    };
    // lo = LR plane jlo, hi = plane jlo+1 (d0 is non-decreasing and d1 is d0 or d0+1)
    float Glo[NH], Ghi[NH], glo[NH], ghi[NH];
    int jlo = 0;
    plane(0, Glo);
    if (Dl > 1) plane(1, Ghi);
#pragma unroll
    for (int k = 0; k < NH; ++k) { glo[k] = ghi[k] = 0.f; if (Dl <= 1) Ghi[k] = Glo[k]; }
    for (int D = 0; D < Do; ++D) {
        int d0, d1; float ld;
        src_index(D, dscale, Dl, d0, d1, ld);
        while (d0 > jlo) {                                   // plane jlo is finished
            retire(jlo, glo);
            ++jlo;
#pragma unroll
            for (int k = 0; k < NH; ++k) { Glo[k] = Ghi[k]; glo[k] = ghi[k]; ghi[k] = 0.f; }
            if (jlo + 1 < Dl) plane(jlo + 1, Ghi);
        }
        const bool same = d1 == d0;
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            const float v = (1.f - ld) * Glo[k] + ld * (same ? Glo[k] : Ghi[k]);
            const float gv = expf(v - mx[k]) * inv[k] * ((float)D - pred[k]) * g[k];
            glo[k] = fmaf(gv, same ? 1.f : 1.f - ld, glo[k]);
            if (!same) ghi[k] = fmaf(gv, ld, ghi[k]);
        }
    }
    retire(jlo, glo);
    if (jlo + 1 < Dl) retire(jlo + 1, ghi);
}


}  // namespace

#define DISPATCH_NH(KERNEL, ...)                                                       \
    switch (nheads) {                                                                  \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                     \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                     \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                     \
        default: return ECM_EUNSUP;                                                    \
    }

extern "C" int ecm_volume_mapping_fwd(const float* c0, long long head_stride, const float* m5, const float* mt3,
                                      float* disp, int nheads, int B, int Dl, int h, int w, int s, void* stream) {
    ECM_CHECK_ARG(c0 && m5 && mt3 && disp && B > 0 && Dl > 0 && h > 0 && w > 0 && s > 0);
    const long long n = (long long)B * h * s * w * s;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(volume_mapping_fwd, grid, block, 0, ecm_stream(stream), c0, head_stride, m5, mt3, disp, B, Dl, h, w, s)
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_trilinear_softargmin_fwd(const float* c0, long long head_stride, float* disp, int nheads, int B, int Dl,
                                            int h, int w, int Do, int H, int W, void* stream) {
    ECM_CHECK_ARG(c0 && disp && B > 0 && Dl > 0 && h > 0 && w > 0 && Do > 0 && H > 0 && W > 0);
    const long long n = (long long)B * H * W;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(trilinear_softargmin_fwd, grid, block, 0, ecm_stream(stream), c0, head_stride, disp, B, Dl, h, w, Do, H, W)
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_volume_mapping_bwd(const float* c0, long long head_stride, const float* m5, const float* mt3,
                                      const float* gdisp, float* gc0, float* gm5, float* gmt3, int nheads, int B, int Dl,
                                      int h, int w, int s, void* stream) {
    ECM_CHECK_ARG(c0 && m5 && mt3 && gdisp && gc0 && gm5 && gmt3 && B > 0 && Dl > 0 && h > 0 && w > 0 && s > 0);
    if ((s & (s - 1)) != 0 || s > 64) return ECM_EUNSUP;          // cell lanes are reduced with xor-shuffles
    hipStream_t st = ecm_stream(stream);
    const int H = h * s, W = w * s, Dmax = Dl * s;
    hipError_t e = hipMemsetAsync(gc0, 0, (size_t)nheads * B * Dl * h * w * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(gmt3, 0, (size_t)B * 3 * H * W * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    const int xblocks = (W + 255) / 256;
    const size_t lds = (size_t)3 * (256 + Dmax) * sizeof(float);
    dim3 grid((unsigned)((long long)B * H * xblocks)), block(256);
    DISPATCH_NH(volume_mapping_bwd, grid, block, lds, st, c0, head_stride, m5, mt3, gdisp, gc0, gm5, gmt3, B, Dl, h, w, s,
                xblocks)
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_trilinear_softargmin_bwd(const float* c0, long long head_stride, const float* gdisp, float* gc0,
                                            int nheads, int B, int Dl, int h, int w, int Do, int H, int W, void* stream) {
    ECM_CHECK_ARG(c0 && gdisp && gc0 && B > 0 && Dl > 0 && h > 0 && w > 0 && Do > 0 && H > 0 && W > 0);
    hipStream_t st = ecm_stream(stream);
    hipError_t e = hipMemsetAsync(gc0, 0, (size_t)nheads * B * Dl * h * w * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    const long long n = (long long)B * H * W;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(trilinear_softargmin_bwd, grid, block, 0, st, c0, head_stride, gdisp, gc0, B, Dl, h, w, Do, H, W)
    return ECM_LAUNCH_RESULT();
}
