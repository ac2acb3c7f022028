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

__device__ __forceinline__ void wave_fence() {
    // LDS operations of ONE wave execute in issue order; this keeps the compiler from moving them across the point
    // (a later lane-shifted read must see an earlier store of the neighbouring lane)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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
// Backward of the volume-mapping head -- deterministic: every sum below is taken in a fixed order, there is no float
// atomic anywhere (round 2's version scattered with atomicAdd; 8 of the 9 architectures then had run-to-run noise in
// their gradients).
//
// Stage 1 (volume_mapping_bwd): a wave owns ONE image row and walks it in chunks of 64 consecutive X; a workgroup is four
// such rows.  Per pixel: one forward sweep (online softmax statistics), one backward sweep that recomputes v[D], forms
// gv[D] = p[D] (D - pred) g and
//   * accumulates gF[j] (rolling j-1, j, j+1): g_m5 stays in registers; the LR-logit gradient m5[n] * suf_k[j] is reduced
//     over the s lanes of a cell with xor-shuffles and stored, per row, into part[k,b,j,Y,n,cell] (plain stores);
//   * accumulates the three target-plane gradients along the X - D diagonal in the WAVE'S OWN LDS row [3][W]: at a given D
//     the 64 lanes hit 64 different addresses, and the wave walks D and the chunks in program order, so plain
//     read-modify-writes behind a wave-level fence give one fixed summation order; the finished row is stored once.
// Stage 2 (volume_mapping_bwd_gather): per LR voxel, gc = sum over the 5 neighbours n and the s rows of the source cell
// of part[...] in a fixed order.
template <int NH>
__global__ __launch_bounds__(256) void volume_mapping_bwd(const float* __restrict__ c, long long hs,
                                                          const float* __restrict__ m5p, const float* __restrict__ mt3p,
                                                          const float* __restrict__ gout, float* __restrict__ part,
                                                          float* __restrict__ gm5, float* __restrict__ gmt3, int B, int Dl,
                                                          int h, int w, int s) {
    extern __shared__ float lds[];                      // [4 waves][3][W]
    const int H = h * s, W = w * s;
    const long long HW = (long long)H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;          // b * H + Y
    if (row >= (long long)B * H) return;                             // whole waves only; no workgroup barrier below
    const int b = (int)(row / H), Y = (int)(row - (long long)b * H);
    float* T = lds + (size_t)wave * 3 * W;
    for (int e = lane; e < 3 * W; e += 64) T[e] = 0.f;
    wave_fence();
    const int cy = Y / s;
    const float* mt = mt3p + (size_t)b * 3 * HW + (size_t)Y * W;
    const size_t bbase = (size_t)b * Dl * h * w;
    const int dy[5] = {0, 0, 0, -1, 1}, dx[5] = {0, 1, -1, 0, 0};
    for (int X0 = 0; X0 < W; X0 += 64) {
        const int X = X0 + lane;
        const bool valid = X < W;
        const int Xc = valid ? X : W - 1;
        const int r = Y * W + Xc;
        const int cx = Xc / s;
        float m5[5], gm[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) { m5[n] = m5p[((size_t)b * 5 + n) * HW + r]; gm[n] = 0.f; }
        float Fp[NH], Fc[NH], Fn[NH];
        // ---- forward sweep: statistics ----------------------------------------------------------------------
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
        // ---- backward sweep ---------------------------------------------------------------------------------
        float gFp[NH], gFc[NH], gFn[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) { Fp[k] = 0.f; gFp[k] = gFc[k] = gFn[k] = 0.f; }
        fuse5<NH>(c, hs, bbase, 0, h, w, cy, cx, m5, Fc);
        // emit the finished gF of depth jj: g_m5 and this row's share of the LR-logit gradient
        auto emit = [&](int jj, const float (&gF)[NH]) {
            float suf[NH];                               // gradient w.r.t. the RAW head outputs: suffix sums over heads
            float run = 0.f;
#pragma unroll
            for (int k = NH - 1; k >= 0; --k) { run += gF[k]; suf[k] = run; }
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const int yy = cy + dy[n], xx = cx + dx[n];
                if (yy < 0 || yy >= h) continue;                              // wave-uniform
                const bool inb = xx >= 0 && xx < w;
                if (inb) {
                    float v[NH];
                    load_cum<NH>(c, hs, bbase + ((size_t)jj * h + yy) * w + xx, v);
#pragma unroll
                    for (int k = 0; k < NH; ++k) gm[n] = fmaf(gF[k], v[k], gm[n]);
                }
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                    float t = m5[n] * suf[k];
                    for (int off = 1; off < s; off <<= 1) t += __shfl_xor(t, off, 64);   // lanes of one cell, fixed tree
                    if ((lane & (s - 1)) == 0 && valid && inb)
                        part[(((((size_t)k * B + b) * Dl + jj) * H + Y) * 5 + n) * w + cx] = t;
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
                if (in && valid) {                       // 64 lanes, 64 different addresses X - D
                    const int li = Xc - D;
                    T[li] += a0;
                    T[W + li] += ar;
                    T[2 * W + li] += al;
                }
                wave_fence();                            // the next D's read-modify-write hits a neighbour lane's address
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
    }
    wave_fence();
    float* gt = gmt3 + (size_t)b * 3 * HW + (size_t)Y * W;
    for (int e = lane; e < 3 * W; e += 64) {
        const int pl = e / W, x = e - pl * W;
        gt[(size_t)pl * HW + x] = T[e];
    }
}

// Stage 2: gc[k,b,j,y,x] = sum_n sum_{rows Y of cell y - dy[n]} part[k,b,j,Y,n,x - dx[n]]   (fixed order: n, then Y)
__global__ __launch_bounds__(256) void volume_mapping_bwd_gather(const float* __restrict__ part, float* __restrict__ gc,
                                                                 long long hs, int NH, int B, int Dl, int h, int w, int s) {
    const long long per = (long long)B * Dl * h * w;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * NH) return;
    const int k = (int)(i / per);
    long long r = i - (long long)k * per;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h); r /= h;
    const int j = (int)(r % Dl);
    const int b = (int)(r / Dl);
    const int H = h * s;
    const int dy[5] = {0, 0, 0, -1, 1}, dx[5] = {0, 1, -1, 0, 0};
    float acc = 0.f;
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const int sy = y - dy[n], sx = x - dx[n];                     // the source cell whose neighbour n is (y, x)
        if (sy < 0 || sy >= h || sx < 0 || sx >= w) continue;
        const float* p = part + (((((size_t)k * B + b) * Dl + j) * H + (size_t)sy * s) * 5 + n) * w + sx;
        float a = 0.f;
        for (int q = 0; q < s; ++q) a += p[(size_t)q * 5 * w];
        acc += a;
    }
    gc[(size_t)k * hs + (((size_t)b * Dl + j) * h + y) * w + x] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the trilinear head -- deterministic, three stages, no atomics:
//   1 (trilinear_softargmin_bwd): per output pixel, forward sweep for the softmax statistics, backward sweep distributing
//     gv[D] to the two depth planes; when a plane index retires, the pixel's gradient w.r.t. the bilinear sample of that
//     plane (suffix-summed over heads) goes to S[k,b,j,Y,X] (plain coalesced stores);
//   2 (trilinear_bwd_reduce_x): Tm[k,b,j,Y,x] = sum_X wx(X -> x) S[..,Y,X]   over the output columns whose corners hit x;
//   3 (trilinear_bwd_reduce_y): gc[k,b,j,y,x] = sum_Y wy(Y -> y) Tm[..,Y,x].
template <int NH>
__global__ __launch_bounds__(256) void trilinear_softargmin_bwd(const float* __restrict__ c, long long hs,
                                                                const float* __restrict__ gout, float* __restrict__ S,
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
        float run = 0.f;
#pragma unroll
        for (int k = NH - 1; k >= 0; --k) {
            run += gG[k];
            S[(((size_t)k * B + b) * Dl + j) * HW + r] = run;
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
    float zero[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) zero[k] = 0.f;
    for (int j = jlo + 2; j < Dl; ++j) retire(j, zero);      // planes the output depths never sample (Do < Dl)
}

// output indices whose source cell along one axis can touch input index t: src = scale*(dst+0.5)-0.5 in [t-1, t+1)
__device__ __forceinline__ void dst_range(int t, float scale, int out_size, int& lo, int& hi) {
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)t - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)t + 1.5f) * inv - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

__global__ __launch_bounds__(256) void trilinear_bwd_reduce_x(const float* __restrict__ S, float* __restrict__ Tm, long long rows,
                                                              int w, int W) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // (row of S, x)
    if (i >= rows * w) return;
    const long long row = i / w;
    const int x = (int)(i - row * w);
    const float scale = (float)w / (float)W;
    int lo, hi;
    dst_range(x, scale, W, lo, hi);
    const float* sp = S + (size_t)row * W;
    float acc = 0.f;
    for (int X = lo; X <= hi; ++X) {
        int x0, x1; float lx;
        src_index(X, scale, w, x0, x1, lx);
        float wt = 0.f;
        if (x0 == x) wt += 1.f - lx;
        if (x1 == x) wt += lx;
        if (wt != 0.f) acc = fmaf(wt, sp[X], acc);
    }
    Tm[i] = acc;
}

__global__ __launch_bounds__(256) void trilinear_bwd_reduce_y(const float* __restrict__ Tm, float* __restrict__ gc, long long hs,
                                                              int NH, int B, int Dl, int h, int w, int H) {
    const long long per = (long long)B * Dl * h * w;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * NH) return;
    const int k = (int)(i / per);
    long long r = i - (long long)k * per;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h); r /= h;                     // r = b * Dl + j
    const float scale = (float)h / (float)H;
    int lo, hi;
    dst_range(y, scale, H, lo, hi);
    const float* tp = Tm + (((size_t)k * B * Dl + (size_t)r) * H) * w + x;
    float acc = 0.f;
    for (int Y = lo; Y <= hi; ++Y) {
        int y0, y1; float ly;
        src_index(Y, scale, h, y0, y1, ly);
        float wt = 0.f;
        if (y0 == y) wt += 1.f - ly;
        if (y1 == y) wt += ly;
        if (wt != 0.f) acc = fmaf(wt, tp[(size_t)Y * w], acc);
    }
    gc[(size_t)k * hs + ((size_t)r * h + y) * w + x] = acc;
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

extern "C" long long ecm_volume_mapping_bwd_scratch_bytes(int nheads, int B, int Dl, int h, int w, int s) {
    if (nheads <= 0 || B <= 0 || Dl <= 0 || h <= 0 || w <= 0 || s <= 0) return 0;
    return (long long)nheads * B * Dl * h * s * 5 * w * (long long)sizeof(float);
}

extern "C" int ecm_volume_mapping_bwd(const float* c0, long long head_stride, const float* m5, const float* mt3,
                                      const float* gdisp, float* gc0, float* gm5, float* gmt3, void* scratch,
                                      long long scratch_bytes, int nheads, int B, int Dl, int h, int w, int s, void* stream) {
    ECM_CHECK_ARG(c0 && m5 && mt3 && gdisp && gc0 && gm5 && gmt3 && scratch && B > 0 && Dl > 0 && h > 0 && w > 0 && s > 0);
    if ((s & (s - 1)) != 0 || s > 64) return ECM_EUNSUP;          // cell lanes are reduced with xor-shuffles inside a wave
    if (scratch_bytes < ecm_volume_mapping_bwd_scratch_bytes(nheads, B, Dl, h, w, s)) return ECM_ESCRATCH;
    hipStream_t st = ecm_stream(stream);
    const int H = h * s, W = w * s;
    const size_t lds = (size_t)4 * 3 * W * sizeof(float);         // one [3][W] row per wave
    if (lds > 160 * 1024) return ECM_EUNSUP;
    float* part = static_cast<float*>(scratch);
    dim3 grid((unsigned)(((long long)B * H + 3) / 4)), block(256);
    switch (nheads) {
#define ECM_VM_CASE(N)                                                                                                         \
        case N: {                                                                                                              \
            const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(volume_mapping_bwd<N>), (int)lds);               \
            if (e != hipSuccess) return (int)e;                                                                                \
            hipLaunchKernelGGL(volume_mapping_bwd<N>, grid, block, lds, st, c0, head_stride, m5, mt3, gdisp, part, gm5, gmt3,  \
                               B, Dl, h, w, s);                                                                                \
        } break;
        ECM_VM_CASE(1) ECM_VM_CASE(2) ECM_VM_CASE(3)
#undef ECM_VM_CASE
        default: return ECM_EUNSUP;
    }
    const long long n = (long long)nheads * B * Dl * h * w;
    hipLaunchKernelGGL(volume_mapping_bwd_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, gc0, head_stride,
                       nheads, B, Dl, h, w, s);
    return ECM_LAUNCH_RESULT();
}

extern "C" long long ecm_trilinear_softargmin_bwd_scratch_bytes(int nheads, int B, int Dl, int h, int w, int H, int W) {
    if (nheads <= 0 || B <= 0 || Dl <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return 0;
    return (long long)nheads * B * Dl * H * ((long long)W + w) * (long long)sizeof(float);
}

extern "C" int ecm_trilinear_softargmin_bwd(const float* c0, long long head_stride, const float* gdisp, float* gc0,
                                            void* scratch, long long scratch_bytes, int nheads, int B, int Dl, int h, int w,
                                            int Do, int H, int W, void* stream) {
    ECM_CHECK_ARG(c0 && gdisp && gc0 && scratch && B > 0 && Dl > 0 && h > 0 && w > 0 && Do > 0 && H > 0 && W > 0);
    if (scratch_bytes < ecm_trilinear_softargmin_bwd_scratch_bytes(nheads, B, Dl, h, w, H, W)) return ECM_ESCRATCH;
    hipStream_t st = ecm_stream(stream);
    float* S = static_cast<float*>(scratch);
    const long long rows = (long long)nheads * B * Dl * H;
    float* Tm = S + rows * W;
    const long long n = (long long)B * H * W;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    DISPATCH_NH(trilinear_softargmin_bwd, grid, block, 0, st, c0, head_stride, gdisp, S, B, Dl, h, w, Do, H, W)
    hipLaunchKernelGGL(trilinear_bwd_reduce_x, dim3((unsigned)((rows * w + 255) / 256)), dim3(256), 0, st, S, Tm, rows, w, W);
    const long long nv = (long long)nheads * B * Dl * h * w;
    hipLaunchKernelGGL(trilinear_bwd_reduce_y, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, Tm, gc0, head_stride, nheads,
                       B, Dl, h, w, H);
    return ECM_LAUNCH_RESULT();
}
