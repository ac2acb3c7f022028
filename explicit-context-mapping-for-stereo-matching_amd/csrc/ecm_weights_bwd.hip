// Backward of the context-mapping weight generators (autograd of eight_related_context_mapping.forward,
// cmfsm.py:443-593, and of six_related_context_mapping.forward, cmfsm_sub_8.py:449-572): gradients w.r.t. lr, hr and
// the 2,760 MLP weights from the gradient of the [B,N,H,W] planes.  Variants as in ecm_nbr.h; scale s in {4, 8, 16, ...}
// (s % 4 == 0).
//
// Kernel B (per HR pixel, one wave per 64 consecutive X of one row, workgroup = 4 rows inside one LR cell row):
//   softmax (or softmax*logit) backward -> g_logit[n]; per in-image neighbour recompute the MLP and back-propagate
//   to g0 = dL/d(h0pre).
//   * ghr         = W0_hr^T sum_n g0                        (per pixel, registers)
//   * gA9[rb,cell,n] = sum over the cell's pixels in this 4-row block of g0   (plain stores, no atomics)
//   * weight gradients are sums over PIXELS of per-pixel outer products, i.e. GEMMs whose reduction index is
//     the lane: each wave transposes the operands through a [64 px][48] LDS scratch and reduces them on the
//     fp32 matrix cores (v_mfma_f32_16x16x4_f32, K = 4 pixels per step); accumulators persist across tiles.
// Kernel C (per LR cell): gA = sum_n sum_rowblocks gA9[.., cell - d_n, n];  glr = W0_lr^T gA;  gW0_lr partials.
// Kernel D: fixed-order sum of the per-workgroup partials -> gW (deterministic).
#include "common.h"
#include "ecm_nbr.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CF = 32, TX = 64, TY = 4;
constexpr int ASTRIDE = 36;
constexpr int NCXMAX = TX / 4 + 2, NCY = 3;           // cell window of a tile (+1 halo), sized for the smallest scale
constexpr int UST = 48;                               // per-pixel stride of the transpose scratch (== 16 mod 32)
constexpr int PB_HR = 0, PB_OFF = 1024, PB_W1 = 1088, PB_W2 = 1600, PB_W3 = 1728, PB_N = 1736;   // partial layout

__device__ __forceinline__ float dleaky(float h) { return h > 0.f ? 1.f : 0.01f; }      // phi'(pre); sign(h)==sign(pre)

__device__ __forceinline__ void wave_lds_sync() {
    // LDS ops of one wave execute in issue order; this only stops the compiler from reordering across it.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A[b, cell, j] = sum_c W0[j, c] lr[b, c, cell]  (same as the forward's projection)
__global__ __launch_bounds__(256) void bwd_lr_proj(const float* __restrict__ lr, const float* __restrict__ W0,
                                                   float* __restrict__ A, float* __restrict__ W0hrT, int B, int hw) {
    if (blockIdx.x == 0)                                 // W0hrT[c][j] = W0[j][32 + c]: rows for the "row . vector" form of ghr
        for (int e = threadIdx.x; e < CF * CF; e += 256) W0hrT[e] = W0[(e % CF) * 66 + 32 + e / CF];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * hw) return;
    const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
    float v[CF];
#pragma unroll
    for (int c = 0; c < CF; ++c) v[c] = lr[((size_t)b * CF + c) * hw + p];
    float4* out = reinterpret_cast<float4*>(A + (size_t)i * CF);
#pragma unroll
    for (int j = 0; j < CF; j += 4) {
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CF; ++c) acc = fmaf(W0[(j + u) * 66 + c], v[c], acc);
            o[u] = acc;
        }
        out[j / 4] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// forward MLP of one neighbour: a = projected LR cell (LDS), Bv = W0_hr hr; returns the pre-final-activation output
__device__ __forceinline__ float mlp_forward(const float* __restrict__ a, const float (&Bv)[CF], float ox, float oy,
                                             const float* __restrict__ W0, const float* __restrict__ W1,
                                             const float* __restrict__ W2, const float* __restrict__ W3, float (&h0)[CF],
                                             float (&h1)[16], float (&h2)[8]) {
#pragma unroll
    for (int j = 0; j < CF; j += 4) {
        const float4 av = *reinterpret_cast<const float4*>(a + j);
        h0[j + 0] = leaky(fmaf(W0[(j + 0) * 66 + 65], oy, fmaf(W0[(j + 0) * 66 + 64], ox, av.x + Bv[j + 0])));
        h0[j + 1] = leaky(fmaf(W0[(j + 1) * 66 + 65], oy, fmaf(W0[(j + 1) * 66 + 64], ox, av.y + Bv[j + 1])));
        h0[j + 2] = leaky(fmaf(W0[(j + 2) * 66 + 65], oy, fmaf(W0[(j + 2) * 66 + 64], ox, av.z + Bv[j + 2])));
        h0[j + 3] = leaky(fmaf(W0[(j + 3) * 66 + 65], oy, fmaf(W0[(j + 3) * 66 + 64], ox, av.w + Bv[j + 3])));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < CF; ++j) acc = fmaf(W1[i * CF + j], h0[j], acc);
        h1[i] = leaky(acc);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = fmaf(W2[i * 16 + j], h1[j], acc);
        h2[i] = leaky(acc);
    }
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o = fmaf(W3[j], h2[j], o);
    return o;
}

// (The round-2 form of kernel B -- one wave per 64-pixel row, for any s % 4 == 0 -- was removed in round 4: it was the
// library's one kernel with a private segment, 260 B per lane = 65 spilled VGPRs, and no registered architecture has a
// scale other than 4, 8 or 16.  tools/check_private_segment.py now fails the build if any kernel spills.)

// Kernel C: per LR cell.  gA = sum_n sum_{row blocks of the source cell} gA9[rb, cell - d_n, n];
// glr[c] = sum_j W0[j][c] gA[j];  per-workgroup partial of gW0_lr[j][c] = sum_cells gA[j] lr[c].
template <int VAR>
__global__ __launch_bounds__(128) void ecm_weights_bwd_cells(const float* __restrict__ gA9, const float* __restrict__ lr,
                                                             const float* __restrict__ W0, float* __restrict__ glr,
                                                             float* __restrict__ partC, int B, int h, int w, int s) {
    using NB = Nbr<VAR>;
    constexpr int NN = NB::N;
    __shared__ float Gs[128 * 33];
    __shared__ float Ls[128 * 33];
    const int hw = h * w, rpc = s / TY, rbs = h * rpc;            // row blocks per cell / per image
    const long long i = (long long)blockIdx.x * 128 + threadIdx.x;
    const bool valid = i < (long long)B * hw;
    float gA[CF], lv[CF];
#pragma unroll
    for (int j = 0; j < CF; ++j) { gA[j] = 0.f; lv[j] = 0.f; }
    if (valid) {
        const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
        const int cy = p / w, cx = p - cy * w;
#pragma unroll 1
        for (int n = 0; n < NN; ++n) {
            const int sy = cy - NB::dy(n), sx = cx - NB::dx(n);
            if (sy < 0 || sy >= h || sx < 0 || sx >= w) continue;
            for (int q = 0; q < rpc; ++q) {
                const float4* src = reinterpret_cast<const float4*>(
                    gA9 + ((((size_t)b * rbs + sy * rpc + q) * w + sx) * NN + n) * CF);
#pragma unroll
                for (int u = 0; u < CF / 4; ++u) {
                    const float4 v = src[u];
                    gA[4 * u] += v.x; gA[4 * u + 1] += v.y; gA[4 * u + 2] += v.z; gA[4 * u + 3] += v.w;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CF; ++c) {
            lv[c] = lr[((size_t)b * CF + c) * hw + p];
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < CF; ++j) acc = fmaf(W0[j * 66 + c], gA[j], acc);
            glr[((size_t)b * CF + c) * hw + p] = acc;
        }
    }
#pragma unroll
    for (int j = 0; j < CF; ++j) { Gs[threadIdx.x * 33 + j] = gA[j]; Ls[threadIdx.x * 33 + j] = lv[j]; }
    __syncthreads();
    // thread owns gW0_lr[j][c0..c0+7]
    const int j = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 8;
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    for (int cell = 0; cell < 128; ++cell) {
        const float gv = Gs[cell * 33 + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = fmaf(gv, Ls[cell * 33 + c0 + u], a[u]);
    }
    float* pc = partC + (size_t)blockIdx.x * 1024 + j * 32 + c0;
#pragma unroll
    for (int u = 0; u < 8; ++u) pc[u] = a[u];
}

// Kernel D: gW = [gW0 (32x66) | gW1 (16x32) | gW2 (8x16) | gW3 (8)], fixed-order sums of the partials: a workgroup owns
// 32 outputs, its 8 lane groups each sum every 8th partial, and the 8 group sums are added in order (one thread per
// output walking up to ~1000 partials was latency-bound: 0.56 ms).
__global__ __launch_bounds__(256) void ecm_weights_bwd_reduce(const float* __restrict__ partB, int nB,
                                                              const float* __restrict__ partC, int nC,
                                                              float* __restrict__ gW) {
    __shared__ float sm[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + o;
    float s = 0.f;
    if (k < 2760) {
        const float* src;
        size_t stride;
        int n;
        if (k < 2112) {
            const int j = k / 66, c = k - j * 66;
            if (c < 32) { src = partC + j * 32 + c; stride = 1024; n = nC; }
            else if (c < 64) { src = partB + PB_HR + j * 32 + (c - 32); stride = PB_N; n = nB; }
            else { src = partB + PB_OFF + j * 2 + (c - 64); stride = PB_N; n = nB; }
        } else if (k < 2624) { src = partB + PB_W1 + (k - 2112); stride = PB_N; n = nB; }
        else if (k < 2752) { src = partB + PB_W2 + (k - 2624); stride = PB_N; n = nB; }
        else { src = partB + PB_W3 + (k - 2752); stride = PB_N; n = nB; }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = grp;
        for (; p + 24 < n; p += 32) {
            s0 += src[(size_t)p * stride];
            s1 += src[(size_t)(p + 8) * stride];
            s2 += src[(size_t)(p + 16) * stride];
            s3 += src[(size_t)(p + 24) * stride];
        }
        for (; p < n; p += 8) s0 += src[(size_t)p * stride];
        s = (s0 + s1) + (s2 + s3);
    }
    sm[grp][o] = s;
    __syncthreads();
    if (grp == 0 && k < 2760) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sm[g][o];
        gW[k] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel B for s in {4, 8, 16} (every registered architecture; other scales are ECM_EUNSUP).  Same mathematics and partial /
// gA9 layouts as the round-2 kernel it replaced, restructured around two measured defects of that one:
//   * the compiler hoisted all 2,760 loop-invariant weight loads out of the neighbour loop and then spilled ~1,800 SGPRs
//     into VGPR lanes -- 3,500 v_readlane (+ 500 s_nop for the hazards) against 2,200 FMA instructions in the kernel, and
//     65 VGPRs in scratch memory on top.  Here every weight matrix is reached through a pointer the compiler cannot see
//     through (OPAQUE), re-made per use, so a weight is loaded by the scalar unit right where it is consumed;
//   * the cell sums of g0 crossed the four waves of a workgroup (two workgroup barriers per neighbour).  A wave now owns a
//     4 x 16 pixel patch instead of one 64-pixel row, so a cell's pixels (4 rows x s columns of the patch) sit in ONE
//     wave and the sums need no barrier, no second LDS array and no private scratch memory at all.
// The order of the three matrix-core reductions follows the lifetime of the operands: (h1, g2) first, then (h0, g1), then
// (g0, offsets), each operand dead right after its reduction.
// an offset of zero the compiler cannot see through; added to a kernel-argument pointer it keeps the pointer's provenance
// (global, read-only, uniform: scalar loads) but not its loop invariance
__device__ __forceinline__ int opaque_zero() { int z = 0; asm volatile("" : "+s"(z)); return z; }
#define OPAQUE(ptr) ptr += opaque_zero()

// ---- weights through hand-issued scalar loads ------------------------------------------------------------------------------
// Every dense product of the MLP is either "row . vector" or "row * scalar -> vector"; a row of 8 / 16 / 32 weights is fetched
// by s_load_dwordx8/x16 into an SGPR block and consumed as the scalar operand of packed FMAs, the next row's load in flight
// under the current row's arithmetic.  The compiler treats an asm output as valid at once, so every consumer reads the block
// through the "+s" operand of the s_waitcnt that retires it (SMEM returns out of order: only lgkmcnt(0) is meaningful).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// `off` (floats) must fold to a constant after inlining / unrolling: it becomes the instruction's immediate offset, so that one
// base pointer per matrix serves every chunk (per-chunk pointers are loop-invariant: the compiler precomputes all ~200 of
// them ahead of the neighbour loop and spills them, two v_readlane per load -- measured)
__device__ __forceinline__ f32x16 sload16(const float* p, int off = 0) {
    f32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "n"(off * 4));
    return v;
}
// the same load, ordered behind the instructions that produced `dep` (a VGPR value): volatile asm statements keep their order
// among themselves, but the FMAs between them are free to sink below later loads -- whose SGPR blocks then all stay live and
// spill (measured: 1,300 spilled SGPRs).  Passing the accumulator state pins "FMAs of chunk q-1, then load of chunk q+1".
__device__ __forceinline__ f32x16 sload16_after(const float* p, int off, const f32x2& dep) {
    f32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(p), "n"(off * 4), "v"(dep));
    return v;
}
__device__ __forceinline__ f32x8 sload8(const float* p) {
    f32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(v) : "s"(p));
    return v;
}
__device__ __forceinline__ f32x2 sload2(const float* p, int off) {
    f32x2 v;
    asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(v) : "s"(p), "n"(off * 4));
    return v;
}
__device__ __forceinline__ void swait(f32x16& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
__device__ __forceinline__ void swait(f32x8& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
// sum_j (w[2j], w[2j+1]) * x[j]   (8 packed FMAs; even / odd inputs accumulate separately)
// (two independent chains: back-to-back DEPENDENT packed FMAs cost a wait state each on gfx950)
__device__ __forceinline__ f32x2 dot16(const f32x16& w, const f32x2* x, f32x2 acc) {
    f32x2 acc1 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 w0 = {w[2 * j], w[2 * j + 1]}, w1 = {w[2 * j + 2], w[2 * j + 3]};
        acc = w0 * x[j] + acc;
        acc1 = w1 * x[j + 1] + acc1;
    }
    return acc + acc1;
}
// acc[j] += (w[2j], w[2j+1]) * g
__device__ __forceinline__ void axpy16(const f32x16& w, float g, f32x2* acc) {
    const f32x2 gg = {g, g};
#pragma unroll
    for (int j = 0; j < 8; ++j) { const f32x2 ww = {w[2 * j], w[2 * j + 1]}; acc[j] = ww * gg + acc[j]; }
}
__device__ __forceinline__ f32x2 leaky2(f32x2 v) {
    const f32x2 t = v * 0.01f;
    return {fmaxf(v.x, t.x), fmaxf(v.y, t.y)};                           // max(x, 0.01 x) == LeakyReLU(0.01)
}
__device__ __forceinline__ float dmask(float gv, float hval) { return hval > 0.f ? gv : 0.01f * gv; }   // gv * phi'(pre)

__device__ __forceinline__ void swait8(f32x2 (&w)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w[0]), "+s"(w[1]), "+s"(w[2]), "+s"(w[3]), "+s"(w[4]), "+s"(w[5]), "+s"(w[6]), "+s"(w[7]));
}
// All four helpers stream the matrix as 16-float chunks with ONE chunk in flight under the current chunk's arithmetic
// (32 SGPRs in all: the kernel's pointers and sizes leave room for little more).
// y[i] = W[i, 0:32] . x for NOUT rows STRIDE floats apart (outputs as (even, odd) pairs)
template <int NOUT, int STRIDE, int BASE>
__device__ __forceinline__ void dense_rows(const float* W, const f32x2 (&x)[16], f32x2 (&y)[NOUT / 2]) {
    f32x16 a = sload16(W, BASE);
    swait(a);
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2 * NOUT; ++q) {
        const int i = q >> 1, half = q & 1;
        f32x16 nx = a;
        if (q + 1 < 2 * NOUT) nx = sload16_after(W, BASE + ((q + 1) >> 1) * STRIDE + ((q + 1) & 1) * 16, acc);
        acc = dot16(a, x + half * 8, half ? acc : f32x2{0.f, 0.f});
        if (half) {
            const float v = acc.x + acc.y;
            if (i & 1) y[i / 2].y = v; else y[i / 2].x = v;
        }
        if (q + 1 < 2 * NOUT) { swait(nx); a = nx; }
    }
}
// y[i] = W[i, 0:16] . x, rows 16 floats apart
template <int NOUT>
__device__ __forceinline__ void dense16_rows(const float* W, const f32x2 (&x)[8], f32x2 (&y)[NOUT / 2]) {
    f32x16 a = sload16(W);
    swait(a);
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        f32x16 nx = a;
        if (i + 1 < NOUT) nx = sload16_after(W, (i + 1) * 16, acc);
        acc = dot16(a, x, f32x2{0.f, 0.f});
        const float v = acc.x + acc.y;
        if (i & 1) y[i / 2].y = v; else y[i / 2].x = v;
        if (i + 1 < NOUT) { swait(nx); a = nx; }
    }
}
// acc[0:16] += W[i, 0:16] * g[i] over NROW rows 16 floats apart (g as pairs)
template <int NROW>
__device__ __forceinline__ void axpy16_rows(const float* W, const f32x2 (&g)[NROW / 2], f32x2 (&acc)[8]) {
    f32x16 a = sload16(W);
    swait(a);
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        f32x16 nx = a;
        if (i + 1 < NROW) nx = sload16_after(W, (i + 1) * 16, acc[7]);
        axpy16(a, (i & 1) ? g[i / 2].y : g[i / 2].x, acc);
        if (i + 1 < NROW) { swait(nx); a = nx; }
    }
}
// acc[0:32] += W[i, 0:32] * g[i] over NROW rows STRIDE floats apart
template <int NROW, int STRIDE, int BASE>
__device__ __forceinline__ void axpy32_rows(const float* W, const f32x2 (&g)[NROW / 2], f32x2 (&acc)[16]) {
    f32x16 a = sload16(W, BASE);
    swait(a);
#pragma unroll
    for (int q = 0; q < 2 * NROW; ++q) {
        const int i = q >> 1, half = q & 1;
        f32x16 nx = a;
        if (q + 1 < 2 * NROW) nx = sload16_after(W, BASE + ((q + 1) >> 1) * STRIDE + ((q + 1) & 1) * 16, acc[(half ? 7 : 15)]);
        axpy16(a, (i & 1) ? g[i / 2].y : g[i / 2].x, acc + half * 8);
        if (q + 1 < 2 * NROW) { swait(nx); a = nx; }
    }
}

template <int VAR>
__global__ __launch_bounds__(256, 2) void ecm_weights_bwd_kernel_p(
    const float* __restrict__ A, const float* __restrict__ hr, const float* __restrict__ W0g, const float* __restrict__ W1g,
    const float* __restrict__ W2g, const float* __restrict__ W3g, const float* __restrict__ wsaved,
    const float* __restrict__ gw, float* __restrict__ ghr, float* __restrict__ gA9, float* __restrict__ partB,
    const float* __restrict__ W0hrT, int B, int h, int w, int s, int tiles_x) {
    using NB = Nbr<VAR>;
    constexpr int NN = NB::N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [NCY*ncx][ASTRIDE]
    float* Uall = As + NCY * NCXMAX * ASTRIDE;          // [4 waves][64][UST]
    const int H = h * s, W = w * s;
    const size_t HW = (size_t)H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    float* U = Uall + wave * 64 * UST;
    const int ncx = TX / s + 2, rbs = H / TY;
    const int lpc = 4 * s;                               // lanes per cell inside a wave's 4 x 16 patch (16, 32, 64)

    f32x4 accW1[2], accW2, accOff[2], accHr[2][2];
    float accW3[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accW1[0][i] = accW1[1][i] = accW2[i] = accOff[0][i] = accOff[1][i] = 0.f;
        accHr[0][0][i] = accHr[0][1][i] = accHr[1][0][i] = accHr[1][1][i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) accW3[i] = 0.f;

    const long long ntiles = (long long)B * rbs * tiles_x;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x);
        const int rb = (int)((tile / tiles_x) % rbs);
        const int b = (int)(tile / ((long long)tiles_x * rbs));
        const int X0 = tx * TX, Y0 = rb * TY;
        const int cy = Y0 / s;                            // workgroup-uniform: s % 4 == 0
        const int cx0 = X0 / s - 1, cy0 = cy - 1;
        __syncthreads();
        for (int e = tid; e < NCY * ncx * (CF / 4); e += 256) {
            const int q = e % (CF / 4), cell = e / (CF / 4);
            const int yy = cy0 + cell / ncx, xx = cx0 + cell % ncx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < h && xx >= 0 && xx < w)
                v = reinterpret_cast<const float4*>(A + (((size_t)b * h + yy) * w + xx) * CF)[q];
            *reinterpret_cast<float4*>(As + cell * ASTRIDE + q * 4) = v;
        }
        __syncthreads();
        const int Y = Y0 + l4, X = X0 + wave * 16 + l15;  // lane = (row l4, column l15) of the wave's 4 x 16 patch
        const bool valid = X < W;                         // right-edge partial tiles: dead lanes carry zeros
        const int Xc = valid ? X : W - 1;
        const size_t pix = (size_t)Y * W + Xc;
        // hr / saved planes / gradient planes / ghr through buffer descriptors: ONE 32-bit lane offset (the pixel) and the
        // channel as the instruction's scalar offset.  Plain pointers made the compiler keep a 64-bit address per channel
        // alive from the loads at the head of the tile to the re-loads at its tail (~64 registers): that, not the neighbour
        // loop, was what spilled 50 VGPRs to private memory in round 3.
        const unsigned HWb = (unsigned)HW * 4u, pixb = (unsigned)pix * 4u;
        const auto hr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hr + (size_t)b * CF * HW), 0, (unsigned)CF * HWb, 0x00020000);
        const auto sv_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsaved + (size_t)b * NN * HW), 0, (unsigned)NN * HWb, 0x00020000);
        const auto gw_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gw + (size_t)b * NN * HW), 0, (unsigned)NN * HWb, 0x00020000);
        auto ldb = [](decltype(hr_rs) rs, unsigned voff, unsigned soff) {
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
        };
        const int cx = Xc / s, ry = Y - cy * s, rx = Xc - cx * s;

        f32x2 Bv[CF / 2];                                // Bv = W0_hr hv, as (even, odd) channel pairs
        {
            f32x2 hv[CF / 2];
#pragma unroll
            for (int c = 0; c < CF / 2; ++c) { hv[c].x = ldb(hr_rs, pixb, (unsigned)(2 * c) * HWb); hv[c].y = ldb(hr_rs, pixb, (unsigned)(2 * c + 1) * HWb); }
            dense_rows<CF, 66, 32>(W0g, hv, Bv);
        }
        // gradient w.r.t. the logits
        float gl[NN];
        if (!NB::TIMES_LOGIT) {
            float wv[NN], dot = 0.f;
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                wv[n] = ldb(sv_rs, pixb, (unsigned)n * HWb);
                gl[n] = ldb(gw_rs, pixb, (unsigned)n * HWb);
                dot = fmaf(wv[n], gl[n], dot);
            }
#pragma unroll
            for (int n = 0; n < NN; ++n) gl[n] = valid ? wv[n] * (gl[n] - dot) : 0.f;
        } else {
            float lg[NN];
#pragma unroll 1
            for (int n = 0; n < NN; ++n) {
                const int yy = cy + NB::dy(n), xx = cx + NB::dx(n);
                float v = NB::PAD;
                if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                    const float *W0 = W0g, *W1 = W1g, *W2 = W2g, *W3 = W3g;
                    OPAQUE(W0); OPAQUE(W1); OPAQUE(W2); OPAQUE(W3);
                    float h0[CF], h1[16], h2[8], Bf[CF];
#pragma unroll
                    for (int j = 0; j < CF / 2; ++j) { Bf[2 * j] = Bv[j].x; Bf[2 * j + 1] = Bv[j].y; }
                    const float* a = As + ((yy - cy0) * ncx + (xx - cx0)) * ASTRIDE;
                    v = mlp_forward(a, Bf, ecm_off_x(NB::tab(n), rx, s), ecm_off_y(NB::tab(n), ry, s), W0, W1, W2, W3, h0, h1, h2);
                    if (NB::FINAL_ACT) v = leaky(v);
                }
                lg[n] = v;
            }
            float m = lg[0];
#pragma unroll
            for (int n = 1; n < NN; ++n) m = fmaxf(m, lg[n]);
            float p[NN], sum = 0.f;
#pragma unroll
            for (int n = 0; n < NN; ++n) { p[n] = expf(lg[n] - m); sum += p[n]; }
            const float inv = 1.f / sum;
            float dot = 0.f;
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                p[n] *= inv;
                gl[n] = ldb(gw_rs, pixb, (unsigned)n * HWb);
                dot = fmaf(gl[n], p[n] * lg[n], dot);
            }
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                float g = p[n] * (gl[n] * (lg[n] + 1.f) - dot);
                if (NB::FINAL_ACT) g *= dleaky(lg[n]);
                gl[n] = valid ? g : 0.f;
            }
        }
        f32x2 gBv[CF / 2];
#pragma unroll
        for (int j = 0; j < CF / 2; ++j) gBv[j] = f32x2{0.f, 0.f};

#pragma unroll 1
        for (int n = 0; n < NN; ++n) {
            const int yy = cy + NB::dy(n);
            if (yy < 0 || yy >= h) continue;              // workgroup-uniform (the 4 rows share one LR cell row)
            const int xx = cx + NB::dx(n);
            const bool inb = xx >= 0 && xx < w;
            const float g = inb ? gl[n] : 0.f;            // out-of-image neighbours carry a constant: no gradient
            const int tab = NB::tab(n);
            const float ox = ecm_off_x(tab, rx, s), oy = ecm_off_y(tab, ry, s);
            const int xcl = min(max(xx, cx0), cx0 + ncx - 1);
            const float* a = As + ((yy - cy0) * ncx + (xcl - cx0)) * ASTRIDE;
            // ---- forward recompute (weights through hand-issued scalar loads, see sload16) ---------------------
            f32x2 h0[CF / 2], h1[8], h2[4];
#pragma unroll
            for (int jb = 0; jb < CF; jb += 8) {          // h0 = phi(a + Bv + W0[:,64] ox + W0[:,65] oy)
                f32x2 wo[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wo[u] = sload2(W0g, (jb + u) * 66 + 64);
                swait8(wo);
                const float4 av0 = *reinterpret_cast<const float4*>(a + jb), av1 = *reinterpret_cast<const float4*>(a + jb + 4);
                const float av[8] = {av0.x, av0.y, av0.z, av0.w, av1.x, av1.y, av1.z, av1.w};
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    f32x2 v;
                    v.x = fmaf(wo[u].y, oy, fmaf(wo[u].x, ox, av[u] + Bv[(jb + u) / 2].x));
                    v.y = fmaf(wo[u + 1].y, oy, fmaf(wo[u + 1].x, ox, av[u + 1] + Bv[(jb + u) / 2].y));
                    h0[(jb + u) / 2] = leaky2(v);
                }
            }
            dense_rows<16, 32, 0>(W1g, h0, h1);
#pragma unroll
            for (int i = 0; i < 8; ++i) h1[i] = leaky2(h1[i]);
            dense16_rows<8>(W2g, h1, h2);
#pragma unroll
            for (int i = 0; i < 4; ++i) h2[i] = leaky2(h2[i]);
            // ---- backward chain -----------------------------------------------------------------------
            f32x2 g2[4];
            {
                f32x8 w3 = sload8(W3g);
                swait(w3);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    accW3[2 * i] = fmaf(g, h2[i].x, accW3[2 * i]);
                    accW3[2 * i + 1] = fmaf(g, h2[i].y, accW3[2 * i + 1]);
                    g2[i].x = dmask(w3[2 * i] * g, h2[i].x);
                    g2[i].y = dmask(w3[2 * i + 1] * g, h2[i].y);
                }
            }
            // (3) gW2 += g2^T h1 over the wave's 64 pixels
#pragma unroll
            for (int j = 0; j < 8; j += 2) *reinterpret_cast<float4*>(U + lane * UST + 2 * j) = make_float4(h1[j].x, h1[j].y, h1[j + 1].x, h1[j + 1].y);
#pragma unroll
            for (int j = 0; j < 4; j += 2) *reinterpret_cast<float4*>(U + lane * UST + 16 + 2 * j) = make_float4(g2[j].x, g2[j].y, g2[j + 1].x, g2[j + 1].y);
            wave_lds_sync();
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const float* up = U + (k0 + l4) * UST;
                const float av = l15 < 8 ? up[16 + l15] : 0.f;
                accW2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, up[l15], accW2, 0, 0, 0);
            }
            wave_lds_sync();
            f32x2 g1[8];                                  // g1 = (W2^T g2) * phi'(h1)
#pragma unroll
            for (int j = 0; j < 8; ++j) g1[j] = f32x2{0.f, 0.f};
            axpy16_rows<8>(W2g, g2, g1);
#pragma unroll
            for (int j = 0; j < 8; ++j) { g1[j].x = dmask(g1[j].x, h1[j].x); g1[j].y = dmask(g1[j].y, h1[j].y); }
            // (1) gW1 += g1^T h0
#pragma unroll
            for (int j = 0; j < 16; j += 2) *reinterpret_cast<float4*>(U + lane * UST + 2 * j) = make_float4(h0[j].x, h0[j].y, h0[j + 1].x, h0[j + 1].y);
#pragma unroll
            for (int j = 0; j < 8; j += 2) *reinterpret_cast<float4*>(U + lane * UST + 32 + 2 * j) = make_float4(g1[j].x, g1[j].y, g1[j + 1].x, g1[j + 1].y);
            wave_lds_sync();
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const float* up = U + (k0 + l4) * UST;
                const float av = up[32 + l15];
                accW1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, up[l15], accW1[0], 0, 0, 0);
                accW1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, up[16 + l15], accW1[1], 0, 0, 0);
            }
            wave_lds_sync();
            {   // g0 = (W1^T g1) * phi'(h0)   (overwrites h0);  gBv += g0
                f32x2 acc[CF / 2];
#pragma unroll
                for (int c = 0; c < CF / 2; ++c) acc[c] = f32x2{0.f, 0.f};
                axpy32_rows<16, 32, 0>(W1g, g1, acc);
#pragma unroll
                for (int c = 0; c < CF / 2; ++c) {
                    h0[c].x = dmask(acc[c].x, h0[c].x);
                    h0[c].y = dmask(acc[c].y, h0[c].y);
                    gBv[c] += h0[c];
                }
            }
            // (2) gW0[:,64:66] += g0^T [ox,oy];  cell sums of g0 -> gA9
#pragma unroll
            for (int j = 0; j < 16; j += 2) *reinterpret_cast<float4*>(U + lane * UST + 2 * j) = make_float4(h0[j].x, h0[j].y, h0[j + 1].x, h0[j + 1].y);
            *reinterpret_cast<float2*>(U + lane * UST + 32) = make_float2(ox, oy);
            wave_lds_sync();
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const float* up = U + (k0 + l4) * UST;
                const float bvv = l15 < 2 ? up[32 + l15] : 0.f;
                accOff[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(up[l15], bvv, accOff[0], 0, 0, 0);
                accOff[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(up[16 + l15], bvv, accOff[1], 0, 0, 0);
            }
            {   // the cell's 4 x s pixels are lanes {r * 16 + c0 + x}: lane -> (cell, channel(s)), fixed summation order
                const int cell = lane / lpc, within = lane - cell * lpc;     // 16/s cells in the patch, lpc = 4s lanes each
                const int c0 = cell * s;
                const int cxs = (X0 + wave * 16) / s + cell;
                float* dst = gA9 + ((((size_t)b * rbs + rb) * w + cxs) * NN + n) * CF;
                if (lpc == 16) {                                             // s = 4: two channels per lane
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const float2 v = *reinterpret_cast<const float2*>(U + (r * 16 + c0 + x) * UST + within * 2);
                            s0 += v.x; s1 += v.y;
                        }
                    if (cxs < w) *reinterpret_cast<float2*>(dst + within * 2) = make_float2(s0, s1);
                } else if (lpc == 32) {                                      // s = 8: one channel per lane
                    float s0 = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int x = 0; x < 8; ++x) s0 += U[(r * 16 + c0 + x) * UST + within];
                    if (cxs < w) dst[within] = s0;
                } else {                                                     // s = 16: a channel's 64 pixels over two lanes
                    const int ch = within & 31, half = within >> 5;
                    float s0 = 0.f;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int x = 0; x < 16; ++x) s0 += U[((half * 2 + r) * 16 + x) * UST + ch];
                    s0 += __shfl_xor(s0, 32, 64);                            // rows 0-1 + rows 2-3: commutative, same bits
                    if (half == 0 && cxs < w) dst[ch] = s0;
                }
            }
            wave_lds_sync();
        }
        // ---- per pixel: ghr = W0_hr^T gBv ; gW0_hr += gBv^T hv ------------------------------------------
        {
            f32x2 acc[CF / 2];                            // ghr = W0_hr^T gBv, rows of the transposed copy
            dense_rows<CF, CF, 0>(W0hrT, gBv, acc);
            const auto gh_rs = __builtin_amdgcn_make_buffer_rsrc(ghr + (size_t)b * CF * HW, 0, (unsigned)CF * HWb, 0x00020000);
            const unsigned so = valid ? pixb : 0x80000000u;          // dead lanes: out of range, the store is dropped
#pragma unroll
            for (int c = 0; c < CF / 2; ++c) {
                // (through scalar copies: __builtin_bit_cast applied to an ext_vector ELEMENT lvalue, `acc[c].y`, was lowered by
                //  this hipcc to element 0 -- both stores of a pair wrote acc[c].x; caught by the g2 fixture)
                const float ax = acc[c].x, ay = acc[c].y;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ax), gh_rs, (int)so, (int)((unsigned)(2 * c) * HWb), 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ay), gh_rs, (int)so, (int)((unsigned)(2 * c + 1) * HWb), 0);
            }
        }
        float hv[CF];
#pragma unroll
        for (int c = 0; c < CF; ++c) hv[c] = ldb(hr_rs, valid ? pixb : 0x80000000u, (unsigned)c * HWb);     // dead lanes read 0
#pragma unroll
        for (int j = 0; j < 16; j += 2) *reinterpret_cast<float4*>(U + lane * UST + 2 * j) = make_float4(gBv[j].x, gBv[j].y, gBv[j + 1].x, gBv[j + 1].y);
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
#pragma unroll
            for (int j = 0; j < 16; j += 4)
                *reinterpret_cast<float4*>(U + lane * UST + 32 + j) = make_float4(hv[hs * 16 + j], hv[hs * 16 + j + 1], hv[hs * 16 + j + 2], hv[hs * 16 + j + 3]);
            wave_lds_sync();
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const float* up = U + (k0 + l4) * UST;
                const float bvv = up[32 + l15];
                accHr[0][hs] = __builtin_amdgcn_mfma_f32_16x16x4f32(up[l15], bvv, accHr[0][hs], 0, 0, 0);
                accHr[1][hs] = __builtin_amdgcn_mfma_f32_16x16x4f32(up[16 + l15], bvv, accHr[1][hs], 0, 0, 0);
            }
            wave_lds_sync();
        }
    }

    // ---- workgroup partial: sum the 4 waves' accumulators in LDS, then one store per entry -----------------
    __syncthreads();
    float* P = Uall;                                     // [4 waves][PB_N] (fits: 4*1736 < 4*64*48)
    float* pw = P + wave * PB_N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = l4 * 4 + r;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) pw[PB_HR + (mt * 16 + row) * 32 + hs * 16 + l15] = accHr[mt][hs][r];
            if (l15 < 2) pw[PB_OFF + (mt * 16 + row) * 2 + l15] = accOff[mt][r];
        }
        pw[PB_W1 + row * 32 + l15] = accW1[0][r];
        pw[PB_W1 + row * 32 + 16 + l15] = accW1[1][r];
        if (row < 8) pw[PB_W2 + row * 16 + l15] = accW2[r];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = wave_sum(accW3[i]);
        if (lane == 0) pw[PB_W3 + i] = v;
    }
    __syncthreads();
    for (int e = tid; e < PB_N; e += 256)
        partB[(size_t)blockIdx.x * PB_N + e] = (P[e] + P[PB_N + e]) + (P[2 * PB_N + e] + P[3 * PB_N + e]);
}

struct BwdPlan { long long ntiles; int tiles_x, nB, nC; long long offA, offA9, offPB, offPC, offWT, total; };

inline BwdPlan plan(int B, int h, int w, int s, int nn) {
    BwdPlan p;
    p.tiles_x = (w * s + TX - 1) / TX;
    const long long rbs = (long long)h * s / TY;
    p.ntiles = (long long)B * rbs * p.tiles_x;
    p.nB = (int)(p.ntiles < 512 ? p.ntiles : 512);
    p.nC = (int)(((long long)B * h * w + 127) / 128);
    const long long cells = (long long)B * h * w;
    p.offA = 0;
    p.offA9 = p.offA + cells * CF;
    p.offPB = p.offA9 + (long long)B * rbs * w * nn * CF;
    p.offPC = p.offPB + (long long)p.nB * PB_N;
    p.offWT = p.offPC + (long long)p.nC * 1024;
    p.total = p.offWT + CF * CF;
    return p;
}

constexpr int BWD_LDS_BYTES_P = (NCY * NCXMAX * ASTRIDE + 4 * 64 * UST) * 4;

template <int VAR>
int launch_bwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2, const float* W3,
               const float* saved, const float* gout, float* glr, float* ghr, float* gW, float* base, const BwdPlan& p,
               int B, int h, int w, int s, hipStream_t st) {
    float *A = base + p.offA, *gA9 = base + p.offA9, *partB = base + p.offPB, *partC = base + p.offPC;
    const long long cells = (long long)B * h * w;
    float* W0hrT = base + p.offWT;
    hipLaunchKernelGGL(bwd_lr_proj, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, lr, W0, A, W0hrT, B, h * w);
    {                                                    // s in {4, 8, 16}: a cell's pixels inside one wave's 4 x 16 patch
        const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(ecm_weights_bwd_kernel_p<VAR>), BWD_LDS_BYTES_P);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(ecm_weights_bwd_kernel_p<VAR>, dim3(p.nB), dim3(256), BWD_LDS_BYTES_P, st, A, hr, W0, W1, W2, W3, saved,
                           gout, ghr, gA9, partB, W0hrT, B, h, w, s, p.tiles_x);
    }
    hipLaunchKernelGGL(ecm_weights_bwd_cells<VAR>, dim3(p.nC), dim3(128), 0, st, gA9, lr, W0, glr, partC, B, h, w, s);
    hipLaunchKernelGGL(ecm_weights_bwd_reduce, dim3((2760 + 31) / 32), dim3(256), 0, st, partB, p.nB, partC, p.nC, gW);
    return ECM_LAUNCH_RESULT();
}

inline int planes_of(int variant) { return variant == 0 ? 9 : variant == 1 ? 5 : 3; }

}  // namespace

extern "C" long long ecm_context_weights_bwd_scratch_bytes(int B, int h, int w, int s, int variant) {
    if (B <= 0 || h <= 0 || w <= 0 || (s != 4 && s != 8 && s != 16) || variant < 0 || variant > 2 || (variant == 0 && s != 4)) return 0;
    return plan(B, h, w, s, planes_of(variant)).total * (long long)sizeof(float);
}

extern "C" int ecm_context_weights_bwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                                       const float* W3, const float* out_saved, const float* gout, float* glr, float* ghr,
                                       float* gW, void* scratch, long long scratch_bytes, int B, int h, int w, int s,
                                       int variant, void* stream) {
    ECM_CHECK_ARG(lr && hr && W0 && W1 && W2 && W3 && out_saved && gout && glr && ghr && gW && scratch && B > 0 && h > 0 && w > 0);
    if ((s != 4 && s != 8 && s != 16) || variant < 0 || variant > 2) return ECM_EUNSUP;    // the registered architectures' scales
    if (variant == 0 && s != 4) return ECM_EUNSUP;                                         // see ecm_context_weights_fwd
    const BwdPlan p = plan(B, h, w, s, planes_of(variant));
    if (scratch_bytes < p.total * (long long)sizeof(float)) return ECM_ESCRATCH;
    float* base = static_cast<float*>(scratch);
    hipStream_t st = ecm_stream(stream);
    if (variant == 0) return launch_bwd<0>(lr, hr, W0, W1, W2, W3, out_saved, gout, glr, ghr, gW, base, p, B, h, w, s, st);
    if (variant == 1) return launch_bwd<1>(lr, hr, W0, W1, W2, W3, out_saved, gout, glr, ghr, gW, base, p, B, h, w, s, st);
    return launch_bwd<2>(lr, hr, W0, W1, W2, W3, out_saved, gout, glr, ghr, gW, base, p, B, h, w, s, st);
}

extern "C" long long ecm_weights9_bwd_scratch_bytes(int B, int h, int w, int s) {
    return ecm_context_weights_bwd_scratch_bytes(B, h, w, s, 0);
}

extern "C" int ecm_weights9_bwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                                const float* W3, const float* w9, const float* gw9, float* glr, float* ghr, float* gW,
                                void* scratch, long long scratch_bytes, int B, int h, int w, int s, void* stream) {
    return ecm_context_weights_bwd(lr, hr, W0, W1, W2, W3, w9, gw9, glr, ghr, gW, scratch, scratch_bytes, B, h, w, s, 0,
                                   stream);
}
