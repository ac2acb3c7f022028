// a3: eight-related explicit-context-mapping weights (reference eight_related_context_mapping,
// cmfsm.py:431-593; MLP similarity_measure1 304-358; offset tables matrix_generation 391-428).
//
// Per HR pixel (Y,X) and neighbour n (LR cell (Y/s+dy, X/s+dx)):
//   logit_n = W3 . phi(W2 . phi(W1 . phi( W0[:, :32] lr[cell] + W0[:, 32:64] hr[Y,X] + W0[:,64] offx + W0[:,65] offy )))
//   out-of-image cell -> logit -100;  w9 = softmax_n(logit).
// The first layer is linear in the 66-channel concat the reference materialises nine times at full
// resolution, so it is split: A = W0_lr lr is computed once per LR cell (proj kernel, channels-last
// scratch), B = W0_hr hr once per pixel, and the two offset columns are added per neighbour.
// Traffic is then the algorithmic 32*H*W*4 (hr) + 9*H*W*4 (w9) + small, instead of ~4 GB.
// fp32 VALU-bound (arithmetic intensity ~80 FLOP/B); one thread per HR pixel, one wave per 64
// consecutive X (256-B coalesced hr loads), MLP weights through the scalar cache.
#include "common.h"
#include "ecm_nbr.h"

namespace {

constexpr int CF = 32;                 // feature channels
constexpr int TX = 64, TY = 4;         // pixels per workgroup: 4 rows x 64 cols, one wave per row
constexpr int ASTRIDE = 36;            // padded cell stride in LDS (floats): 36c mod 64 spreads ds_read_b128

__device__ __forceinline__ float off_x(int t, int r, int s) { return ecm_off_x(t, r, s); }
__device__ __forceinline__ float off_y(int t, int r, int s) { return ecm_off_y(t, r, s); }

// A[b, cell, j] = sum_c W0[j, c] * lr[b, c, cell]   (channels-last scratch [B*h*w][32])
__global__ __launch_bounds__(256) void ecm_lr_proj(const float* __restrict__ lr, const float* __restrict__ W0,
                                                   float* __restrict__ A, int B, int hw) {
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

// The dense layers run on PACKED fp32 FMAs (v_pk_fma_f32: two lanes of the sum per instruction -- even and odd input
// channels accumulate separately and are added at the end; the weight pair comes straight from a 64-bit scalar load): half
// the VALU instructions of the scalar formulation, which was at 0.65 of the scalar-FMA rate.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NIN>
__device__ __forceinline__ float dot_pk(const float* __restrict__ wrow, const f32x2 (&x)[NIN / 2]) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NIN / 2; ++j) acc = *reinterpret_cast<const f32x2*>(wrow + 2 * j) * x[j] + acc;
    return acc.x + acc.y;
}

template <bool FINAL_ACT>
__device__ __forceinline__ float mlp_tail(const f32x2 (&h0)[CF / 2], const float* __restrict__ W1,
                                          const float* __restrict__ W2, const float* __restrict__ W3) {
    f32x2 h1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h1[i].x = leaky(dot_pk<CF>(W1 + (2 * i) * CF, h0));
        h1[i].y = leaky(dot_pk<CF>(W1 + (2 * i + 1) * CF, h0));
    }
    f32x2 h2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h2[i].x = leaky(dot_pk<16>(W2 + (2 * i) * 16, h1));
        h2[i].y = leaky(dot_pk<16>(W2 + (2 * i + 1) * 16, h1));
    }
    const float o = dot_pk<8>(W3, h2);
    return FINAL_ACT ? leaky(o) : o;
}

template <int VAR>
__global__ __launch_bounds__(256) void ecm_weights_fwd_kernel(const float* __restrict__ A, const float* __restrict__ hr,
                                                               const float* __restrict__ W0, const float* __restrict__ W1,
                                                               const float* __restrict__ W2, const float* __restrict__ W3,
                                                               float* __restrict__ w9, int h, int w, int s) {
    extern __shared__ __attribute__((aligned(16))) float As[];    // [(cy 3+)][cells_x + 2][ASTRIDE]
    const int H = h * s, W = w * s;
    const int b = blockIdx.z;
    const int Y0 = blockIdx.y * TY, X0 = blockIdx.x * TX;
    // LR cells touched by the tile (+1 halo)
    const int cy0 = Y0 / s - 1, cx0 = X0 / s - 1;
    const int ncy = (Y0 + TY - 1) / s - Y0 / s + 3, ncx = (X0 + TX - 1) / s - X0 / s + 3;
    for (int e = threadIdx.x; e < ncy * ncx * (CF / 4); e += blockDim.x) {
        const int q = e % (CF / 4), cell = e / (CF / 4);
        const int cy = cy0 + cell / ncx, cx = cx0 + cell % ncx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cy >= 0 && cy < h && cx >= 0 && cx < w)
            v = reinterpret_cast<const float4*>(A + (((size_t)b * h + cy) * w + cx) * CF)[q];
        *reinterpret_cast<float4*>(As + cell * ASTRIDE + q * 4) = v;
    }
    __syncthreads();
    const int Y = Y0 + (threadIdx.x >> 6), X = X0 + (threadIdx.x & 63);
    if (Y >= H || X >= W) return;
    const size_t HW = (size_t)H * W;
    const float* hp = hr + (size_t)b * CF * HW + (size_t)Y * W + X;
    f32x2 hv[CF / 2];
#pragma unroll
    for (int c = 0; c < CF / 2; ++c) { hv[c].x = hp[(size_t)(2 * c) * HW]; hv[c].y = hp[(size_t)(2 * c + 1) * HW]; }
    float Bv[CF];
#pragma unroll
    for (int j = 0; j < CF; ++j) Bv[j] = dot_pk<CF>(W0 + j * 66 + 32, hv);      // row stride 66 floats: 8-byte aligned pairs
    const int cy = Y / s, cx = X / s, ry = Y - cy * s, rx = X - cx * s;
    using NB = Nbr<VAR>;
    constexpr int NN = NB::N;
    float logit[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const int yy = cy + NB::dy(n), xx = cx + NB::dx(n);
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) { logit[n] = NB::PAD; continue; }     // cmfsm.py:451-452 / sub_8:461-462
        const float ox = off_x(NB::tab(n), rx, s), oy = off_y(NB::tab(n), ry, s);
        const float* a = As + ((yy - cy0) * ncx + (xx - cx0)) * ASTRIDE;
        f32x2 h0[CF / 2];
#pragma unroll
        for (int j = 0; j < CF; j += 4) {
            const float4 av = *reinterpret_cast<const float4*>(a + j);
            h0[j / 2].x = leaky(fmaf(W0[(j + 0) * 66 + 65], oy, fmaf(W0[(j + 0) * 66 + 64], ox, av.x + Bv[j + 0])));
            h0[j / 2].y = leaky(fmaf(W0[(j + 1) * 66 + 65], oy, fmaf(W0[(j + 1) * 66 + 64], ox, av.y + Bv[j + 1])));
            h0[j / 2 + 1].x = leaky(fmaf(W0[(j + 2) * 66 + 65], oy, fmaf(W0[(j + 2) * 66 + 64], ox, av.z + Bv[j + 2])));
            h0[j / 2 + 1].y = leaky(fmaf(W0[(j + 3) * 66 + 65], oy, fmaf(W0[(j + 3) * 66 + 64], ox, av.w + Bv[j + 3])));
        }
        logit[n] = mlp_tail<NB::FINAL_ACT>(h0, W1, W2, W3);
    }
    float m = logit[0];
#pragma unroll
    for (int n = 1; n < NN; ++n) m = fmaxf(m, logit[n]);
    float e[NN], sum = 0.f;
#pragma unroll
    for (int n = 0; n < NN; ++n) { e[n] = expf(logit[n] - m); sum += e[n]; }
    const float inv = 1.f / sum;
    float* op = w9 + (size_t)b * NN * HW + (size_t)Y * W + X;
#pragma unroll
    for (int n = 0; n < NN; ++n) op[(size_t)n * HW] = NB::TIMES_LOGIT ? e[n] * inv * logit[n] : e[n] * inv;
}

inline int lds_floats(int s) {
    const int ncy = (TY - 1) / s + 4, ncx = (TX - 1) / s + 4;      // upper bound on the cell window
    return ncy * ncx * ASTRIDE;
}

}  // namespace

extern "C" long long ecm_weights9_scratch_bytes(int B, int h, int w) { return (long long)B * h * w * CF * 4; }

extern "C" int ecm_context_weights_fwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                                       const float* W3, float* out, void* scratch, long long scratch_bytes, int B, int h,
                                       int w, int s, int variant, void* stream) {
    ECM_CHECK_ARG(lr && hr && W0 && W1 && W2 && W3 && out && scratch && B > 0 && h > 0 && w > 0 && s > 0);
    if (s % 2 != 0 || B > 65535 || variant < 0 || variant > 2) return ECM_EUNSUP;   // the reference exits on odd scale
    // eight-related (variant 0): the reference's offset tables are hard-coded for scale 4 (matrix_generation, cmfsm.py:391-428,
    // quirk Q3) and its forward fails with a shape error at any other scale -- there is nothing to be equal to, so refuse
    if (variant == 0 && s != 4) return ECM_EUNSUP;
    if (scratch_bytes < ecm_weights9_scratch_bytes(B, h, w)) return ECM_ESCRATCH;
    hipStream_t st = ecm_stream(stream);
    float* A = static_cast<float*>(scratch);
    const long long cells = (long long)B * h * w;
    hipLaunchKernelGGL(ecm_lr_proj, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, lr, W0, A, B, h * w);
    const int H = h * s, W = w * s;
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, B);
    const size_t lds = lds_floats(s) * sizeof(float);
    if (variant == 0) hipLaunchKernelGGL(ecm_weights_fwd_kernel<0>, grid, dim3(256), lds, st, A, hr, W0, W1, W2, W3, out, h, w, s);
    else if (variant == 1) hipLaunchKernelGGL(ecm_weights_fwd_kernel<1>, grid, dim3(256), lds, st, A, hr, W0, W1, W2, W3, out, h, w, s);
    else hipLaunchKernelGGL(ecm_weights_fwd_kernel<2>, grid, dim3(256), lds, st, A, hr, W0, W1, W2, W3, out, h, w, s);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_weights9_fwd(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                                const float* W3, float* w9, void* scratch, long long scratch_bytes, int B, int h, int w,
                                int s, void* stream) {
    return ecm_context_weights_fwd(lr, hr, W0, W1, W2, W3, w9, scratch, scratch_bytes, B, h, w, s, 0, stream);
}
