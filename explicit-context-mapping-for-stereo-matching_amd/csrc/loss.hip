// Training loss and validation metrics of the harness in one pass over the three predictions
// (train.py:162,172-174 / train_kitti.py:205-216): mask = 0 < gt < maxdisp;
//   loss = w1 mean_mask smoothL1(p1-gt) + w2 mean_mask smoothL1(p2-gt) + w3 mean_mask smoothL1(p3-gt)   (beta = 1)
//   epe  = mean_mask |p3-gt|;   err3 = 100 - 100 * #{|p3-gt| < 3  or  |p3-gt| < 0.05 gt} / #mask
// The reference gathers `o[mask]` six times (a host sync each) and launches ~30 ATen kernels; here: one streaming
// kernel with per-workgroup partials, a fixed-order final sum in double (deterministic), and one backward kernel.
// An empty mask gives NaN, like the mean of an empty selection in the reference.
#include "common.h"

namespace {

constexpr int LT = 256;
constexpr int LOSS_MAX_BLOCKS = 1024;
constexpr int NPART = 6;           // s1, s2, s3 (smooth-L1 sums), count, sum |e3|, #good3

__device__ __forceinline__ float smooth_l1(float d) {
    const float a = fabsf(d);
    return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

__global__ __launch_bounds__(LT) void stereo_loss_partial(const float* __restrict__ p1, const float* __restrict__ p2,
                                                          const float* __restrict__ p3, const float* __restrict__ gt,
                                                          float* __restrict__ part, long long n, float maxdisp) {
    __shared__ float sm[NPART][LT / 64];
    float acc[NPART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT) {
        const float g = gt[i];
        if (g < maxdisp && g > 0.f) {
            const float e3 = p3[i] - g, a3 = fabsf(e3);
            acc[0] += smooth_l1(p1[i] - g);
            acc[1] += smooth_l1(p2[i] - g);
            acc[2] += smooth_l1(e3);
            acc[3] += 1.f;
            acc[4] += a3;
            acc[5] += (a3 < 3.f || a3 < 0.05f * g) ? 1.f : 0.f;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NPART; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) sm[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < NPART) {
        float v = 0.f;
        for (int w = 0; w < LT / 64; ++w) v += sm[threadIdx.x][w];
        part[(size_t)blockIdx.x * NPART + threadIdx.x] = v;
    }
}

// out[8]: loss, count, epe, err3, mean smooth-L1 of head 1, 2, 3, (unused)
__global__ void stereo_loss_final(const float* __restrict__ part, int nblocks, float w1, float w2, float w3,
                                  float* __restrict__ out) {
    // one wave: lane l adds the partials of blocks l, l+64, ... in fp64, then a fixed-order butterfly (deterministic; a single
    // thread walking all of them took 0.13 ms per step)
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double s[NPART] = {0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 64)
        for (int k = 0; k < NPART; ++k) s[k] += (double)part[(size_t)b * NPART + k];
#pragma unroll
    for (int k = 0; k < NPART; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off, 64);
    if (threadIdx.x != 0) return;
    const double cnt = s[3];
    const float m1 = (float)(s[0] / cnt), m2 = (float)(s[1] / cnt), m3 = (float)(s[2] / cnt);
    out[0] = w1 * m1 + w2 * m2 + w3 * m3;
    out[1] = (float)cnt;
    out[2] = (float)(s[4] / cnt);
    out[3] = (float)(100.0 - s[5] / cnt * 100.0);
    out[4] = m1; out[5] = m2; out[6] = m3; out[7] = 0.f;
}

// g_k[i] = gloss * w_k / count * clamp(p_k - gt, -1, 1) inside the mask, 0 outside
__global__ __launch_bounds__(LT) void stereo_loss_bwd(const float* __restrict__ p1, const float* __restrict__ p2,
                                                      const float* __restrict__ p3, const float* __restrict__ gt,
                                                      const float* __restrict__ out, const float* __restrict__ gloss,
                                                      float* __restrict__ g1, float* __restrict__ g2,
                                                      float* __restrict__ g3, long long n, float maxdisp, float w1,
                                                      float w2, float w3) {
    const float k = gloss[0] / out[1];
    for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT) {
        const float g = gt[i];
        const bool m = g < maxdisp && g > 0.f;
        g1[i] = m ? k * w1 * fminf(fmaxf(p1[i] - g, -1.f), 1.f) : 0.f;
        g2[i] = m ? k * w2 * fminf(fmaxf(p2[i] - g, -1.f), 1.f) : 0.f;
        g3[i] = m ? k * w3 * fminf(fmaxf(p3[i] - g, -1.f), 1.f) : 0.f;
    }
}

inline int loss_blocks(long long n) {
    long long b = (n + LT * 8 - 1) / (LT * 8);
    return (int)(b < 1 ? 1 : (b > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : b));
}

}  // namespace

extern "C" long long ecm_stereo_loss_scratch_bytes(long long n) {
    return n > 0 ? (long long)loss_blocks(n) * NPART * (long long)sizeof(float) : 0;
}

extern "C" int ecm_stereo_loss_fwd(const float* p1, const float* p2, const float* p3, const float* gt, float* out8,
                                   void* scratch, long long scratch_bytes, long long n, float maxdisp, float w1, float w2,
                                   float w3, void* stream) {
    ECM_CHECK_ARG(p1 && p2 && p3 && gt && out8 && scratch && n > 0);
    if (scratch_bytes < ecm_stereo_loss_scratch_bytes(n)) return ECM_ESCRATCH;
    const int nb = loss_blocks(n);
    float* part = static_cast<float*>(scratch);
    hipStream_t st = ecm_stream(stream);
    hipLaunchKernelGGL(stereo_loss_partial, dim3(nb), dim3(LT), 0, st, p1, p2, p3, gt, part, n, maxdisp);
    hipLaunchKernelGGL(stereo_loss_final, dim3(1), dim3(64), 0, st, part, nb, w1, w2, w3, out8);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_stereo_loss_bwd(const float* p1, const float* p2, const float* p3, const float* gt, const float* out8,
                                   const float* gloss, float* g1, float* g2, float* g3, long long n, float maxdisp,
                                   float w1, float w2, float w3, void* stream) {
    ECM_CHECK_ARG(p1 && p2 && p3 && gt && out8 && gloss && g1 && g2 && g3 && n > 0);
    hipLaunchKernelGGL(stereo_loss_bwd, dim3(loss_blocks(n)), dim3(LT), 0, ecm_stream(stream), p1, p2, p3, gt, out8, gloss,
                       g1, g2, g3, n, maxdisp, w1, w2, w3);
    return ECM_LAUNCH_RESULT();
}
