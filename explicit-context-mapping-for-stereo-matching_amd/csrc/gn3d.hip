// GroupNorm(32 groups) over 5-D volumes + fused ReLU / residual add, fwd + bwd
// (reference: nn.GroupNorm in convbn_3d cmfsm.py:49-58, hourglass 269/280, ReLU/skip logic 287-299, 685-693).
// HBM-bound: stats = 1 read of x; apply = 1 read (+1 skip read) + 1 write; all float4, grid-strided.
// Deterministic: two-stage reductions through caller-provided scratch, no float atomics.
#include "common.h"

namespace {

constexpr int GROUPS = 32;
constexpr int THREADS = 256;
constexpr long long CHUNK = 32768;           // floats reduced per workgroup in stage 1

__device__ __forceinline__ void block_reduce2(float& a, float& b, float* sm) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sm[wave * 2] = a; sm[wave * 2 + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float x = 0.f, y = 0.f;
        for (int i = 0; i < THREADS / 64; ++i) { x += sm[2 * i]; y += sm[2 * i + 1]; }
        sm[0] = x; sm[1] = y;
    }
    __syncthreads();
    a = sm[0];
    b = sm[1];
}

// stage 1: partial (sum, sumsq) of chunk `blockIdx.x` of span `blockIdx.y` (= b*32+g); span = n contiguous floats
__global__ __launch_bounds__(THREADS) void gn_stats_partial(const float* __restrict__ x, float* __restrict__ part,
                                                            long long n, int nchunks) {
    __shared__ float sm[2 * THREADS / 64];
    const float* p = x + (size_t)blockIdx.y * n;
    const long long beg = (long long)blockIdx.x * CHUNK;
    const long long end = beg + CHUNK < n ? beg + CHUNK : n;
    float s = 0.f, q = 0.f;
    if ((n & 3) == 0) {
        for (long long i = beg + threadIdx.x * 4; i < end; i += THREADS * 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (long long i = beg + threadIdx.x; i < end; i += THREADS) { const float v = p[i]; s += v; q += v * v; }
    }
    block_reduce2(s, q, sm);
    if (threadIdx.x == 0) {
        part[((size_t)blockIdx.y * nchunks + blockIdx.x) * 2] = s;
        part[((size_t)blockIdx.y * nchunks + blockIdx.x) * 2 + 1] = q;
    }
}

// stage 2: one thread per span, fixed-order sum in double -> (mean, rstd)
__global__ void gn_stats_final(const float* __restrict__ part, float* __restrict__ mean_rstd, int nspans, int nchunks,
                               long long n, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nspans) return;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < nchunks; ++c) { s += part[((size_t)i * nchunks + c) * 2]; q += part[((size_t)i * nchunks + c) * 2 + 1]; }
    const double mean = s / (double)n;
    double var = q / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_rstd[2 * i] = (float)mean;
    mean_rstd[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = relu?( x*a[b,c] + sh[b,c] (+ skip) ),  a = rstd*gamma, sh = beta - mean*a.  grid: (chunks, B*C)
template <bool RELU, bool SKIP>
__global__ __launch_bounds__(THREADS) void gn_apply(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ skip, float* __restrict__ y, int C,
                                                    long long S) {
    const int bc = blockIdx.y;
    const int b = bc / C, c = bc - b * C;
    const int g = c / (C / GROUPS);
    const float mean = mean_rstd[(b * GROUPS + g) * 2], rstd = mean_rstd[(b * GROUPS + g) * 2 + 1];
    const float a = rstd * gamma[c], sh = beta[c] - mean * a;
    const size_t base = (size_t)bc * S;
    const long long stride = (long long)gridDim.x * THREADS * 4;
    if ((S & 3) == 0) {
        for (long long i = ((long long)blockIdx.x * THREADS + threadIdx.x) * 4; i < S; i += stride) {
            float4 v = *reinterpret_cast<const float4*>(x + base + i);
            v.x = v.x * a + sh; v.y = v.y * a + sh; v.z = v.z * a + sh; v.w = v.w * a + sh;
            if (SKIP) {
                const float4 k = *reinterpret_cast<const float4*>(skip + base + i);
                v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
            }
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(y + base + i) = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * THREADS + threadIdx.x; i < S; i += (long long)gridDim.x * THREADS) {
            float v = x[base + i] * a + sh;
            if (SKIP) v += skip[base + i];
            if (RELU) v = fmaxf(v, 0.f);
            y[base + i] = v;
        }
    }
}

// ---- backward -------------------------------------------------------------------------------
// stage 1: per (b,c) chunk partials of  sg = sum g,  sgx = sum g*xhat   with g = gy * [y>0] (relu) ; also writes
// gskip = g when requested.  grid: (nchunks, B*C)
template <bool RELU>
__global__ __launch_bounds__(THREADS) void gn_bwd_partial(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ y, const float* __restrict__ gy,
                                                          float* __restrict__ gskip, float* __restrict__ part, int C,
                                                          long long S, int nchunks) {
    __shared__ float sm[2 * THREADS / 64];
    const int bc = blockIdx.y;
    const int b = bc / C, c = bc - b * C;
    const int g = c / (C / GROUPS);
    const float mean = mean_rstd[(b * GROUPS + g) * 2], rstd = mean_rstd[(b * GROUPS + g) * 2 + 1];
    const size_t base = (size_t)bc * S;
    const long long beg = (long long)blockIdx.x * CHUNK;
    const long long end = beg + CHUNK < S ? beg + CHUNK : S;
    float sg = 0.f, sgx = 0.f;
    if ((S & 3) == 0) {
        for (long long i = beg + threadIdx.x * 4; i < end; i += THREADS * 4) {
            float4 gv = *reinterpret_cast<const float4*>(gy + base + i);
            if (RELU) {
                const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
                if (!(yv.x > 0.f)) gv.x = 0.f;
                if (!(yv.y > 0.f)) gv.y = 0.f;
                if (!(yv.z > 0.f)) gv.z = 0.f;
                if (!(yv.w > 0.f)) gv.w = 0.f;
            }
            if (gskip) *reinterpret_cast<float4*>(gskip + base + i) = gv;
            const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
            sg += (gv.x + gv.y) + (gv.z + gv.w);
            sgx += (gv.x * ((xv.x - mean) * rstd) + gv.y * ((xv.y - mean) * rstd)) +
                   (gv.z * ((xv.z - mean) * rstd) + gv.w * ((xv.w - mean) * rstd));
        }
    } else {
        for (long long i = beg + threadIdx.x; i < end; i += THREADS) {
            float gv = gy[base + i];
            if (RELU && !(y[base + i] > 0.f)) gv = 0.f;
            if (gskip) gskip[base + i] = gv;
            sg += gv;
            sgx += gv * ((x[base + i] - mean) * rstd);
        }
    }
    block_reduce2(sg, sgx, sm);
    if (threadIdx.x == 0) {
        part[((size_t)bc * nchunks + blockIdx.x) * 2] = sg;
        part[((size_t)bc * nchunks + blockIdx.x) * 2 + 1] = sgx;
    }
}

// stage 2: one thread per (b,c): chan[b,c] = (sum g, sum g xhat) in double, fixed order
__global__ void gn_bwd_final_chan(const float* __restrict__ part, float* __restrict__ chan, int nbc, int nchunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbc) return;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < nchunks; ++c) { s += part[((size_t)i * nchunks + c) * 2]; q += part[((size_t)i * nchunks + c) * 2 + 1]; }
    chan[2 * i] = (float)s;
    chan[2 * i + 1] = (float)q;
}

// ggamma[c] += sum_b chan[b,c].sgx ; gbeta[c] += sum_b chan[b,c].sg      (one thread per c)
__global__ void gn_bwd_params(const float* __restrict__ chan, float* __restrict__ ggamma, float* __restrict__ gbeta,
                              int B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sg = 0.f, sgx = 0.f;
    for (int b = 0; b < B; ++b) { sg += chan[(b * C + c) * 2]; sgx += chan[(b * C + c) * 2 + 1]; }
    ggamma[c] += sgx;
    gbeta[c] += sg;
}

// gx = rstd * (g*gamma - s1/n - xhat*s2/n),  s1 = sum_{c in group} gamma_c sg_c,  s2 = sum gamma_c sgx_c
template <bool RELU>
__global__ __launch_bounds__(THREADS) void gn_bwd_apply(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ y,
                                                        const float* __restrict__ gy, const float* __restrict__ chan,
                                                        float* __restrict__ gx, int C, long long S) {
    const int bc = blockIdx.y;
    const int b = bc / C, c = bc - b * C;
    const int cpg = C / GROUPS, g = c / cpg;
    const float mean = mean_rstd[(b * GROUPS + g) * 2], rstd = mean_rstd[(b * GROUPS + g) * 2 + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < cpg; ++j) {
        const int cc = g * cpg + j;
        s1 += gamma[cc] * chan[(b * C + cc) * 2];
        s2 += gamma[cc] * chan[(b * C + cc) * 2 + 1];
    }
    const float invn = 1.f / ((float)cpg * (float)S);
    const float gm = gamma[c], k1 = s1 * invn, k2 = s2 * invn;
    const size_t base = (size_t)bc * S;
    if ((S & 3) == 0) {
        for (long long i = ((long long)blockIdx.x * THREADS + threadIdx.x) * 4; i < S; i += (long long)gridDim.x * THREADS * 4) {
            float4 gv = *reinterpret_cast<const float4*>(gy + base + i);
            if (RELU) {
                const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
                if (!(yv.x > 0.f)) gv.x = 0.f;
                if (!(yv.y > 0.f)) gv.y = 0.f;
                if (!(yv.z > 0.f)) gv.z = 0.f;
                if (!(yv.w > 0.f)) gv.w = 0.f;
            }
            const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
            float4 o;
            o.x = rstd * (gv.x * gm - k1 - ((xv.x - mean) * rstd) * k2);
            o.y = rstd * (gv.y * gm - k1 - ((xv.y - mean) * rstd) * k2);
            o.z = rstd * (gv.z * gm - k1 - ((xv.z - mean) * rstd) * k2);
            o.w = rstd * (gv.w * gm - k1 - ((xv.w - mean) * rstd) * k2);
            *reinterpret_cast<float4*>(gx + base + i) = o;
        }
    } else {
        for (long long i = (long long)blockIdx.x * THREADS + threadIdx.x; i < S; i += (long long)gridDim.x * THREADS) {
            float gv = gy[base + i];
            if (RELU && !(y[base + i] > 0.f)) gv = 0.f;
            const float xh = (x[base + i] - mean) * rstd;
            gx[base + i] = rstd * (gv * gm - k1 - xh * k2);
        }
    }
}

inline int chunks_of(long long n) { return (int)((n + CHUNK - 1) / CHUNK); }

}  // namespace

extern "C" long long ecm_gn3d_scratch_bytes(int B, int C, long long S) {
    // max over: stats partials [B*32][chunks(cpg*S)][2], bwd partials [B*C][chunks(S)][2] + chan [B*C][2]
    const long long cpg = C / 32 > 0 ? C / 32 : 1;
    const long long a = (long long)B * 32 * chunks_of(cpg * S) * 2;
    const long long b = (long long)B * C * chunks_of(S) * 2 + (long long)B * C * 2;
    return (a > b ? a : b) * (long long)sizeof(float);
}

extern "C" int ecm_gn3d_stats(const float* x, float* mean_rstd, void* scratch, long long scratch_bytes, int B, int C,
                              long long S, float eps, void* stream) {
    ECM_CHECK_ARG(x && mean_rstd && scratch && B > 0 && C > 0 && S > 0);
    if (C % GROUPS != 0) return ECM_EUNSUP;
    if (scratch_bytes < ecm_gn3d_scratch_bytes(B, C, S)) return ECM_ESCRATCH;
    const long long n = (long long)(C / GROUPS) * S;
    const int nchunks = chunks_of(n);
    float* part = static_cast<float*>(scratch);
    hipLaunchKernelGGL(gn_stats_partial, dim3(nchunks, B * GROUPS), dim3(THREADS), 0, ecm_stream(stream), x, part, n,
                       nchunks);
    hipLaunchKernelGGL(gn_stats_final, dim3((B * GROUPS + 63) / 64), dim3(64), 0, ecm_stream(stream), part, mean_rstd,
                       B * GROUPS, nchunks, n, eps);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_gn3d_apply(const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                              const float* skip, float* y, int B, int C, long long S, int relu, void* stream) {
    ECM_CHECK_ARG(x && mean_rstd && gamma && beta && y && B > 0 && C > 0 && S > 0);
    if (C % GROUPS != 0 || (long long)B * C > 65535) return ECM_EUNSUP;
    long long per = (S + THREADS * 4 - 1) / (THREADS * 4);
    int gx = (int)(per < 64 ? per : 64);
    dim3 grid(gx, B * C), block(THREADS);
    hipStream_t st = ecm_stream(stream);
    if (relu && skip) hipLaunchKernelGGL((gn_apply<true, true>), grid, block, 0, st, x, mean_rstd, gamma, beta, skip, y, C, S);
    else if (relu) hipLaunchKernelGGL((gn_apply<true, false>), grid, block, 0, st, x, mean_rstd, gamma, beta, skip, y, C, S);
    else if (skip) hipLaunchKernelGGL((gn_apply<false, true>), grid, block, 0, st, x, mean_rstd, gamma, beta, skip, y, C, S);
    else hipLaunchKernelGGL((gn_apply<false, false>), grid, block, 0, st, x, mean_rstd, gamma, beta, skip, y, C, S);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_gn3d_bwd(const float* x, const float* mean_rstd, const float* gamma, const float* y, const float* gy,
                            float* gx, float* gskip, float* ggamma, float* gbeta, void* scratch, long long scratch_bytes,
                            int B, int C, long long S, int relu, void* stream) {
    ECM_CHECK_ARG(x && mean_rstd && gamma && gy && gx && ggamma && gbeta && scratch && B > 0 && C > 0 && S > 0);
    ECM_CHECK_ARG(!relu || y);
    if (C % GROUPS != 0 || (long long)B * C > 65535) return ECM_EUNSUP;
    if (scratch_bytes < ecm_gn3d_scratch_bytes(B, C, S)) return ECM_ESCRATCH;
    const int nchunks = chunks_of(S);
    float* part = static_cast<float*>(scratch);
    float* chan = part + (size_t)B * C * nchunks * 2;
    hipStream_t st = ecm_stream(stream);
    dim3 g1(nchunks, B * C), block(THREADS);
    if (relu) hipLaunchKernelGGL(gn_bwd_partial<true>, g1, block, 0, st, x, mean_rstd, y, gy, gskip, part, C, S, nchunks);
    else hipLaunchKernelGGL(gn_bwd_partial<false>, g1, block, 0, st, x, mean_rstd, y, gy, gskip, part, C, S, nchunks);
    hipLaunchKernelGGL(gn_bwd_final_chan, dim3((B * C + 63) / 64), dim3(64), 0, st, part, chan, B * C, nchunks);
    hipLaunchKernelGGL(gn_bwd_params, dim3((C + 63) / 64), dim3(64), 0, st, chan, ggamma, gbeta, B, C);
    long long per = (S + THREADS * 4 - 1) / (THREADS * 4);
    dim3 g2((int)(per < 64 ? per : 64), B * C);
    if (relu) hipLaunchKernelGGL(gn_bwd_apply<true>, g2, block, 0, st, x, mean_rstd, gamma, y, gy, chan, gx, C, S);
    else hipLaunchKernelGGL(gn_bwd_apply<false>, g2, block, 0, st, x, mean_rstd, gamma, y, gy, chan, gx, C, S);
    return ECM_LAUNCH_RESULT();
}
