// GroupNorm(32 groups) over 5-D volumes + fused ReLU / residual add, fwd + bwd
// (reference: nn.GroupNorm in convbn_3d cmfsm.py:49-58, hourglass 269/280, ReLU/skip logic 287-299, 685-693).
// HBM-bound: stats = 1 read of x; apply = 1 read (+1 skip read) + 1 write; all float4, grid-strided.
// Deterministic: two-stage reductions through caller-provided scratch, no float atomics.
#include "common.h"
#include <cstdlib>

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

// y = fma(x, a, sh): the coefficients are formed the same way wherever they are needed, so that the ReLU mask
// recomputed in the backward pass is bit-identical to the forward decision.
__device__ __forceinline__ void gn_affine(float mean, float rstd, float gamma, float beta, float& a, float& sh) {
    a = rstd * gamma;
    sh = __builtin_fmaf(-mean, a, beta);
}

// Statistics are accumulated as sums of d = x - K and d^2 around a PIVOT K (gn_pivot below), and the variance is
// E[d^2] - E[d]^2: the textbook E[x^2] - E[x]^2 cancels catastrophically when |mean| >> std -- the SPP branches normalise
// maps of 2..6 pixels per group whose values differ in the third digit, and round 2's kernels lost three digits of the
// normalised value there (measured: 1.2e-3 relative against 1.9e-5 for torch's Welford form, which put the whole
// encoder's low-resolution feature 12x further from the fp64 truth than the reference's fp32).
// Round 4: a pivot taken from ONE element makes the accuracy of a whole group hang on that element -- an outlier
// there (|K - mean| >> std: an activation spike in the corner, a border pixel) brings the cancellation back for every
// element of the group.  The pivot is now a TRIMMED mean of 16 elements spread evenly over the span (mid-points of its
// sixteenths; smallest and largest sample dropped), or the plain mean of a span shorter than 16: a single outlier among the
// samples is discarded outright, and a typical span gives |K - mean| ~ std / 4.  Every workgroup of a span and the
// finishing stage evaluate the same expression in the same order, so they agree bit for bit.
__device__ __forceinline__ float gn_pivot(const float* __restrict__ p, long long n) {
    float s = 0.f;
    if (n < 16) {
        for (long long i = 0; i < n; ++i) s += p[i];
        return s / (float)n;
    }
    const long long st = n / 16;
    float lo = p[st >> 1], hi = lo;
    s = lo;
#pragma unroll
    for (int j = 1; j < 16; ++j) {
        const float v = p[(long long)j * st + (st >> 1)];
        s += v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    return (s - lo - hi) * (1.0f / 14.0f);
}

// stage 1: partial (sum d, sum d^2) of chunk `blockIdx.x` of span `blockIdx.y` (= b*32+g); span = n contiguous floats
__global__ __launch_bounds__(THREADS) void gn_stats_partial(const float* __restrict__ x, float* __restrict__ part,
                                                            long long n, int nchunks) {
    __shared__ float sm[2 * THREADS / 64];
    const float* p = x + (size_t)blockIdx.y * n;
    const float K = gn_pivot(p, n);
    const long long beg = (long long)blockIdx.x * CHUNK;
    const long long end = beg + CHUNK < n ? beg + CHUNK : n;
    float s = 0.f, q = 0.f;
    if ((n & 3) == 0) {
        for (long long i = beg + threadIdx.x * 4; i < end; i += THREADS * 4) {
            float4 v = *reinterpret_cast<const float4*>(p + i);
            v.x -= K; v.y -= K; v.z -= K; v.w -= K;
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (long long i = beg + threadIdx.x; i < end; i += THREADS) { const float v = p[i] - K; s += v; q += v * v; }
    }
    block_reduce2(s, q, sm);
    if (threadIdx.x == 0) {
        part[((size_t)blockIdx.y * nchunks + blockIdx.x) * 2] = s;
        part[((size_t)blockIdx.y * nchunks + blockIdx.x) * 2 + 1] = q;
    }
}

// stage 2: one thread per span, fixed-order sum in double -> (mean, rstd)
__global__ void gn_stats_final(const float* __restrict__ x, const float* __restrict__ part, float* __restrict__ mean_rstd,
                               int nspans, int nchunks, long long n, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nspans) return;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < nchunks; ++c) { s += part[((size_t)i * nchunks + c) * 2]; q += part[((size_t)i * nchunks + c) * 2 + 1]; }
    const double dm = s / (double)n;                    // mean of x - K
    const double mean = (double)gn_pivot(x + (size_t)i * n, n) + dm;
    double var = q / (double)n - dm * dm;
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
    float a, sh;
    gn_affine(mean, rstd, gamma[c], beta[c], a, sh);
    const size_t base = (size_t)bc * S;
    const long long stride = (long long)gridDim.x * THREADS * 4;
    if ((S & 3) == 0) {
        for (long long i = ((long long)blockIdx.x * THREADS + threadIdx.x) * 4; i < S; i += stride) {
            float4 v = *reinterpret_cast<const float4*>(x + base + i);
            v.x = __builtin_fmaf(v.x, a, sh); v.y = __builtin_fmaf(v.y, a, sh);
            v.z = __builtin_fmaf(v.z, a, sh); v.w = __builtin_fmaf(v.w, a, sh);
            if (SKIP) {
                const float4 k = *reinterpret_cast<const float4*>(skip + base + i);
                v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
            }
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(y + base + i) = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * THREADS + threadIdx.x; i < S; i += (long long)gridDim.x * THREADS) {
            float v = __builtin_fmaf(x[base + i], a, sh);
            if (SKIP) v += skip[base + i];
            if (RELU) v = fmaxf(v, 0.f);
            y[base + i] = v;
        }
    }
}

// ---- backward -------------------------------------------------------------------------------
// stage 1: per (b,c) chunk partials of  sg = sum g,  sgx = sum g*xhat   with g = gy * [y>0] (relu) ; also writes
// gskip = g when requested.  grid: (nchunks, B*C)
template <int MASK>
__global__ __launch_bounds__(THREADS) void gn_bwd_partial(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
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
    float a = 0.f, sh = 0.f;
    if (MASK == 2) gn_affine(mean, rstd, gamma[c], beta[c], a, sh);
    float sg = 0.f, sgx = 0.f;
    if ((S & 3) == 0) {
        for (long long i = beg + threadIdx.x * 4; i < end; i += THREADS * 4) {
            float4 gv = *reinterpret_cast<const float4*>(gy + base + i);
            const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
            if (MASK == 1) {
                const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
                if (!(yv.x > 0.f)) gv.x = 0.f;
                if (!(yv.y > 0.f)) gv.y = 0.f;
                if (!(yv.z > 0.f)) gv.z = 0.f;
                if (!(yv.w > 0.f)) gv.w = 0.f;
            }
            if (MASK == 2) {
                if (!(__builtin_fmaf(xv.x, a, sh) > 0.f)) gv.x = 0.f;
                if (!(__builtin_fmaf(xv.y, a, sh) > 0.f)) gv.y = 0.f;
                if (!(__builtin_fmaf(xv.z, a, sh) > 0.f)) gv.z = 0.f;
                if (!(__builtin_fmaf(xv.w, a, sh) > 0.f)) gv.w = 0.f;
            }
            if (gskip) *reinterpret_cast<float4*>(gskip + base + i) = gv;
            sg += (gv.x + gv.y) + (gv.z + gv.w);
            sgx += (gv.x * ((xv.x - mean) * rstd) + gv.y * ((xv.y - mean) * rstd)) +
                   (gv.z * ((xv.z - mean) * rstd) + gv.w * ((xv.w - mean) * rstd));
        }
    } else {
        for (long long i = beg + threadIdx.x; i < end; i += THREADS) {
            float gv = gy[base + i];
            if (MASK == 1 && !(y[base + i] > 0.f)) gv = 0.f;
            if (MASK == 2 && !(__builtin_fmaf(x[base + i], a, sh) > 0.f)) gv = 0.f;
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

// ggamma[c] = sum_b chan[b,c].sgx ; gbeta[c] = sum_b chan[b,c].sg      (one thread per c)
__global__ void gn_bwd_params(const float* __restrict__ chan, float* __restrict__ ggamma, float* __restrict__ gbeta,
                              int B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sg = 0.f, sgx = 0.f;
    for (int b = 0; b < B; ++b) { sg += chan[(b * C + c) * 2]; sgx += chan[(b * C + c) * 2 + 1]; }
    ggamma[c] = sgx;
    gbeta[c] = sg;
}

// gx = rstd * (g*gamma - s1/n - xhat*s2/n),  s1 = sum_{c in group} gamma_c sg_c,  s2 = sum gamma_c sgx_c
template <int MASK>
__global__ __launch_bounds__(THREADS) void gn_bwd_apply(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ y, const float* __restrict__ gy,
                                                        const float* __restrict__ chan, float* __restrict__ gx, int C,
                                                        long long S) {
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
    float a = 0.f, sh = 0.f;
    if (MASK == 2) gn_affine(mean, rstd, gm, beta[c], a, sh);
    const size_t base = (size_t)bc * S;
    if ((S & 3) == 0) {
        for (long long i = ((long long)blockIdx.x * THREADS + threadIdx.x) * 4; i < S; i += (long long)gridDim.x * THREADS * 4) {
            float4 gv = *reinterpret_cast<const float4*>(gy + base + i);
            const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
            if (MASK == 1) {
                const float4 yv = *reinterpret_cast<const float4*>(y + base + i);
                if (!(yv.x > 0.f)) gv.x = 0.f;
                if (!(yv.y > 0.f)) gv.y = 0.f;
                if (!(yv.z > 0.f)) gv.z = 0.f;
                if (!(yv.w > 0.f)) gv.w = 0.f;
            }
            if (MASK == 2) {
                if (!(__builtin_fmaf(xv.x, a, sh) > 0.f)) gv.x = 0.f;
                if (!(__builtin_fmaf(xv.y, a, sh) > 0.f)) gv.y = 0.f;
                if (!(__builtin_fmaf(xv.z, a, sh) > 0.f)) gv.z = 0.f;
                if (!(__builtin_fmaf(xv.w, a, sh) > 0.f)) gv.w = 0.f;
            }
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
            if (MASK == 1 && !(y[base + i] > 0.f)) gv = 0.f;
            if (MASK == 2 && !(__builtin_fmaf(x[base + i], a, sh) > 0.f)) gv = 0.f;
            const float xh = (x[base + i] - mean) * rstd;
            gx[base + i] = rstd * (gv * gm - k1 - xh * k2);
        }
    }
}

// ---- cluster-fused kernels -------------------------------------------------------------------
// The two-stage kernels above read every tensor once for the statistics and again for the elementwise pass (forward:
// 2 reads + 1 write; backward with ReLU: 6 reads + 1 write).  The fused kernels keep a workgroup's slice of the span
// in REGISTERS across the reduction: a cluster of `cl` co-resident workgroups owns one (sample, group) span, each
// workgroup loads its slice once (<= MAXV4 float4 per thread), publishes a partial sum into its slot, collects the
// cl slots of the span (agent-scope atomics), reduces them in a fixed order and finishes from registers
// (forward: 1 read + 1 write; backward: 2 reads + 1 write, the ReLU mask is recomputed from x when no skip was added).
// The grid never exceeds what is resident at once (occupancy query), so the collect cannot deadlock; a bounded
// spin turns a lost workgroup into NaN statistics instead of a hang.  Shapes that do not fit (S % 4, spans larger
// than the register capacity of 256 workgroups) take the two-stage path.
constexpr int FUSED_MAX_CPG = 8;
#ifndef ECM_GN_FWD_MAXV4
#define ECM_GN_FWD_MAXV4 32       // measured on MI355X (tools/gn_bench.py, B=4 x 32 x 48x144x240): 32@2 / 20@2 -> fwd 0.344 ms,
#define ECM_GN_FWD_OCC 2          // bwd 0.544 ms; 16@4 / 8@4 -> 0.417 / 0.594; 24@3 / 12@3 -> 0.368 / 0.589 (two-stage fwd 0.517)
#define ECM_GN_BWD_MAXV4 20
#define ECM_GN_BWD_OCC 2
#endif
#ifndef ECM_GN_FWDS_MAXV4
#define ECM_GN_FWDS_MAXV4 24      // forward WITH a residual operand: x and skip both held in registers across the reduction (as the
                                  // backward holds x and gy), so the pass after the rendezvous only writes.  Round 4: the skip used to be
                                  // read there, between the stores -- 0.553 -> 0.519 ms at 4 x 32 x 48x144x240, 0.128 -> 0.083 at
                                  // 8 x 128 x 144x240, 0.157 -> 0.128 at 4 x 64 x 24x72x120 (16: 0.570 / 0.086 / 0.131; 20: 0.532 / 0.080 / 0.129)
#endif
constexpr int FWD_MAXV4 = ECM_GN_FWD_MAXV4;   // float4 per thread kept in registers; OCC workgroups per CU overlap one
constexpr int BWD_MAXV4 = ECM_GN_BWD_MAXV4;   // cluster's wait with another's loads/stores
constexpr int FWDS_MAXV4 = ECM_GN_FWDS_MAXV4;
// Deferred stores.  A workgroup is idle for a quarter of every ticket: the rendezvous -- publish the partial sums, collect the
// cluster's -- is two dependent trips through a busy memory pipeline, 5.6-10 us whatever the cluster size
// (tools/gn_phase_profile.py).  Nothing can be LOADED for the next ticket meanwhile (the registers hold this one's slab), but
// something can be STORED: the finishing pass parks the first STASH float4 per thread of its output in LDS (otherwise unused
// by these kernels) instead of storing them, and the next ticket sends them out between its publish and its collect -- under
// the wait.  (The last ticket of a workgroup flushes after its own finishing pass.)
#ifndef ECM_GN_STASH_V4
#define ECM_GN_STASH_V4 16        // x 256 threads x 16 B = 64 KB per workgroup, two workgroups per CU; 0 = off
#endif
constexpr int STASH_V4 = ECM_GN_STASH_V4;
// (the backward variant that reads its mask from y AND writes the residual operand's gradient has no registers left for it)
constexpr int bwd_stash_v4(int mask, int gskip) { return (mask == 1 && gskip == 1) ? 0 : (STASH_V4 < BWD_MAXV4 ? STASH_V4 : BWD_MAXV4); }

// Slice access through buffer descriptors (base = first float4 of this workgroup's slice, num_records = slice bytes):
// one 32-bit per-thread offset + immediates instead of a 64-bit address per register tile row, and float4s past the
// slice read as 0 / are not written, so the tile loops carry no bounds predicate.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const float* p, int nv4) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, nv4 > 0 ? nv4 * 16 : 0, 0x00020000);
}
// cache-policy bits of the streaming loads / stores (aux operand: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef ECM_GN_LD_AUX
#define ECM_GN_LD_AUX 0
#endif
#ifndef ECM_GN_ST_AUX
#define ECM_GN_ST_AUX 2          // nt: measured 4-18 % per launch (profiles/r04_gn_store_policy.txt); loads gain nothing
#endif
__device__ __forceinline__ float4 slice_ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, ECM_GN_LD_AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void slice_st(__amdgpu_buffer_rsrc_t r, unsigned off, const float4& f) {
    u32x4 v;
    v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, ECM_GN_ST_AUX);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Cluster exchange without fences: a workgroup's partial (two floats) is published as ONE 64-bit agent-scope atomic
// store into a slot preset to EMPTY, and read back with agent-scope atomic loads, so data and "ready" flag travel
// together and no release/acquire (= whole-L2 write-back / invalidate on a multi-XCD part) is needed.
constexpr unsigned long long SLOT_EMPTY = ~0ull;          // (NaN, NaN) with an all-ones payload: never produced by sums

__device__ __forceinline__ void slot_publish(unsigned long long* slot, float s, float q) {
    const unsigned long long v = ((unsigned long long)__float_as_uint(q) << 32) | (unsigned long long)__float_as_uint(s);
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Poll until the slot is filled.  Bounded by WALL TIME (s_memrealtime: 100 MHz, constant): a cluster member that never
// became resident within `poll_ticks` turns into NaN results AND a sticky error word the host reports (ECM_EASYNC) --
// never into a hang, never into a silent rc 0.
__device__ __forceinline__ bool slot_collect(const unsigned long long* slot, float& s, float& q, unsigned long long poll_ticks) {
    unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == SLOT_EMPTY) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int spins = 0;
        do {
            __builtin_amdgcn_s_sleep(4);
            v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((++spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > poll_ticks) break;
        } while (v == SLOT_EMPTY);
    }
    s = __uint_as_float((unsigned)(v & 0xffffffffull));
    q = __uint_as_float((unsigned)(v >> 32));
    return v != SLOT_EMPTY;
}

// Work assignment.  A cluster's `cl` workgroups must all be running for any of them to finish, so membership must not
// depend on which workgroups the dispatcher happens to have placed: every workgroup draws TICKETS from one counter
// (ticket t -> span t / cl, member t % cl), and never waits while holding a ticket it has not published for.  All
// tickets below a drawn one are then held by workgroups that are running towards their publish (or done), so at most ONE cluster is ever incomplete and every other running workgroup sits in a complete cluster, finishes
// and draws the next ticket -- progress needs only `cl` running workgroups of this launch, whatever else shares the
// device (a second stream or process, CU masks) and in whatever order the hardware dispatches.  The counter lives behind
// the slots and is preset to 0xFFFFFFFF by the same memset (first draw wraps to 0).  STATIC = the round-1 scheme
// (member = blockIdx): kept only as a diagnostic (ecm_gn3d_cluster_mode(2)).
struct Tickets {
    unsigned* ctr;
    unsigned total, stride;
    bool dynamic;
    __device__ __forceinline__ unsigned first() const {
        return dynamic ? atomicAdd(ctr, 1u) + 1u : blockIdx.x;
    }
    __device__ __forceinline__ unsigned next(unsigned t) const {
        return dynamic ? atomicAdd(ctr, 1u) + 1u : t + stride;
    }
};

__device__ __forceinline__ void report_timeout(unsigned* status) {
    __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct FusedGeom {
    int cpg, wpc, cl, nspans, grid;
    long long v4_per_wg;
    bool ok;
};

// cl <= FUSED_MAX_CL_RUN: a cluster never needs more than a quarter of the device to be running at once, so two or
// three cluster launches sharing the device (streams, processes) cannot hold each other's last members out for good.
constexpr int FUSED_MAX_CL_RUN = 128;

inline FusedGeom fused_geom(int B, int C, long long S, int maxv4, int resident) {
    FusedGeom g{};
    g.ok = false;
    if (S % 4 != 0 || S / 4 >= 0x7fffffffLL - 0x10000 || C % GROUPS != 0 || resident <= 0) return g;
    g.cpg = C / GROUPS;
    if (g.cpg > FUSED_MAX_CPG) return g;
    const long long nv4 = S / 4, cap = (long long)THREADS * maxv4;
    const long long wpc = (nv4 + cap - 1) / cap;
    if (wpc * g.cpg > FUSED_MAX_CL_RUN || wpc * g.cpg * 4 > resident) return g;
    g.wpc = (int)wpc;
    g.cl = g.wpc * g.cpg;
    g.nspans = B * GROUPS;
    const long long total = (long long)g.nspans * g.cl;
    if (total >= 0x7fffffffLL) return g;
    g.grid = (int)(total < resident ? total : resident);
    g.v4_per_wg = (nv4 + wpc - 1) / wpc;
    g.ok = true;
    return g;
}

template <class K>
inline int resident_workgroups(K kern, int dynamic_lds = 0) {
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, THREADS, (size_t)dynamic_lds) != hipSuccess) return 0;
    return cus * per;
}

// scratch layout of the fused path: slots [nspans][cl] x 64 bit, then the ticket counter (all preset to 0xFF.. by the launcher)
struct FusedCtl {
    unsigned long long* slots;
    unsigned* ticket;
    unsigned* done;                   // per span: members that have collected (preset 0xFFFFFFFF); the last one restores the
                                      // span's slots and this counter to their preset state (see ecm_gn3d_fwd_p)
    unsigned* status;                 // host-mapped sticky error word
    unsigned long long poll_ticks;    // 100 MHz ticks
    int dynamic;                      // 1: tickets, 0: static ids (diagnostic)
};

// Self-cleaning exchange memory.  Every member of a cluster, once it has COLLECTED the span's slots, counts itself on the
// span's `done` word; the last one (nobody polls those slots any more) puts the slots and the word back to their preset
// (all-ones) state, and the workgroup that draws the very last ticket of the launch does the same for the ticket counter.
// A launch that ends without a time-out therefore leaves the exchange memory exactly as it found it, and a caller that
// keeps it across calls (ecm_gn3d_fwd_p / _bwd_p) needs no memset per launch (172 per training step before).
// Every workgroup of the launch draws until it gets a ticket >= total, i.e. total + gridDim.x draws in all; whoever makes the
// last one puts the counter back.  That can be a workgroup's FIRST draw -- one the dispatcher placed only after the others had
// worked through every ticket (another stream's kernels held its CU): such a workgroup never enters the ticket loop, so the
// check sits on both draws.  (Found when weight gradients were moved to a second stream: the counter stayed at total + grid - 1,
// the next launch started in the middle of its ticket range, and its clusters waited for members nobody would ever be.)
__device__ __forceinline__ void ticket_drawn(const FusedCtl& ctl, const Tickets& tk, unsigned t) {
    if (tk.dynamic && t == tk.total + tk.stride - 1u)                 // the last draw of the launch (stride == gridDim.x)
        __hip_atomic_store(ctl.ticket, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cluster_done(const FusedCtl& ctl, const Tickets& tk, int span, int cl, unsigned tnext) {
    ticket_drawn(ctl, tk, tnext);
    const unsigned old = atomicAdd(ctl.done + span, 1u);              // preset 0xFFFFFFFF: the k-th arrival reads k - 2
    return old + 2u == (unsigned)cl;
}
__device__ __forceinline__ void cluster_restore(const FusedCtl& ctl, unsigned long long* sp, int span, int cl, int tid) {
    if (tid < cl) __hip_atomic_store(sp + tid, SLOT_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(ctl.done + span, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef GN_PROFILE
__device__ unsigned long long gn_prof[4];          // load phase, rendezvous, finishing pass, tickets (100 MHz ticks; sums over workgroups)
#define GN_T(i) do { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&gn_prof[i], now_ - last_); last_ = now_; } } while (0)
#else
#define GN_T(i) do { } while (0)
#endif

template <bool RELU, bool SKIP>
__global__ __launch_bounds__(THREADS, ECM_GN_FWD_OCC) void gn_fused_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ skip,
                                                           float* __restrict__ y, float* __restrict__ mean_rstd,
                                                           FusedCtl ctl, int C, long long S, int cpg,
                                                           int wpc, int nspans, long long v4_per_wg, float eps) {
    constexpr int MAXV4 = SKIP ? FWDS_MAXV4 : FWD_MAXV4;
    constexpr int NST = STASH_V4 < MAXV4 ? STASH_V4 : MAXV4;          // float4 per thread whose store is deferred through LDS
    extern __shared__ __attribute__((aligned(16))) float4 stash[];     // [NST][THREADS]; a thread reads back only what it wrote
    __shared__ float sm[2 * THREADS / 64];
    __shared__ double smd[2 * THREADS / 64];
    __shared__ unsigned tick_s, last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = cpg * wpc;
    bool parked = false;                                // the previous ticket's first NST float4 per thread wait in `stash`
    __amdgpu_buffer_rsrc_t parked_r = slice_rsrc(y, 0);
    // (Unconditional: switching it off at run time for launches of one or two tickets per workgroup -- where there is no later
    // rendezvous to hide under and the trip through LDS costs 4-6 % -- made hipcc schedule BOTH paths worse: 22.0 ms of
    // GroupNorm per step against 20.9 with the parking always on and 21.3 without it.)
    constexpr bool stash_on = NST > 0;
    const long long nv4 = S >> 2;                       // < 2^31 (checked by the host): 32-bit indices within a channel
    const unsigned toff = (unsigned)(wave * MAXV4 * 64 + lane) * 16u;     // wave-contiguous rows of 64 float4 (1 KB)
    const Tickets tk{ctl.ticket, (unsigned)nspans * (unsigned)cl, gridDim.x, ctl.dynamic != 0};
    if (tid == 0) { tick_s = tk.first(); if (tick_s >= tk.total) ticket_drawn(ctl, tk, tick_s); }
    __syncthreads();
    unsigned t = tick_s;
#ifdef GN_PROFILE
    unsigned long long last_ = __builtin_amdgcn_s_memrealtime();
#endif
    while (t < tk.total) {
        const int span = (int)(t / (unsigned)cl), wic = (int)(t - (unsigned)span * (unsigned)cl);
        const int cig = wic / wpc, w = wic - cig * wpc;
        const int v0 = (int)((long long)w * v4_per_wg);
        const int v1 = (int)(v0 + v4_per_wg < nv4 ? v0 + v4_per_wg : nv4);
        const int b = span / GROUPS, g = span - b * GROUPS;
        const int c = g * cpg + cig;
        const size_t base = ((size_t)b * C + c) * S;
        const auto xr = slice_rsrc(x + base + (size_t)v0 * 4, v1 - v0);
        // pivot of the span (see gn_pivot): the same for every member of the cluster
        const float K = gn_pivot(x + ((size_t)b * C + (size_t)g * cpg) * S, (long long)cpg * S);
        const unsigned nslice = (unsigned)(v1 - v0), vidx = (unsigned)(wave * MAXV4 * 64 + lane);
        float4 v[MAXV4], kv[SKIP ? MAXV4 : 1];
        float s = 0.f, q = 0.f;
        const auto kr = slice_rsrc(SKIP ? skip + base + (size_t)v0 * 4 : x, SKIP ? v1 - v0 : 0);
        if (SKIP) {                                     // the residual operand travels with x: in flight under the reduction
#pragma unroll
            for (int j = 0; j < MAXV4; ++j) kv[j] = slice_ld(kr, toff + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < MAXV4; ++j) {
            v[j] = slice_ld(xr, toff + j * 1024);
            const bool in = vidx + (unsigned)j * 64u < nslice;          // float4s past the slice read as 0: no part of the sums
            const float dx = in ? v[j].x - K : 0.f, dy = in ? v[j].y - K : 0.f;
            const float dz = in ? v[j].z - K : 0.f, dw = in ? v[j].w - K : 0.f;
            s += (dx + dy) + (dz + dw);
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        block_reduce2(s, q, sm);
        GN_T(0);
        unsigned long long* sp = ctl.slots + (size_t)span * cl;
        if (tid == 0) slot_publish(sp + wic, s, q);
        if (NST > 0 && parked) {                        // the previous ticket's parked output leaves under this ticket's rendezvous
#pragma unroll
            for (int j = 0; j < NST; ++j) slice_st(parked_r, toff + j * 1024, stash[j * THREADS + tid]);
            parked = false;
        }
        // fixed-order total of the cl (<= 128) partials, identical in every workgroup of the cluster
        double ds = 0.0, dq = 0.0;
        int got = 1;
        if (tid < cl) { float ps, pq; got = slot_collect(sp + tid, ps, pq, ctl.poll_ticks) ? 1 : 0; ds = (double)ps; dq = (double)pq; }
        const bool arrived = __syncthreads_and(got) != 0;
        GN_T(1);
        // Draw the next ticket only now: a workgroup must never WAIT while it holds a ticket it has not published for
        // (the drawn ticket could belong to the very cluster it waits on).  From here on nothing blocks, and the draw's
        // latency hides under the finishing pass.
        unsigned tnext = 0;
        if (tid == 0) {
            tnext = tk.next(t);
            last_s = cluster_done(ctl, tk, span, cl, tnext) ? 1u : 0u;
        }
        ds = wave_sum_d(ds); dq = wave_sum_d(dq);
        if (lane == 0) { smd[wave * 2] = ds; smd[wave * 2 + 1] = dq; }
        __syncthreads();
        if (last_s) cluster_restore(ctl, sp, span, cl, tid);
        ds = (smd[0] + smd[2]) + (smd[4] + smd[6]);
        dq = (smd[1] + smd[3]) + (smd[5] + smd[7]);
        const double n = (double)cpg * (double)S;
        const double dm = ds / n;                       // mean of x - K
        double var = dq / n - dm * dm;
        if (var < 0.0) var = 0.0;
        float mean = (float)((double)K + dm), rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (!arrived) {
            mean = rstd = __builtin_nanf("");
            if (tid == 0) report_timeout(ctl.status);
        }
        if (wic == 0 && tid == 0) { mean_rstd[span * 2] = mean; mean_rstd[span * 2 + 1] = rstd; }
        float a, sh;
        gn_affine(mean, rstd, gamma[c], beta[c], a, sh);
        const auto yr = slice_rsrc(y + base + (size_t)v0 * 4, v1 - v0);
#pragma unroll
        for (int j = 0; j < MAXV4; ++j) {
            float4 o = v[j];
            o.x = __builtin_fmaf(o.x, a, sh); o.y = __builtin_fmaf(o.y, a, sh);
            o.z = __builtin_fmaf(o.z, a, sh); o.w = __builtin_fmaf(o.w, a, sh);
            if (SKIP) { const float4 k = kv[j]; o.x += k.x; o.y += k.y; o.z += k.z; o.w += k.w; }
            if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (j < NST && stash_on) stash[j * THREADS + tid] = o;
            else slice_st(yr, toff + j * 1024, o);
        }
        if (stash_on) { parked = true; parked_r = yr; }
        if (tid == 0) tick_s = tnext;
        __syncthreads();          // tick_s; smd / sm are reused by the next span
        t = tick_s;
        GN_T(2);
#ifdef GN_PROFILE
        if (tid == 0) atomicAdd(&gn_prof[3], 1ull);
#endif
    }
    if (NST > 0 && parked) {                            // the last ticket's parked part
#pragma unroll
        for (int j = 0; j < NST; ++j) slice_st(parked_r, toff + j * 1024, stash[j * THREADS + tid]);
    }
}

// MASK: 0 = no ReLU, 1 = mask from the forward output y, 2 = mask recomputed as fma(x, a, sh) > 0 (forward without skip)
// GSKIP: 0 = none, 1 = write the masked gradient to gskip (the residual operand's gradient; where several gradients meet
// at one tensor they are summed by ecm_sum_n or in a data-gradient epilogue, see ops.fork / ops._fork_out).
template <int MASK, int GSKIP>
__global__ __launch_bounds__(THREADS, ECM_GN_BWD_OCC) void gn_fused_bwd(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ y, const float* __restrict__ gy,
                                                           float* __restrict__ gx, float* __restrict__ gskip,
                                                           float* __restrict__ chan, FusedCtl ctl, int C,
                                                           long long S, int cpg, int wpc, int nspans,
                                                           long long v4_per_wg) {
    constexpr int MAXV4 = BWD_MAXV4;
    constexpr int NST = bwd_stash_v4(MASK, GSKIP);                    // deferred stores of gx (see STASH_V4)
    extern __shared__ __attribute__((aligned(16))) float4 stash[];
    __shared__ float sm[2 * THREADS / 64];
    __shared__ double chs[2 * FUSED_MAX_CPG];
    __shared__ unsigned tick_s, last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = cpg * wpc;
    bool parked = false;
    __amdgpu_buffer_rsrc_t parked_r = slice_rsrc(gx, 0);
    constexpr bool stash_on = NST > 0;                   // see gn_fused_fwd
    const long long nv4 = S >> 2;                       // < 2^31 (checked by the host): 32-bit indices within a channel
    const unsigned toff = (unsigned)(wave * MAXV4 * 64 + lane) * 16u;     // wave-contiguous rows of 64 float4 (1 KB)
    const Tickets tk{ctl.ticket, (unsigned)nspans * (unsigned)cl, gridDim.x, ctl.dynamic != 0};
    if (tid == 0) { tick_s = tk.first(); if (tick_s >= tk.total) ticket_drawn(ctl, tk, tick_s); }
    __syncthreads();
    unsigned t = tick_s;
    while (t < tk.total) {
        const int span = (int)(t / (unsigned)cl), wic = (int)(t - (unsigned)span * (unsigned)cl);
        const int cig = wic / wpc, w = wic - cig * wpc;
        const int v0 = (int)((long long)w * v4_per_wg);
        const int v1 = (int)(v0 + v4_per_wg < nv4 ? v0 + v4_per_wg : nv4);
        const int b = span / GROUPS, g = span - b * GROUPS;
        const int c = g * cpg + cig;
        const size_t base = ((size_t)b * C + c) * S;
        const float mean = mean_rstd[span * 2], rstd = mean_rstd[span * 2 + 1];
        const float gm = gamma[c];
        float a = 0.f, sh = 0.f;
        if (MASK == 2) gn_affine(mean, rstd, gm, beta[c], a, sh);
        const auto xr = slice_rsrc(x + base + (size_t)v0 * 4, v1 - v0);
        const auto gr = slice_rsrc(gy + base + (size_t)v0 * 4, v1 - v0);
        const auto yr = slice_rsrc(MASK == 1 ? y + base + (size_t)v0 * 4 : x, MASK == 1 ? v1 - v0 : 0);
        const auto kr = slice_rsrc(GSKIP ? gskip + base + (size_t)v0 * 4 : x, GSKIP ? v1 - v0 : 0);
        float4 xv[MAXV4], gv[MAXV4];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV4; ++j) {
            float4 xx = slice_ld(xr, toff + j * 1024), gg = slice_ld(gr, toff + j * 1024);
            if (MASK == 1) {
                const float4 yv = slice_ld(yr, toff + j * 1024);
                if (!(yv.x > 0.f)) gg.x = 0.f;
                if (!(yv.y > 0.f)) gg.y = 0.f;
                if (!(yv.z > 0.f)) gg.z = 0.f;
                if (!(yv.w > 0.f)) gg.w = 0.f;
            }
            if (MASK == 2) {
                if (!(__builtin_fmaf(xx.x, a, sh) > 0.f)) gg.x = 0.f;
                if (!(__builtin_fmaf(xx.y, a, sh) > 0.f)) gg.y = 0.f;
                if (!(__builtin_fmaf(xx.z, a, sh) > 0.f)) gg.z = 0.f;
                if (!(__builtin_fmaf(xx.w, a, sh) > 0.f)) gg.w = 0.f;
            }
            if (GSKIP == 1) slice_st(kr, toff + j * 1024, gg);
            // keep xhat (x itself is not needed again).  Past the slice x reads 0 => xhat = -mean*rstd, but g = 0 there.
            xx.x = (xx.x - mean) * rstd; xx.y = (xx.y - mean) * rstd;
            xx.z = (xx.z - mean) * rstd; xx.w = (xx.w - mean) * rstd;
            sg += (gg.x + gg.y) + (gg.z + gg.w);
            sgx += (gg.x * xx.x + gg.y * xx.y) + (gg.z * xx.z + gg.w * xx.w);
            xv[j] = xx; gv[j] = gg;
        }
        block_reduce2(sg, sgx, sm);
        unsigned long long* sp = ctl.slots + (size_t)span * cl;
        if (tid == 0) slot_publish(sp + wic, sg, sgx);
        if (NST > 0 && parked) {                        // the previous ticket's parked gx leaves under this ticket's rendezvous
#pragma unroll
            for (int j = 0; j < NST; ++j) slice_st(parked_r, toff + j * 1024, stash[j * THREADS + tid]);
            parked = false;
        }
        // per-channel totals (fixed order): wave k reduces the wpc partials of channels k, k+4, ...
        int got = 1;
        for (int cc = wave; cc < cpg; cc += THREADS / 64) {
            double ds = 0.0, dq = 0.0;
            for (int i = lane; i < wpc; i += 64) {
                float ps, pq;
                if (!slot_collect(sp + cc * wpc + i, ps, pq, ctl.poll_ticks)) got = 0;
                ds += (double)ps; dq += (double)pq;
            }
            ds = wave_sum_d(ds); dq = wave_sum_d(dq);
            if (lane == 0) { chs[cc * 2] = ds; chs[cc * 2 + 1] = dq; }
        }
        const bool arrived = __syncthreads_and(got) != 0;
        unsigned tnext = 0;
        if (tid == 0) {
            tnext = tk.next(t);                        // only after the wait (see gn_fused_fwd)
            last_s = cluster_done(ctl, tk, span, cl, tnext) ? 1u : 0u;
        }
        if (!arrived && tid == 0) report_timeout(ctl.status);
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cpg; ++j) {
            const float gj = gamma[g * cpg + j];
            s1 += gj * (float)chs[j * 2];
            s2 += gj * (float)chs[j * 2 + 1];
        }
        if (w == 0 && tid == 0) {
            chan[((size_t)b * C + c) * 2] = arrived ? (float)chs[cig * 2] : __builtin_nanf("");
            chan[((size_t)b * C + c) * 2 + 1] = (float)chs[cig * 2 + 1];
        }
        const float invn = 1.f / ((float)cpg * (float)S);
        const float k1 = arrived ? s1 * invn : __builtin_nanf(""), k2 = s2 * invn;
        const auto orr = slice_rsrc(gx + base + (size_t)v0 * 4, v1 - v0);
#pragma unroll
        for (int j = 0; j < MAXV4; ++j) {
            float4 o;
            o.x = rstd * (gv[j].x * gm - k1 - xv[j].x * k2);
            o.y = rstd * (gv[j].y * gm - k1 - xv[j].y * k2);
            o.z = rstd * (gv[j].z * gm - k1 - xv[j].z * k2);
            o.w = rstd * (gv[j].w * gm - k1 - xv[j].w * k2);
            if (j < NST && stash_on) stash[j * THREADS + tid] = o;
            else slice_st(orr, toff + j * 1024, o);
        }
        if (stash_on) { parked = true; parked_r = orr; }
        if (tid == 0) tick_s = tnext;
        __syncthreads();          // tick_s; chs / sm are reused by the next span
        t = tick_s;
        if (last_s) cluster_restore(ctl, sp, span, cl, tid);
    }
    if (NST > 0 && parked) {
#pragma unroll
        for (int j = 0; j < NST; ++j) slice_st(parked_r, toff + j * 1024, stash[j * THREADS + tid]);
    }
}

// 64-bit slots [B*32][cl <= FUSED_MAX_CL_RUN] + one 16-byte line for the ticket counter + one `done` word per span, in floats
inline long long fused_scratch_floats(int B) { return (long long)B * GROUPS * FUSED_MAX_CL_RUN * 2 + 4 + (long long)B * GROUPS; }

inline int chunks_of(long long n) { return (int)((n + CHUNK - 1) / CHUNK); }

}  // namespace

extern "C" long long ecm_gn3d_scratch_bytes(int B, int C, long long S) {
    // max over: stats partials [B*32][chunks(cpg*S)][2], bwd partials [B*C][chunks(S)][2] + chan [B*C][2]
    const long long cpg = C / 32 > 0 ? C / 32 : 1;
    const long long a = (long long)B * 32 * chunks_of(cpg * S) * 2;
    const long long b = (long long)B * C * chunks_of(S) * 2 + (long long)B * C * 2;
    const long long f = fused_scratch_floats(B) + (long long)B * C * 2;          // cluster partials + counters, chan
    const long long m = a > b ? a : b;
    return (m > f ? m : f) * (long long)sizeof(float);
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
    hipLaunchKernelGGL(gn_stats_final, dim3((B * GROUPS + 63) / 64), dim3(64), 0, ecm_stream(stream), x, part, mean_rstd,
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

namespace {

// ---- process-wide control of the cluster kernels ---------------------------------------------------------------------
// mode: 1 = cluster kernels with ticket assignment (default; 4 = the same, ALSO on a stream under graph capture), 0 = two-stage kernels only (no inter-workgroup waits at
// all), 2 = cluster kernels with static member ids and 3 = a deliberately undersized grid (both diagnostics: 3 makes
// every cluster time out so the error path can be tested).  Env ECM_GN_CLUSTER_MODE / ECM_GN_POLL_MS preset them.
struct GnControl {
    std::mutex mu;
    int mode = 1;
    unsigned long long poll_ticks = 200000000ull;      // 2 s at 100 MHz
    unsigned* status_host = nullptr;                    // pinned, mapped: the kernels OR a bit in on a timeout
    unsigned* status_dev = nullptr;
    bool env_read = false;
    // Two cluster launches that run CONCURRENTLY (two streams of one process) can starve each other: each keeps workgroups
    // resident that wait for cluster members the other launch's waiting workgroups leave no room for -- both then sit in
    // their bounded waits (found in round 4 by replaying two captured graphs on two streams: ~85 launches x 2 s).  Cluster
    // launches of one process are therefore ORDERED across streams on the host.  As long as every cluster launch of a device
    // comes from one stream nothing is done.  The first launch from another stream waits for the device to drain once (the
    // earlier stream may no longer exist -- HIP aborts on a stale handle, so it is never touched again) and switches the
    // device to multi-stream mode: from then on an event is recorded behind every cluster launch and a launch on a stream
    // other than the previous one's waits for it.  `launch_mu` keeps [wait, launch, record] atomic between threads.  A
    // stream that is being captured into a graph never takes the cluster kernels at all (no such ordering can be recorded
    // against streams outside the capture): it gets the two-stage ones.
    std::mutex launch_mu;
    static constexpr int MAX_DEV = 32;
    hipStream_t last_stream[MAX_DEV] = {};
    bool have_last[MAX_DEV] = {};
    bool multi[MAX_DEV] = {};
    hipEvent_t order_ev[MAX_DEV] = {};
};
GnControl& gn_ctl() { static GnControl c; return c; }

int gn_ctl_init_locked(GnControl& c) {
    if (!c.env_read) {
        c.env_read = true;
        if (const char* m = getenv("ECM_GN_CLUSTER_MODE")) c.mode = atoi(m);
        if (const char* p = getenv("ECM_GN_POLL_MS")) { const long long ms = atoll(p); if (ms > 0) c.poll_ticks = (unsigned long long)ms * 100000ull; }
    }
    if (!c.status_host) {
        void* h = nullptr;
        hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess) return (int)e;
        *static_cast<volatile unsigned*>(h) = 0u;
        void* d = nullptr;
        e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) { (void)hipHostFree(h); return (int)e; }
        c.status_host = static_cast<unsigned*>(h);
        c.status_dev = static_cast<unsigned*>(d);
    }
    return 0;
}

// Launch helpers of the fused kernels; -100 = the shape (or the mode) needs the two-stage path.
struct FusedLaunch { FusedGeom g; FusedCtl ctl; int rc; };

inline FusedLaunch fused_prepare(float* scratch, bool preset, int B, int C, long long S, int maxv4, int resident, hipStream_t st) {
    FusedLaunch L{};
    GnControl& c = gn_ctl();
    int mode;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        L.rc = gn_ctl_init_locked(c);
        if (L.rc) return L;
        mode = c.mode;
        L.ctl.status = c.status_dev;
        L.ctl.poll_ticks = c.poll_ticks;
    }
    L.rc = -100;
    if (mode == 0) return L;
    L.g = fused_geom(B, C, S, maxv4, resident);
    if (!L.g.ok) return L;
    if (mode == 2) L.g.grid -= L.g.grid % L.g.cl;                    // static ids: whole clusters only
    if (mode == 3) L.g.grid = L.g.cl > 1 ? L.g.cl - 1 : 1;           // fault injection: a cluster can never be complete
    if (mode == 3 && L.g.cl == 1) return L;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        bool capturing = false;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
        else capturing = cs != hipStreamCaptureStatusNone;
        // captured: two-stage kernels (see GnControl) -- unless the caller has promised, with mode 4, that graphs holding cluster
        // launches are never replayed concurrently with each other or with eager GroupNorm work (no ordering can be recorded)
        if (capturing && mode != 4) return L;
        int dev = 0;
        if (!capturing && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < GnControl::MAX_DEV) {   // (launch_mu is held by the caller)
            if (c.have_last[dev] && c.last_stream[dev] != st) {
                if (!c.multi[dev]) {
                    c.multi[dev] = true;
                    if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
                    if (hipEventCreateWithFlags(&c.order_ev[dev], hipEventDisableTiming) != hipSuccess) {
                        c.order_ev[dev] = nullptr;
                        (void)hipGetLastError();
                    }
                } else if (c.order_ev[dev]) {
                    if (hipStreamWaitEvent(st, c.order_ev[dev], 0) != hipSuccess) (void)hipGetLastError();
                }
            }
            c.last_stream[dev] = st;
            c.have_last[dev] = true;
        }
    }
    L.ctl.dynamic = mode == 2 ? 0 : 1;
    L.ctl.slots = reinterpret_cast<unsigned long long*>(scratch);
    const size_t nslots = (size_t)L.g.nspans * L.g.cl;
    L.ctl.ticket = reinterpret_cast<unsigned*>(L.ctl.slots + nslots);
    L.ctl.done = L.ctl.ticket + 4;
    L.rc = 0;
    if (!preset) {
        // one memset presets the slots to EMPTY, the ticket counter to 0xFFFFFFFF (first draw wraps to ticket 0) and the
        // spans' `done` words; exchange memory kept by the caller (ecm_gn3d_*_p) is in that state already
        const hipError_t e = hipMemsetAsync(L.ctl.slots, 0xff, nslots * sizeof(unsigned long long) + 16 + (size_t)L.g.nspans * 4, st);
        L.rc = e == hipSuccess ? 0 : (int)e;
    }
    return L;
}

// behind a cluster launch (launch_mu held): in multi-stream mode, the event the next launch on another stream waits for
inline void fused_launched(hipStream_t st) {
    GnControl& c = gn_ctl();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs != hipStreamCaptureStatusNone) return;           // (mode 4: a captured cluster launch takes no part in the ordering)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GnControl::MAX_DEV || !c.multi[dev]) return;
    if (!c.order_ev[dev]) {                                 // creation failed earlier: fall back to draining the device
        (void)hipDeviceSynchronize();
        return;
    }
    if (hipEventRecord(c.order_ev[dev], st) != hipSuccess) (void)hipGetLastError();
}

template <bool RELU, bool SKIP>
int launch_fused_fwd(const float* x, const float* gamma, const float* beta, const float* skip, float* y, float* mean_rstd,
                     float* scratch, bool preset, int B, int C, long long S, float eps, hipStream_t st) {
    auto kern = gn_fused_fwd<RELU, SKIP>;
    constexpr int MAXV4 = SKIP ? FWDS_MAXV4 : FWD_MAXV4;
    constexpr int lds = (STASH_V4 < MAXV4 ? STASH_V4 : MAXV4) * THREADS * 16;
    static int resident = -1;
    std::lock_guard<std::mutex> order(gn_ctl().launch_mu);
    if (lds > 0 && ecm_allow_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return ECM_EINVAL;
    if (resident < 0) resident = resident_workgroups(kern, lds);
    const FusedLaunch L = fused_prepare(scratch, preset, B, C, S, MAXV4, resident, st);
    if (L.rc) return L.rc;
    hipLaunchKernelGGL(kern, dim3(L.g.grid), dim3(THREADS), lds, st, x, gamma, beta, skip, y, mean_rstd, L.ctl, C,
                       S, L.g.cpg, L.g.wpc, L.g.nspans, L.g.v4_per_wg, eps);
    const int rc = ECM_LAUNCH_RESULT();
    fused_launched(st);
    return rc;
}

template <int MASK, int GSKIP>
int launch_fused_bwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* y,
                     const float* gy, float* gx, float* gskip, float* chan, float* scratch, bool preset, int B, int C, long long S,
                     hipStream_t st) {
    auto kern = gn_fused_bwd<MASK, GSKIP>;
    constexpr int lds = bwd_stash_v4(MASK, GSKIP) * THREADS * 16;
    static int resident = -1;
    std::lock_guard<std::mutex> order(gn_ctl().launch_mu);
    if (lds > 0 && ecm_allow_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess) return ECM_EINVAL;
    if (resident < 0) resident = resident_workgroups(kern, lds);
    const FusedLaunch L = fused_prepare(scratch, preset, B, C, S, BWD_MAXV4, resident, st);
    if (L.rc) return L.rc;
    hipLaunchKernelGGL(kern, dim3(L.g.grid), dim3(THREADS), lds, st, x, mean_rstd, gamma, beta, y, gy, gx, gskip,
                       chan, L.ctl, C, S, L.g.cpg, L.g.wpc, L.g.nspans, L.g.v4_per_wg);
    const int rc = ECM_LAUNCH_RESULT();
    fused_launched(st);
    return rc;
}

// A cluster launch that timed out earlier leaves the sticky word set: every later GroupNorm call fails until cleared.
inline int gn_pending_error() {
    GnControl& c = gn_ctl();
    std::lock_guard<std::mutex> lock(c.mu);
    return (c.status_host && *static_cast<volatile unsigned*>(c.status_host)) ? ECM_EASYNC : 0;
}

}  // namespace

#ifdef GN_PROFILE
extern "C" int ecm_gn3d_profile(unsigned long long* out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(gn_prof), 4 * sizeof(unsigned long long)) != hipSuccess) return ECM_EINVAL;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(gn_prof), z, sizeof(z)); }
    return 0;
}
#endif

extern "C" int ecm_gn3d_cluster_mode(int mode) {
    GnControl& c = gn_ctl();
    std::lock_guard<std::mutex> lock(c.mu);
    c.env_read = true;                       // an explicit call overrides the environment preset
    const int old = c.mode;
    if (mode >= 0 && mode <= 4) c.mode = mode;
    return old;
}

extern "C" int ecm_gn3d_poll_ms(int ms) {
    GnControl& c = gn_ctl();
    std::lock_guard<std::mutex> lock(c.mu);
    const int old = (int)(c.poll_ticks / 100000ull);
    if (ms > 0) c.poll_ticks = (unsigned long long)ms * 100000ull;
    return old;
}

extern "C" int ecm_async_status(int clear) {
    GnControl& c = gn_ctl();
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.status_host) return 0;
    volatile unsigned* w = static_cast<volatile unsigned*>(c.status_host);
    const unsigned v = *w;
    if (clear) *w = 0u;
    return v ? ECM_EASYNC : 0;
}

namespace {
// cluster: the exchange memory of the one-pass kernels (slots, ticket counter, per-span counters); preset: it already holds
// the all-ones pattern (kept by the caller across calls, see ecm_gn3d_fwd_p) -- otherwise it is memset here.
int gn_fwd_impl(const float* x, const float* gamma, const float* beta, const float* skip, float* y, float* mean_rstd,
                void* scratch, long long scratch_bytes, float* cluster, bool preset, int B, int C, long long S, int relu,
                float eps, void* stream) {
    if (C % GROUPS != 0 || (long long)B * C > 65535) return ECM_EUNSUP;
    if (y == x || y == skip) return ECM_EINVAL;           // not in place: a span's pivot samples are read by all its workgroups
    if (scratch_bytes < ecm_gn3d_scratch_bytes(B, C, S)) return ECM_ESCRATCH;
    if (const int pe = gn_pending_error()) return pe;
    hipStream_t st = ecm_stream(stream);
    int rc;
    if (relu && skip) rc = launch_fused_fwd<true, true>(x, gamma, beta, skip, y, mean_rstd, cluster, preset, B, C, S, eps, st);
    else if (relu) rc = launch_fused_fwd<true, false>(x, gamma, beta, skip, y, mean_rstd, cluster, preset, B, C, S, eps, st);
    else if (skip) rc = launch_fused_fwd<false, true>(x, gamma, beta, skip, y, mean_rstd, cluster, preset, B, C, S, eps, st);
    else rc = launch_fused_fwd<false, false>(x, gamma, beta, skip, y, mean_rstd, cluster, preset, B, C, S, eps, st);
    if (rc != -100) return rc;
    rc = ecm_gn3d_stats(x, mean_rstd, scratch, scratch_bytes, B, C, S, eps, stream);
    if (rc) return rc;
    return ecm_gn3d_apply(x, mean_rstd, gamma, beta, skip, y, B, C, S, relu, stream);
}

int gn_bwd_impl(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* y,
                const float* gy, float* gx, float* gskip, float* ggamma, float* gbeta, void* scratch,
                long long scratch_bytes, float* cluster, bool preset, float* chan_fused, int B, int C, long long S, int relu,
                void* stream) {
    if (C % GROUPS != 0 || (long long)B * C > 65535) return ECM_EUNSUP;
    if (gx == x || gx == gy || gx == y || (gskip && (gskip == x || gskip == gy || gskip == gx))) return ECM_EINVAL;   // not in place
    if (scratch_bytes < ecm_gn3d_scratch_bytes(B, C, S)) return ECM_ESCRATCH;
    if (const int pe = gn_pending_error()) return pe;
    const int mask = !relu ? 0 : (y ? 1 : 2);
    hipStream_t st = ecm_stream(stream);
    float* sc = static_cast<float*>(scratch);
    {
        float* chan = chan_fused;
        int rc;
        if (mask == 0) rc = gskip ? launch_fused_bwd<0, 1>(x, mean_rstd, gamma, beta, y, gy, gx, gskip, chan, cluster, preset, B, C, S, st)
                                  : launch_fused_bwd<0, 0>(x, mean_rstd, gamma, beta, y, gy, gx, gskip, chan, cluster, preset, B, C, S, st);
        else if (mask == 1) rc = gskip ? launch_fused_bwd<1, 1>(x, mean_rstd, gamma, beta, y, gy, gx, gskip, chan, cluster, preset, B, C, S, st)
                                       : launch_fused_bwd<1, 0>(x, mean_rstd, gamma, beta, y, gy, gx, gskip, chan, cluster, preset, B, C, S, st);
        else rc = gskip ? launch_fused_bwd<2, 1>(x, mean_rstd, gamma, beta, y, gy, gx, gskip, chan, cluster, preset, B, C, S, st)
                        : launch_fused_bwd<2, 0>(x, mean_rstd, gamma, beta, y, gy, gx, gskip, chan, cluster, preset, B, C, S, st);
        if (rc == 0) {
            hipLaunchKernelGGL(gn_bwd_params, dim3((C + 63) / 64), dim3(64), 0, st, chan, ggamma, gbeta, B, C);
            return ECM_LAUNCH_RESULT();
        }
        if (rc != -100) return rc;
    }
    const int nchunks = chunks_of(S);
    float* part = sc;
    float* chan = part + (size_t)B * C * nchunks * 2;
    dim3 g1(nchunks, B * C), block(THREADS);
    if (mask == 0) hipLaunchKernelGGL(gn_bwd_partial<0>, g1, block, 0, st, x, mean_rstd, gamma, beta, y, gy, gskip, part, C, S, nchunks);
    else if (mask == 1) hipLaunchKernelGGL(gn_bwd_partial<1>, g1, block, 0, st, x, mean_rstd, gamma, beta, y, gy, gskip, part, C, S, nchunks);
    else hipLaunchKernelGGL(gn_bwd_partial<2>, g1, block, 0, st, x, mean_rstd, gamma, beta, y, gy, gskip, part, C, S, nchunks);
    hipLaunchKernelGGL(gn_bwd_final_chan, dim3((B * C + 63) / 64), dim3(64), 0, st, part, chan, B * C, nchunks);
    hipLaunchKernelGGL(gn_bwd_params, dim3((C + 63) / 64), dim3(64), 0, st, chan, ggamma, gbeta, B, C);
    long long per = (S + THREADS * 4 - 1) / (THREADS * 4);
    dim3 g2((int)(per < 64 ? per : 64), B * C);
    if (mask == 0) hipLaunchKernelGGL(gn_bwd_apply<0>, g2, block, 0, st, x, mean_rstd, gamma, beta, y, gy, chan, gx, C, S);
    else if (mask == 1) hipLaunchKernelGGL(gn_bwd_apply<1>, g2, block, 0, st, x, mean_rstd, gamma, beta, y, gy, chan, gx, C, S);
    else hipLaunchKernelGGL(gn_bwd_apply<2>, g2, block, 0, st, x, mean_rstd, gamma, beta, y, gy, chan, gx, C, S);
    return ECM_LAUNCH_RESULT();
}
}  // namespace

extern "C" int ecm_gn3d_fwd(const float* x, const float* gamma, const float* beta, const float* skip, float* y,
                            float* mean_rstd, void* scratch, long long scratch_bytes, int B, int C, long long S,
                            int relu, float eps, void* stream) {
    ECM_CHECK_ARG(x && gamma && beta && y && mean_rstd && scratch && B > 0 && C > 0 && S > 0);
    return gn_fwd_impl(x, gamma, beta, skip, y, mean_rstd, scratch, scratch_bytes, static_cast<float*>(scratch), false, B, C, S,
                       relu, eps, stream);
}

extern "C" int ecm_gn3d_bwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* y,
                            const float* gy, float* gx, float* gskip, float* ggamma, float* gbeta, void* scratch,
                            long long scratch_bytes, int B, int C, long long S, int relu, void* stream) {
    ECM_CHECK_ARG(x && mean_rstd && gamma && gy && gx && ggamma && gbeta && scratch && B > 0 && C > 0 && S > 0);
    ECM_CHECK_ARG(!relu || y || beta);          // the ReLU mask comes from y, or is recomputed from x with beta
    float* sc = static_cast<float*>(scratch);
    return gn_bwd_impl(x, mean_rstd, gamma, beta, y, gy, gx, gskip, ggamma, gbeta, scratch, scratch_bytes, sc, false,
                       sc + fused_scratch_floats(B), B, C, S, relu, stream);      // chan lives behind the cluster memory
}

extern "C" long long ecm_gn3d_cluster_bytes(int B) { return B > 0 ? fused_scratch_floats(B) * (long long)sizeof(float) : 0; }

extern "C" int ecm_gn3d_cluster_preset(void* cluster, long long cluster_bytes, void* stream) {
    ECM_CHECK_ARG(cluster && cluster_bytes > 0);
    return (int)hipMemsetAsync(cluster, 0xff, (size_t)cluster_bytes, ecm_stream(stream));
}

extern "C" int ecm_gn3d_fwd_p(const float* x, const float* gamma, const float* beta, const float* skip, float* y,
                              float* mean_rstd, void* scratch, long long scratch_bytes, void* cluster, long long cluster_bytes,
                              int B, int C, long long S, int relu, float eps, void* stream) {
    ECM_CHECK_ARG(x && gamma && beta && y && mean_rstd && scratch && cluster && B > 0 && C > 0 && S > 0);
    if (cluster_bytes < ecm_gn3d_cluster_bytes(B)) return ECM_ESCRATCH;
    return gn_fwd_impl(x, gamma, beta, skip, y, mean_rstd, scratch, scratch_bytes, static_cast<float*>(cluster), true, B, C, S,
                       relu, eps, stream);
}

extern "C" int ecm_gn3d_bwd_p(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* y,
                              const float* gy, float* gx, float* gskip, float* ggamma, float* gbeta, void* scratch,
                              long long scratch_bytes, void* cluster, long long cluster_bytes, int B, int C, long long S,
                              int relu, void* stream) {
    ECM_CHECK_ARG(x && mean_rstd && gamma && gy && gx && ggamma && gbeta && scratch && cluster && B > 0 && C > 0 && S > 0);
    ECM_CHECK_ARG(!relu || y || beta);
    if (cluster_bytes < ecm_gn3d_cluster_bytes(B)) return ECM_ESCRATCH;
    return gn_bwd_impl(x, mean_rstd, gamma, beta, y, gy, gx, gskip, ggamma, gbeta, scratch, scratch_bytes,
                       static_cast<float*>(cluster), true, static_cast<float*>(scratch), B, C, S, relu, stream);
}
