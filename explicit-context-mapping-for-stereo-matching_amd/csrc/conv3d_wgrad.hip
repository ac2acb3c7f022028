// Weight gradient of the 3x3x3 Conv3d (stride 1|2, pad 1) on the fp32 matrix cores:
//   gw[co,ci,tap] = sum_{b,o} gy[b,co,o] * x[b,ci,o*S+tap-1]
// (autograd of the reference's nn.Conv3d, cmfsm.py:52-57; with x:=gy_big, gy:=x_small it is also the weight
// gradient of ConvTranspose3d, cmfsm.py:262-281).
//
// GEMM view per tap: D[co][ci] += sum_voxel A[co][voxel] * B[voxel][ci]  -- the reduction runs over voxels.
//   A = gy : lane l holds gy[co = l&31][voxel = v + (l>>5)]   (LDS image [co][voxels], odd row stride)
//   B = x  : lane l holds x[ci = l&31][voxel shifted by the tap]  (LDS image [ci][halo tile], odd stride)
// so a k-step is two x-adjacent output voxels and every tap re-reads the same staged halo tile at a
// different immediate offset.  Workgroups are persistent over spatial tiles (accumulators stay in
// registers: 27 taps x 16 regs split over the 4 waves), write one partial [Co][Ci][27] each, and a
// second kernel sums the partials in a fixed order (deterministic, no float atomics).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CT = 32;        // channel tile (both co and ci)
constexpr int TWV = 16;       // output voxels per tile row

template <int STRIDE, int TD, int TH>
struct WgCfg {
    static constexpr int ID = (TD - 1) * STRIDE + 3, IH = (TH - 1) * STRIDE + 3, IW = (TWV - 1) * STRIDE + 3;
    static constexpr int RS = IW;
    static constexpr int XCH = ID * IH * RS;
    static constexpr int XSTR = (XCH % 2 == 0) ? XCH + 1 : XCH;     // odd => 32 lanes (ci) hit 32 banks
    static constexpr int NV = TD * TH * TWV;
    static constexpr int GSTR = NV + 1;                              // odd
    static constexpr int LDS_FLOATS = CT * XSTR + CT * GSTR;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};


// One staged tile: TD*TH*TWV/2 k-steps (two x-adjacent voxels each), taps [T0, T0+7) of this wave.
// The LDS operands of k-step i+1 are read before the MFMAs of k-step i are issued (explicit register double
// buffer), so with one wave per SIMD the ~100-cycle ds_read latency hides under 7 x 64 cycles of matrix work.
template <int STRIDE, int TD, int TH, int T0>
__device__ __forceinline__ void wg_tile(const float* __restrict__ ga, const float* __restrict__ xb, f32x16 (&acc)[7]) {
    using Cfg = WgCfg<STRIDE, TD, TH>;
    constexpr int IH = Cfg::IH, RS = Cfg::RS;
    constexpr int NT = (T0 + 7 <= 27) ? 7 : 27 - T0;          // taps of this wave (the last wave has 6)
    constexpr int KS = TD * TH * TWV / 2;
    auto a_off = [](int ks) constexpr { return ks * 2; };     // voxel index of the pair: (dz*TH+hy)*TWV + xx == 2*ks
    auto b_off = [](int ks, int t) constexpr {
        const int xx = (ks * 2) % TWV, hy = ((ks * 2) / TWV) % TH, dz = (ks * 2) / (TWV * TH);
        const int tap = T0 + t, kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        return ((dz * STRIDE + kd) * IH + hy * STRIDE + kh) * RS + xx * STRIDE + kw;
    };
    float a_cur = ga[a_off(0)], b_cur[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b_cur[t] = xb[b_off(0, t)];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float a_nxt = 0.f, b_nxt[NT];
        if (ks + 1 < KS) {
            a_nxt = ga[a_off(ks + 1)];
#pragma unroll
            for (int t = 0; t < NT; ++t) b_nxt[t] = xb[b_off(ks + 1, t)];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < NT; ++t) b_cur[t] = b_nxt[t];
    }
}

template <int STRIDE, int TD, int TH>
__global__ __launch_bounds__(256, 1) void conv3d_wgrad_mfma(const float* __restrict__ x, const float* __restrict__ gy,
                                                         float* __restrict__ partial, int B, int Ci, int Co, int D,
                                                         int H, int W, int Do, int Ho, int Wo, int tiles_d, int tiles_h,
                                                         int tiles_w, int ci_tiles) {
    using Cfg = WgCfg<STRIDE, TD, TH>;
    constexpr int ID = Cfg::ID, IH = Cfg::IH, IW = Cfg::IW, RS = Cfg::RS, XSTR = Cfg::XSTR, NV = Cfg::NV,
                  GSTR = Cfg::GSTR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                     // [32 ci][XSTR]
    float* Gs = smem + CT * XSTR;         // [32 co][GSTR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ci0 = (blockIdx.y % ci_tiles) * CT, co0 = (blockIdx.y / ci_tiles) * CT;
    const int t0 = wave * 7;                                   // this wave's taps [t0, t0+nt)
    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const size_t HWi = (size_t)H * W, DHWi = (size_t)D * HWi;
    const size_t HWo = (size_t)Ho * Wo, DHWo = (size_t)Do * HWo;
    const long long ntiles = (long long)B * tiles_d * tiles_h * tiles_w;

    // Register-pipelined staging through buffer descriptors (see conv3d.hip): the loads of the NEXT tile are issued
    // before this tile's MFMA loop; positions outside the volume carry offset 0x80000000 and read back as 0.
    constexpr int NPOSX = ID * IH * IW;
    constexpr int PPX = (NPOSX + 255) / 256;                 // x positions per thread per channel
    constexpr int GPL = 256 / NV;                            // gy channels covered by one 256-thread pass
    static_assert(256 % NV == 0 && CT % GPL == 0, "gy tile must divide the workgroup");
    constexpr int NGRP = CT / GPL;
    float xr[CT * PPX], gr[NGRP];
    const unsigned xplane = (unsigned)DHWi * 4u, gplane = (unsigned)DHWo * 4u;
    auto prefetch = [&](long long tile) {
        long long r = tile;
        const int tw = (int)(r % tiles_w); r /= tiles_w;
        const int th = (int)(r % tiles_h); r /= tiles_h;
        const int td = (int)(r % tiles_d);
        const int b = (int)(r / tiles_d);
        const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TWV;
        const int id0 = od0 * STRIDE - 1, ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;
        unsigned xoff[PPX];
#pragma unroll
        for (int j = 0; j < PPX; ++j) {
            const int p = tid + j * 256;
            int t = p;
            const int xx = t % IW; t /= IW;
            const int hy = t % IH;
            const int dz = t / IH;
            const int gz = id0 + dz, gyy = ih0 + hy, gx = iw0 + xx;
            const bool ok = p < NPOSX && (unsigned)gz < (unsigned)D && (unsigned)gyy < (unsigned)H && (unsigned)gx < (unsigned)W;
            xoff[j] = ok ? (unsigned)(gz * (int)HWi + gyy * W + gx) * 4u : 0x80000000u;
        }
        // ONE descriptor per tensor and sample; the (uniform) channel offset is added to the per-lane offset.  The
        // out-of-volume marker 0x80000000 stays >= num_records after adding any channel offset (< 2^31 bytes per sample).
        const int nci = Ci - ci0 < CT ? Ci - ci0 : CT, nco = Co - co0 < CT ? Co - co0 : CT;
        const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((size_t)b * Ci + ci0) * DHWi), 0,
                                                           (unsigned)nci * xplane, 0x00020000);
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
#pragma unroll
            for (int j = 0; j < PPX; ++j)
                xr[cc * PPX + j] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff[j] + (unsigned)cc * xplane, 0, 0));
        // gy: thread -> (channel within group = tid / NV, voxel = tid % NV)
        const int v = tid % NV, ccl = tid / NV;
        const int xx = v % TWV, hy = (v / TWV) % TH, dz = v / (TWV * TH);
        const int od = od0 + dz, oh = oh0 + hy, ow = ow0 + xx;
        const bool vok = od < Do && oh < Ho && ow < Wo;
        const unsigned goff = vok ? (unsigned)(od * (int)HWo + oh * Wo + ow) * 4u + (unsigned)ccl * gplane : 0x80000000u;
        const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy + ((size_t)b * Co + co0) * DHWo), 0,
                                                           (unsigned)nco * gplane, 0x00020000);
#pragma unroll
        for (int g = 0; g < NGRP; ++g)
            gr[g] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(grs, goff + (unsigned)(g * GPL) * gplane, 0, 0));
    };
    if ((long long)blockIdx.x < ntiles) prefetch(blockIdx.x);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
#pragma unroll
            for (int j = 0; j < PPX; ++j) {
                const int p = tid + j * 256;
                if (p < NPOSX) Xs[cc * XSTR + p] = xr[cc * PPX + j];       // [ci][dz][hy][xx], odd channel stride
            }
#pragma unroll
        for (int g = 0; g < NGRP; ++g) Gs[(g * GPL + tid / NV) * GSTR + tid % NV] = gr[g];
        __syncthreads();
        if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);
        const float* ga = Gs + l31 * GSTR + half;
        const float* xb = Xs + l31 * XSTR + half * STRIDE;
        switch (wave) {          // wave-uniform: makes every tap offset a compile-time immediate
            case 0: wg_tile<STRIDE, TD, TH, 0>(ga, xb, acc); break;
            case 1: wg_tile<STRIDE, TD, TH, 7>(ga, xb, acc); break;
            case 2: wg_tile<STRIDE, TD, TH, 14>(ga, xb, acc); break;
            default: wg_tile<STRIDE, TD, TH, 21>(ga, xb, acc); break;
        }
    }
    // ---- write this workgroup's partial: partial[blockIdx.x][co][ci][tap] ------------------------------
    float* pp = partial + (size_t)blockIdx.x * Co * Ci * 27;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int tap = t0 + t;
        if (tap >= 27) continue;
        const int ci = ci0 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (co < Co && ci < Ci) pp[((size_t)co * Ci + ci) * 27 + tap] = acc[t][i];
        }
    }
}

__global__ void wgrad_reduce(const float* __restrict__ partial, float* __restrict__ gw, int n, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += partial[(size_t)p * n + i];
    gw[i] = s;
}

inline int wgrad_workers(int Ci, int Co, long long ntiles) {
    const int ytiles = ((Ci + CT - 1) / CT) * ((Co + CT - 1) / CT);
    long long p = 256 / ytiles;                 // one persistent workgroup per CU in total (the kernel runs 1 block/CU)
    if (p < 1) p = 1;
    if (p > ntiles) p = ntiles;
    return (int)p;
}

template <int STRIDE, int TD, int TH>
int launch_wgrad(const float* x, const float* gy, float* gw, float* partial, int B, int Ci, int Co, int D, int H, int W,
                 hipStream_t st) {
    using Cfg = WgCfg<STRIDE, TD, TH>;
    const int Do = (D - 1) / STRIDE + 1, Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const int tiles_d = (Do + TD - 1) / TD, tiles_h = (Ho + TH - 1) / TH, tiles_w = (Wo + TWV - 1) / TWV;
    const long long ntiles = (long long)B * tiles_d * tiles_h * tiles_w;
    const int ci_tiles = (Ci + CT - 1) / CT, co_tiles = (Co + CT - 1) / CT;
    const int P = wgrad_workers(Ci, Co, ntiles);
    auto kern = conv3d_wgrad_mfma<STRIDE, TD, TH>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(P, ci_tiles * co_tiles), dim3(256), Cfg::LDS_BYTES, st, x, gy, partial, B, Ci, Co, D, H,
                       W, Do, Ho, Wo, tiles_d, tiles_h, tiles_w, ci_tiles);
    const int n = Co * Ci * 27;
    hipLaunchKernelGGL(wgrad_reduce, dim3((n + 255) / 256), dim3(256), 0, st, partial, gw, n, P);
    return ECM_LAUNCH_RESULT();
}

inline long long ntiles_for(int B, int D, int H, int W, int stride) {
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int TD = stride == 1 ? 2 : 1, TH = stride == 1 ? 8 : 4;
    return (long long)B * ((Do + TD - 1) / TD) * ((Ho + TH - 1) / TH) * ((Wo + TWV - 1) / TWV);
}

}  // namespace

extern "C" long long ecm_conv3d_wgrad_scratch_bytes(int B, int Ci, int Co, int D, int H, int W, int stride) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return 0;
    return (long long)wgrad_workers(Ci, Co, ntiles_for(B, D, H, W, stride)) * Co * Ci * 27 * (long long)sizeof(float);
}

extern "C" int ecm_conv3d_k3_wgrad(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes,
                                   int B, int Ci, int Co, int D, int H, int W, int stride, void* stream) {
    ECM_CHECK_ARG(x && gy && gw && scratch && B > 0 && Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0);
    if (stride != 1 && stride != 2) return ECM_EUNSUP;
    if ((long long)D * H * W * 4 * 32 >= 0x7fffffffLL) return ECM_EUNSUP;  // 32-bit buffer offsets over a 32-channel tile
    if (scratch_bytes < ecm_conv3d_wgrad_scratch_bytes(B, Ci, Co, D, H, W, stride)) return ECM_ESCRATCH;
    float* partial = static_cast<float*>(scratch);
    hipStream_t st = ecm_stream(stream);
    if (stride == 1) return launch_wgrad<1, 2, 8>(x, gy, gw, partial, B, Ci, Co, D, H, W, st);
    return launch_wgrad<2, 1, 4>(x, gy, gw, partial, B, Ci, Co, D, H, W, st);
}
