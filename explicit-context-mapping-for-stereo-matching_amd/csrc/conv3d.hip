// a5-a7: 3x3x3 Conv3d (pad 1, stride 1|2, no bias) as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate -> bit-for-bit an fmaf chain).
// Reference: nn.Conv3d inside convbn_3d (cmfsm.py:49-58), call sites dres0/1 604-613, hourglass 244-259,
// classif 621-634.
//
// GEMM view:  D[co][voxel] += sum_k A[co][k] * B[k][voxel],  k = (tap, ci).
//   A = weights   : lane l holds W[co = l&31][k = l>>5]          (LDS image [tap][ci][co], conflict-free)
//   B = activations: lane l holds X[k = l>>5][voxel = l&31]       (32 consecutive x of one row -> the
//        reference's own NCDHW layout is already the MFMA operand layout: consecutive lanes read
//        consecutive LDS words of the staged halo tile, no transposition anywhere)
//   D: lane holds voxel l&31, 16 regs = 16 output channels -> every store is two 128-B row segments.
// One workgroup (4 waves) owns a TD x TH x 32 output tile for ALL output channels and walks the
// input channels in chunks of CIC: stage halo tile + weight slice in LDS, then 27*CIC/2 k-steps.
// Each wave owns NT = TD*TH/4 rows of 32 voxels: one A fragment feeds NT MFMAs, one B fragment
// feeds CO_TILES MFMAs; every LDS offset inside the chunk loop is a compile-time immediate.
#include "common.h"

#ifdef CV_PROFILE
__device__ unsigned long long cv_prof[4 * 8];     // [wave][phase] cycles of workgroup 0; debugging aid only
#define CV_T(i) do { const unsigned long long now_ = clock64(); prof[i] += now_ - last; last = now_; } while (0)
#else
#define CV_T(i) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TW = 32;

// KD = 3: the 3x3x3 Conv3d (the 2-D layers have their own family in conv2d.hip, the stride-1 layers run by default on the
// Winograd kernel of conv_wino.hip).
template <int CO_TILES, int STRIDE, int TD, int TH, int CIC, int KD = 3>
struct ConvCfg {
    static constexpr int NTAPS = 9 * KD;
    static constexpr int ID = (TD - 1) * STRIDE + KD;
    static constexpr int IH = (TH - 1) * STRIDE + 3;
    static constexpr int IW = (TW - 1) * STRIDE + 3;
    static constexpr int RS = IW;                       // LDS row stride (floats)
    static constexpr int ROWS = TD * TH;
    static constexpr int NT = ROWS / 4;                 // rows per wave
    static constexpr int COP = CO_TILES * 32;
    static constexpr int XS_FLOATS = CIC * ID * IH * RS;
    static constexpr int WS_FLOATS = NTAPS * CIC * COP;
    static constexpr int LDS_BYTES = (XS_FLOATS + 2 * WS_FLOATS) * 4;     // weight slice is double-buffered (LDS-DMA)
    static_assert(ROWS % 4 == 0, "rows must split over 4 waves");
    static_assert((NT <= TH && TH % NT == 0) || (NT % TH == 0), "wave rows must tile (dz,hy) statically");
    static_assert(CIC % 2 == 0, "k-step is 2 channels");
};

template <int CO_TILES, int STRIDE, int TD, int TH, int CIC, int KD = 3>
__global__ __launch_bounds__(256, 2) void conv3d_k3_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                      float* __restrict__ y, int Ci, int Co, int D, int H, int W,
                                                      int Do, int Ho, int Wo, int tiles_d, int tiles_h, int tiles_w) {
    using Cfg = ConvCfg<CO_TILES, STRIDE, TD, TH, CIC, KD>;
    constexpr int ID = Cfg::ID, IH = Cfg::IH, IW = Cfg::IW, RS = Cfg::RS, NT = Cfg::NT, COP = Cfg::COP, NTAPS = Cfg::NTAPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [CIC][ID][IH][RS]
    float* Ws = smem + Cfg::XS_FLOATS;      // 2 x [27][CIC][COP]

    // tile decode: one contiguous run of tiles per XCD (ecm_xcd_tile), depth fastest inside it -- depth neighbours share
    // 2 of their ID input planes (the largest halo overlap), so they should meet in the same L2 at about the same time.
    // Measured (32->32, B=4, rocprofv3 FETCH_SIZE): L2-miss traffic 1370 -> 767 MB per launch, time 2.95 -> 2.91 ms.
    int bid = ecm_xcd_tile(blockIdx.x, gridDim.x);
    const int td = bid % tiles_d; bid /= tiles_d;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h;
    const int b = bid / tiles_h;
    const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
    const int id0 = od0 * STRIDE - KD / 2, ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
#ifdef CV_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = clock64();
#endif
    // this wave's first row -> (dz0, hy0); later rows are compile-time offsets from it
    const int row0 = wave * NT;
    const int dz0 = (NT <= TH) ? row0 / TH : (row0 / TH);
    const int hy0 = (NT <= TH) ? row0 % TH : 0;
    const int xbase = ((half * ID + dz0 * STRIDE) * IH + hy0 * STRIDE) * RS + l31 * STRIDE;
    const int wbase = half * COP + l31;

    f32x16 acc[NT][CO_TILES];
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int ct = 0; ct < CO_TILES; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][ct][i] = 0.f;

    const size_t HWi = (size_t)H * W, DHWi = (size_t)D * HWi;
    const float* xb = x + (size_t)b * Ci * DHWi;

    // Staging is software-pipelined through registers: the global loads of chunk c+1 are issued before the
    // MFMA loop of chunk c and only waited for when they are written to LDS, so HBM/L2 latency hides under
    // ~27k cycles of matrix work instead of being paid 32 times per chunk.
    // Channel-major staging: a thread owns PP fixed (dz,hy,xx) positions of the 3-D halo window (offsets and bounds
    // computed once per tile) and walks the CIC channel planes of a chunk, whose base addresses are wave-uniform.
    constexpr int NPOS = ID * IH * IW;
    constexpr int PP = (NPOS + 255) / 256;                         // positions per thread
    constexpr int NX = CIC * PP;
    constexpr int NWQ = (NTAPS * CIC * COP / 4 + 255) / 256;       // weight float4s per thread
    float xr[NX];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // Loads go through buffer descriptors (one per channel plane, built from wave-uniform scalars): 32-bit per-lane
    // byte offsets, and the hardware range check returns 0 for the 0x80000000 offset given to every position outside
    // the volume -- zero padding costs no compare, no select and no 64-bit address math.
    unsigned posoff[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int p = tid + j * 256;
        int t = p;
        const int xx = t % IW; t /= IW;
        const int hy = t % IH;
        const int dz = t / IH;
        const int gz = id0 + dz, gy = ih0 + hy, gx = iw0 + xx;
        const bool ok = p < NPOS && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        posoff[j] = ok ? (unsigned)(gz * (int)HWi + gy * W + gx) * 4u : 0x80000000u;
    }
    const unsigned plane_bytes = (unsigned)DHWi * 4u;
    auto prefetch = [&](int c0, float* wdst) {                // the chunk's weight slice
        // global -> LDS directly (global_load_lds_dwordx4: no VGPRs, lands at wave base + lane*16)
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            const int e = tid + i * 256;
            if (e < NTAPS * CIC * COP / 4) {
                const int tap = e / (CIC * COP / 4), r = e - tap * (CIC * COP / 4);
                const float* src = wp + ((size_t)tap * Ci + c0) * COP + (size_t)r * 4;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(wdst + (wave_u * 64 + i * 256) * 4), 16, 0, 0);
            }
        }
    };
    // the halo loads of a chunk, one at a time (i = cc * PP + j): inside the MFMA loop they are issued a few per tap --
    // issued as one burst they fill the memory pipeline's queue and the wave sits on it with the matrix core idle
    auto prefetch_x = [&](int c0, int i) {
        const int cc = i / PP, j = i % PP;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb + (size_t)(c0 + cc) * DHWi), 0, plane_bytes,
                                                            0x00020000);
        xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, posoff[j], 0, 0));
    };
    prefetch(0, Ws);
#pragma unroll
    for (int i = 0; i < NX; ++i) prefetch_x(0, i);
    CV_T(0);
    int buf = 0;
    for (int c0 = 0; c0 < Ci; c0 += CIC, buf ^= 1) {
        __syncthreads();                                   // previous chunk's LDS reads are done
        CV_T(1);
#pragma unroll
        for (int cc = 0; cc < CIC; ++cc)
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const int p = tid + j * 256;
                if (p < NPOS) Xs[cc * NPOS + p] = xr[cc * PP + j];          // [cc][dz][hy][xx], RS == IW
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight DMA for this chunk has landed
        CV_T(2);
        __syncthreads();
        CV_T(3);
        const float* Wc = Ws + buf * Cfg::WS_FLOATS;
        const bool more = c0 + CIC < Ci;
        if (more) prefetch(c0 + CIC, Ws + (buf ^ 1) * Cfg::WS_FLOATS);            // weights: in flight during the MFMA loop below
        CV_T(4);
        constexpr int LPT = (NX + NTAPS - 1) / NTAPS;                              // halo loads per tap
        // ---- 27 * CIC/2 k-steps ---------------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int kd = KD == 3 ? tap / 9 : 0, kh = (tap / 3) % 3, kw = tap % 3;
            if (more) {
#pragma unroll
                for (int q = 0; q < LPT; ++q)
                    if (tap * LPT + q < NX) prefetch_x(c0 + CIC, tap * LPT + q);
            }
#pragma unroll
            for (int kk = 0; kk < CIC / 2; ++kk) {
                float a[CO_TILES];
#pragma unroll
                for (int ct = 0; ct < CO_TILES; ++ct) a[ct] = Wc[wbase + (tap * CIC + kk * 2) * COP + ct * 32];
#pragma unroll
                for (int r = 0; r < NT; ++r) {
                    const int dz = (NT <= TH) ? 0 : r / TH;
                    const int hy = (NT <= TH) ? r : r % TH;
                    const float bv = Xs[xbase + (((kk * 2) * ID + dz * STRIDE + kd) * IH + hy * STRIDE + kh) * RS + kw];
#pragma unroll
                    for (int ct = 0; ct < CO_TILES; ++ct)
                        acc[r][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct], bv, acc[r][ct], 0, 0, 0);
                }
            }
        }
        CV_T(5);
    }

    // ---- epilogue: D[co][voxel] -> y[b,co,od,oh,ow] ---------------------------------------------
    const size_t HWo = (size_t)Ho * Wo, DHWo = (size_t)Do * HWo;
    float* yb = y + (size_t)b * Co * DHWo;
    const int ow = ow0 + l31;
#pragma unroll
    for (int r = 0; r < NT; ++r) {
        const int dz = dz0 + ((NT <= TH) ? 0 : r / TH);
        const int hy = hy0 + ((NT <= TH) ? r : r % TH);
        const int od = od0 + dz, oh = oh0 + hy;
        if (od >= Do || oh >= Ho || ow >= Wo) continue;
        float* yp = yb + (size_t)od * HWo + (size_t)oh * Wo + ow;
#pragma unroll
        for (int ct = 0; ct < CO_TILES; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                if (co < Co) yp[(size_t)co * DHWo] = acc[r][ct][i];      // (plain: a non-temporal hint measured +-0 here, round 4)
            }
    }
#ifdef CV_PROFILE
    CV_T(6);
    if (blockIdx.x == 700 && lane == 0)
        for (int i = 0; i < 8; ++i) cv_prof[wave * 8 + i] = prof[i];
#endif
}

// Conv3d weight [Co,Ci,27] -> [27][Ci][COP] (zero padded co), or the dgrad operator:
// w'[ci][co][26-tap] viewed as a conv with Cin'=Co, Cout'=Ci -> packed[tap'][co][CiP].
__global__ void pack_conv_weight(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int cop,
                                 int flip_transpose, int taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int Kin = flip_transpose ? Co : Ci;       // input channels of the packed operator
    const int Kout = flip_transpose ? Ci : Co;
    if (i >= taps * Kin * cop) return;
    const int o = i % cop;
    const int k = (i / cop) % Kin;
    const int tap = i / (cop * Kin);
    float v = 0.f;
    if (o < Kout) {
        if (!flip_transpose) v = w[((size_t)o * Ci + k) * taps + tap];
        else v = w[((size_t)k * Ci + o) * taps + (taps - 1 - tap)];
    }
    packed[i] = v;
}

template <int CO_TILES, int STRIDE, int TD, int TH, int CIC, int KD = 3>
int launch_conv(const float* x, const float* wp, float* y, int B, int Ci, int Co, int D, int H, int W, hipStream_t st) {
    using Cfg = ConvCfg<CO_TILES, STRIDE, TD, TH, CIC, KD>;
    const int Do = (D - 1) / STRIDE + 1, Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const int tiles_d = (Do + TD - 1) / TD, tiles_h = (Ho + TH - 1) / TH, tiles_w = (Wo + TW - 1) / TW;
    const long long nblk = (long long)B * tiles_d * tiles_h * tiles_w;
    if (nblk > 0x7fffffffLL || (long long)D * H * W * 4 >= 0x80000000LL) return ECM_EUNSUP;
    auto kern = conv3d_k3_mfma<CO_TILES, STRIDE, TD, TH, CIC, KD>;
    {
        const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), Cfg::LDS_BYTES, st, x, wp, y, Ci, Co, D, H, W, Do, Ho, Wo,
                       tiles_d, tiles_h, tiles_w);
    return ECM_LAUNCH_RESULT();
}

}  // namespace

extern "C" long long ecm_conv3d_packed_floats(int Ci, int Co) {
    const long long cop = ((Co + 31) / 32) * 32;
    return 27LL * Ci * cop;
}

extern "C" int ecm_conv3d_pack_weight(const float* w, float* packed, int Co, int Ci, int flip_transpose, void* stream) {
    ECM_CHECK_ARG(w && packed && Co > 0 && Ci > 0);
    const int Kin = flip_transpose ? Co : Ci, Kout = flip_transpose ? Ci : Co;
    const int cop = ((Kout + 31) / 32) * 32;
    const int n = 27 * Kin * cop;
    hipLaunchKernelGGL(pack_conv_weight, dim3((n + 255) / 256), dim3(256), 0, ecm_stream(stream), w, packed, Co, Ci, cop,
                       flip_transpose, 27);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_conv3d_k3_fwd(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int D, int H,
                                 int W, int stride, void* stream) {
    ECM_CHECK_ARG(x && wpacked && y && B > 0 && D > 0 && H > 0 && W > 0);
    if (Ci % 4 != 0 || Co < 1 || Co > 64 || (stride != 1 && stride != 2)) return ECM_EUNSUP;
    hipStream_t st = ecm_stream(stream);
    const bool two = Co > 32;
    // Small volumes (the 1/8- and 1/16-resolution levels of the hourglass) would give only a few dozen workgroups with
    // the large tile; a 1 x 4 x 32 tile trades operand reuse for enough workgroups to cover the 256 CUs.
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long big_blocks = (long long)B * ((Do + 1) / 2) * ((Ho + 7) / 8) * ((Wo + TW - 1) / TW);
    const bool small = big_blocks < 384;
    if (stride == 1) {
        if (!two) return small ? launch_conv<1, 1, 1, 4, 4>(x, wpacked, y, B, Ci, Co, D, H, W, st)
                               : launch_conv<1, 1, 4, 8, 4>(x, wpacked, y, B, Ci, Co, D, H, W, st);
        return small ? launch_conv<2, 1, 1, 4, 4>(x, wpacked, y, B, Ci, Co, D, H, W, st)
                     : launch_conv<2, 1, 2, 8, 4>(x, wpacked, y, B, Ci, Co, D, H, W, st);
    }
    if (!two) return small ? launch_conv<1, 2, 1, 4, 2>(x, wpacked, y, B, Ci, Co, D, H, W, st)
                           : launch_conv<1, 2, 2, 8, 2>(x, wpacked, y, B, Ci, Co, D, H, W, st);
    return small ? launch_conv<2, 2, 1, 4, 2>(x, wpacked, y, B, Ci, Co, D, H, W, st)
                 : launch_conv<2, 2, 2, 8, 2>(x, wpacked, y, B, Ci, Co, D, H, W, st);
}
