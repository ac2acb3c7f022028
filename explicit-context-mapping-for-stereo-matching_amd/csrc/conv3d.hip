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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TW = 32;

template <int CO_TILES, int STRIDE, int TD, int TH, int CIC>
struct ConvCfg {
    static constexpr int ID = (TD - 1) * STRIDE + 3;
    static constexpr int IH = (TH - 1) * STRIDE + 3;
    static constexpr int IW = (TW - 1) * STRIDE + 3;
    static constexpr int RS = IW;                       // LDS row stride (floats)
    static constexpr int ROWS = TD * TH;
    static constexpr int NT = ROWS / 4;                 // rows per wave
    static constexpr int COP = CO_TILES * 32;
    static constexpr int XS_FLOATS = CIC * ID * IH * RS;
    static constexpr int WS_FLOATS = 27 * CIC * COP;
    static constexpr int LDS_BYTES = (XS_FLOATS + WS_FLOATS) * 4;
    static_assert(ROWS % 4 == 0, "rows must split over 4 waves");
    static_assert((NT <= TH && TH % NT == 0) || (NT % TH == 0), "wave rows must tile (dz,hy) statically");
    static_assert(CIC % 2 == 0, "k-step is 2 channels");
};

template <int CO_TILES, int STRIDE, int TD, int TH, int CIC>
__global__ __launch_bounds__(256) void conv3d_k3_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                      float* __restrict__ y, int Ci, int Co, int D, int H, int W,
                                                      int Do, int Ho, int Wo, int tiles_d, int tiles_h, int tiles_w) {
    using Cfg = ConvCfg<CO_TILES, STRIDE, TD, TH, CIC>;
    constexpr int ID = Cfg::ID, IH = Cfg::IH, IW = Cfg::IW, RS = Cfg::RS, NT = Cfg::NT, COP = Cfg::COP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [CIC][ID][IH][RS]
    float* Ws = smem + Cfg::XS_FLOATS;      // [27][CIC][COP]

    // tile decode (x fastest so neighbouring workgroups share halo rows in L2)
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int td = bid % tiles_d;
    const int b = bid / tiles_d;
    const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TW;
    const int id0 = od0 * STRIDE - 1, ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
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

    for (int c0 = 0; c0 < Ci; c0 += CIC) {
        __syncthreads();
        // ---- stage the halo tile (zero padded) -------------------------------------------------
        for (int e = tid; e < CIC * ID * IH * IW; e += 256) {
            int t = e;
            const int xx = t % IW; t /= IW;
            const int hy = t % IH; t /= IH;
            const int dz = t % ID;
            const int cc = t / ID;
            const int gz = id0 + dz, gy = ih0 + hy, gx = iw0 + xx;
            float v = 0.f;
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = xb[(size_t)(c0 + cc) * DHWi + (size_t)gz * HWi + (size_t)gy * W + gx];
            Xs[((cc * ID + dz) * IH + hy) * RS + xx] = v;
        }
        // ---- stage the weight slice [27][CIC][COP] (packed global layout [27][Ci][COP]) --------
        for (int e = tid; e < 27 * CIC * COP / 4; e += 256) {
            const int tap = e / (CIC * COP / 4), r = e - tap * (CIC * COP / 4);
            reinterpret_cast<float4*>(Ws)[e] =
                reinterpret_cast<const float4*>(wp + ((size_t)tap * Ci + c0) * COP)[r];
        }
        __syncthreads();
        // ---- 27 * CIC/2 k-steps ---------------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            constexpr int dummy = 0; (void)dummy;
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
            for (int kk = 0; kk < CIC / 2; ++kk) {
                float a[CO_TILES];
#pragma unroll
                for (int ct = 0; ct < CO_TILES; ++ct) a[ct] = Ws[wbase + (tap * CIC + kk * 2) * COP + ct * 32];
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
                if (co < Co) yp[(size_t)co * DHWo] = acc[r][ct][i];
            }
    }
}

// Conv3d weight [Co,Ci,27] -> [27][Ci][COP] (zero padded co), or the dgrad operator:
// w'[ci][co][26-tap] viewed as a conv with Cin'=Co, Cout'=Ci -> packed[tap'][co][CiP].
__global__ void pack_conv_weight(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int cop,
                                 int flip_transpose) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int Kin = flip_transpose ? Co : Ci;       // input channels of the packed operator
    const int Kout = flip_transpose ? Ci : Co;
    if (i >= 27 * Kin * cop) return;
    const int o = i % cop;
    const int k = (i / cop) % Kin;
    const int tap = i / (cop * Kin);
    float v = 0.f;
    if (o < Kout) {
        if (!flip_transpose) v = w[((size_t)o * Ci + k) * 27 + tap];
        else v = w[((size_t)k * Ci + o) * 27 + (26 - tap)];
    }
    packed[i] = v;
}

template <int CO_TILES, int STRIDE, int TD, int TH, int CIC>
int launch_conv(const float* x, const float* wp, float* y, int B, int Ci, int Co, int D, int H, int W, hipStream_t st) {
    using Cfg = ConvCfg<CO_TILES, STRIDE, TD, TH, CIC>;
    const int Do = (D - 1) / STRIDE + 1, Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const int tiles_d = (Do + TD - 1) / TD, tiles_h = (Ho + TH - 1) / TH, tiles_w = (Wo + TW - 1) / TW;
    const long long nblk = (long long)B * tiles_d * tiles_h * tiles_w;
    if (nblk > 0x7fffffffLL) return ECM_EUNSUP;
    auto kern = conv3d_k3_mfma<CO_TILES, STRIDE, TD, TH, CIC>;
    static bool attr_set = false;       // benign race: same value every time
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
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
                       flip_transpose);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_conv3d_k3_fwd(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int D, int H,
                                 int W, int stride, void* stream) {
    ECM_CHECK_ARG(x && wpacked && y && B > 0 && D > 0 && H > 0 && W > 0);
    if (Ci % 4 != 0 || Co < 1 || Co > 64 || (stride != 1 && stride != 2)) return ECM_EUNSUP;
    hipStream_t st = ecm_stream(stream);
    const bool two = Co > 32;
    if (stride == 1) {
        if (!two) return launch_conv<1, 1, 4, 8, 4>(x, wpacked, y, B, Ci, Co, D, H, W, st);
        return launch_conv<2, 1, 2, 8, 4>(x, wpacked, y, B, Ci, Co, D, H, W, st);
    }
    if (!two) return launch_conv<1, 2, 2, 8, 2>(x, wpacked, y, B, Ci, Co, D, H, W, st);
    return launch_conv<2, 2, 2, 8, 2>(x, wpacked, y, B, Ci, Co, D, H, W, st);
}
