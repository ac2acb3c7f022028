// 2-D convolutions of the path as implicit GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32):
//   * the encoder's Conv2d layers (feature_extraction, cmfsm.py:126-236; convbn 36-46): 3x3 with stride 1|2 and dilation
//     1|2|4, the 3-channel stem, the 64/128/320-channel stages, and the 1x1 projections (downsample, SPP branches, lastconv);
//   * the class-indexed convolutions of the collapsed cost volume + dres0.0 (cmfsm.py:667-684; see costvol_conv.hip):
//     P = 3x3, 32 -> 15*32 on the reference features and Q = sheared 3x5, 32 -> 6*32 on the left-padded target features;
//   * every stride-1 data gradient (the same kernel on the flipped / transposed weights).
// Same GEMM view and staging as conv3d.hip (its KD = 1 instantiations remain the 32/64-channel fast path):
//   D[co][pixel] += sum_k A[co][k] B[k][pixel],  k = (tap, ci);  A = weights from LDS (global->LDS DMA, double buffered),
//   B = 32 consecutive x of one row of the staged halo tile (NCHW as it stands is the operand layout), register-pipelined
//   through buffer descriptors whose range check supplies the zero padding.
// New here: output channels beyond one workgroup's COT*32 go to blockIdx.y ("co groups", packed weights grouped to match);
// kernel shape KH x KW, stride, dilation and the (possibly asymmetric) padding are template / run-time parameters; input
// channel counts that are not a multiple of the chunk (the 3-channel stem) read zeros for the missing planes.
#pragma once
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TW = 32;

template <int COT, int NT, int CIC, int KH, int KW, int STRIDE, int DIL>
struct C2Cfg {
    static constexpr int TH = 4 * NT;                                   // 4 waves x NT rows
    static constexpr int NTAPS = KH * KW;
    static constexpr int IH = (TH - 1) * STRIDE + (KH - 1) * DIL + 1;
    static constexpr int IW = (TW - 1) * STRIDE + (KW - 1) * DIL + 1;
    static constexpr int RS = IW;
    static constexpr int COP = COT * 32;
    static constexpr int XS_FLOATS = CIC * IH * RS;
    static constexpr int WS_FLOATS = NTAPS * CIC * COP;
    static constexpr int LDS_BYTES = (XS_FLOATS + 2 * WS_FLOATS) * 4;
    static_assert(CIC % 2 == 0, "k-step is 2 channels");
    static_assert((NTAPS * CIC * COP) % 4 == 0, "weight slice moves as float4");
};

// packed weights: [co group][tap][CiP][COP]  (CiP = Ci rounded up to CIC: zero rows; COP = COT*32: zero columns past Co)
template <int COT, int NT, int CIC, int KH, int KW, int STRIDE, int DIL>
__global__ __launch_bounds__(256, 2) void conv2d_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                   float* __restrict__ y, int Ci, int CiP, int Co, int H, int W, int Ho,
                                                   int Wo, int pad_top, int pad_left, int tiles_h, int tiles_w) {
    using Cfg = C2Cfg<COT, NT, CIC, KH, KW, STRIDE, DIL>;
    constexpr int IH = Cfg::IH, IW = Cfg::IW, RS = Cfg::RS, COP = Cfg::COP, NTAPS = Cfg::NTAPS, TH = Cfg::TH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [CIC][IH][RS]
    float* Ws = smem + Cfg::XS_FLOATS;      // 2 x [NTAPS][CIC][COP]

    int bid = ecm_xcd_tile(blockIdx.x, gridDim.x);      // one contiguous run of tiles per XCD: row neighbours share halo rows
    const int th = bid % tiles_h; bid /= tiles_h;
    const int tw = bid % tiles_w;
    const int b = bid / tiles_w;
    const int grp = blockIdx.y;
    const int oh0 = th * TH, ow0 = tw * TW;
    const int ih0 = oh0 * STRIDE - pad_top, iw0 = ow0 * STRIDE - pad_left;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int hy0 = wave * NT;
    const int xbase = (half * IH + hy0 * STRIDE) * RS + l31 * STRIDE;
    const int wbase = half * COP + l31;

    f32x16 acc[NT][COT];
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][ct][i] = 0.f;

    const size_t HWi = (size_t)H * W;
    const float* xb = x + (size_t)b * Ci * HWi;
    const float* wg = wp + (size_t)grp * NTAPS * CiP * COP;

    constexpr int NPOS = IH * IW;
    constexpr int PP = (NPOS + 255) / 256;
    constexpr int NWQ = (NTAPS * CIC * COP / 4 + 255) / 256;
    float xr[CIC * PP];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned posoff[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int p = tid + j * 256;
        const int xx = p % IW, hy = p / IW;
        const int gy = ih0 + hy, gx = iw0 + xx;
        const bool ok = p < NPOS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        posoff[j] = ok ? (unsigned)(gy * W + gx) * 4u : 0x80000000u;
    }
    const unsigned plane_bytes = (unsigned)HWi * 4u;
    auto prefetch = [&](int c0, float* wdst) {
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            const int e = tid + i * 256;
            if (e < NTAPS * CIC * COP / 4) {
                const int tap = e / (CIC * COP / 4), r = e - tap * (CIC * COP / 4);
                const float* src = wg + ((size_t)tap * CiP + c0) * COP + (size_t)r * 4;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(wdst + (wave_u * 64 + i * 256) * 4), 16, 0, 0);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CIC; ++cc) {
            // planes past Ci (channel padding of the stem) get an empty descriptor: every load returns 0
            const bool live = c0 + cc < Ci;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb + (size_t)(live ? c0 + cc : 0) * HWi), 0,
                                                                live ? plane_bytes : 0u, 0x00020000);
#pragma unroll
            for (int j = 0; j < PP; ++j)
                xr[cc * PP + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, posoff[j], 0, 0));
        }
    };
    prefetch(0, Ws);
    int buf = 0;
    for (int c0 = 0; c0 < CiP; c0 += CIC, buf ^= 1) {
        __syncthreads();                                   // previous chunk's LDS reads are done
#pragma unroll
        for (int cc = 0; cc < CIC; ++cc)
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const int p = tid + j * 256;
                if (p < NPOS) Xs[cc * NPOS + p] = xr[cc * PP + j];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight DMA for this chunk has landed
        __syncthreads();
        const float* Wc = Ws + buf * Cfg::WS_FLOATS;
        if (c0 + CIC < CiP) prefetch(c0 + CIC, Ws + (buf ^ 1) * Cfg::WS_FLOATS);
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int kh = tap / KW, kw = tap % KW;
#pragma unroll
            for (int kk = 0; kk < CIC / 2; ++kk) {
                float a[COT];
#pragma unroll
                for (int ct = 0; ct < COT; ++ct) a[ct] = Wc[wbase + (tap * CIC + kk * 2) * COP + ct * 32];
#pragma unroll
                for (int r = 0; r < NT; ++r) {
                    const float bv = Xs[xbase + ((kk * 2) * IH + r * STRIDE + kh * DIL) * RS + kw * DIL];
#pragma unroll
                    for (int ct = 0; ct < COT; ++ct)
                        acc[r][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct], bv, acc[r][ct], 0, 0, 0);
                }
            }
        }
    }

    const size_t HWo = (size_t)Ho * Wo;
    float* yb = y + (size_t)b * Co * HWo;
    const int ow = ow0 + l31;
#pragma unroll
    for (int r = 0; r < NT; ++r) {
        const int oh = oh0 + hy0 + r;
        if (oh >= Ho || ow >= Wo) continue;
        float* yp = yb + (size_t)oh * Wo + ow;
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = grp * COP + ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                if (co < Co) yp[(size_t)co * HWo] = acc[r][ct][i];
            }
    }
}

// Conv2d weight [Co,Ci,KH,KW] -> [group][tap][CiP][COP] (zero padded), or with flip_transpose the data-gradient operator
// of a stride-1 conv: w'[ci][co][flipped tap], i.e. a conv with Cin' = Co and Cout' = Ci.
__global__ void pack_conv2d_weight(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int taps,
                                   int cop, int kinp, int flip_transpose, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int Kin = flip_transpose ? Co : Ci, Kout = flip_transpose ? Ci : Co;
    const int o = (int)(i % cop);
    long long r = i / cop;
    const int k = (int)(r % kinp); r /= kinp;
    const int tap = (int)(r % taps);
    const int grp = (int)(r / taps);
    const int oc = grp * cop + o;
    float v = 0.f;
    if (oc < Kout && k < Kin) {
        if (!flip_transpose) v = w[((size_t)oc * Ci + k) * taps + tap];
        else v = w[((size_t)k * Ci + oc) * taps + (taps - 1 - tap)];
    }
    packed[i] = v;
}

// Tiling policy shared by the packer and the launcher: output-channel tiles per workgroup and input-channel chunk.
struct C2Plan { int cot, cic, groups, cip; };
inline C2Plan c2_plan(int Ci, int Co, int kh, int kw) {
    C2Plan p;
    p.cot = Co <= 32 ? 1 : Co <= 64 ? 2 : (kh * kw != 1 && Co % 96 == 0 && Co % 128 != 0) ? 3 : 4;
    p.cic = (kh * kw == 1) ? 8 : 4;
    p.groups = (Co + p.cot * 32 - 1) / (p.cot * 32);
    p.cip = (Ci + p.cic - 1) / p.cic * p.cic;
    return p;
}

template <int COT, int NT, int CIC, int KH, int KW, int STRIDE, int DIL>
int launch_c2(const float* x, const float* wp, float* y, int B, int Ci, int CiP, int Co, int H, int W, int Ho, int Wo,
              int pad_top, int pad_left, int groups, hipStream_t st) {
    using Cfg = C2Cfg<COT, NT, CIC, KH, KW, STRIDE, DIL>;
    const int tiles_h = (Ho + Cfg::TH - 1) / Cfg::TH, tiles_w = (Wo + TW - 1) / TW;
    const long long nblk = (long long)B * tiles_h * tiles_w;
    if (nblk > 0x7fffffffLL || groups > 65535 || (long long)H * W * 4 >= 0x80000000LL) return ECM_EUNSUP;
    auto kern = conv2d_mfma<COT, NT, CIC, KH, KW, STRIDE, DIL>;
    const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)groups), dim3(256), Cfg::LDS_BYTES, st, x, wp, y, Ci, CiP, Co, H, W,
                       Ho, Wo, pad_top, pad_left, tiles_h, tiles_w);
    return ECM_LAUNCH_RESULT();
}

// rows per wave: the largest tile that still gives the chip >= ~3 rounds of workgroups
inline int c2_min_blocks() {
    static const int v = [] { const char* e = getenv("ECM_C2_MIN_BLOCKS"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 1536; }();
    return v;
}
inline int c2_nt(long long cols_x_groups, int Ho, int max_nt) {
    for (int nt = max_nt; nt > 1; nt >>= 1)
        if (cols_x_groups * ((Ho + 4 * nt - 1) / (4 * nt)) >= c2_min_blocks()) return nt;
    return 1;
}

// COTS: bit mask of the output-channel tilings (1 << cot) a (shape, stride, dilation) case is instantiated for -- only what
// the registered architectures' layers and their data gradients use; anything else is ECM_EUNSUP, not a silent slow path.
template <int KH, int KW, int STRIDE, int DIL, int COTS>
int dispatch_c2(const float* x, const float* wp, float* y, int B, int Ci, int Co, int H, int W, int Ho, int Wo, int pad_top,
                int pad_left, hipStream_t st) {
    const C2Plan p = c2_plan(Ci, Co, KH, KW);
    const long long cg = (long long)B * ((Wo + TW - 1) / TW) * p.groups;
#define C2_GO(COT, NT, CIC) return launch_c2<COT, NT, CIC, KH, KW, STRIDE, DIL>(x, wp, y, B, Ci, p.cip, Co, H, W, Ho, Wo, pad_top, pad_left, p.groups, st)
    if constexpr (KH * KW == 1) {
        if constexpr ((COTS & 2) != 0) if (p.cot == 1) { C2_GO(1, 4, 8); }
        if constexpr ((COTS & 4) != 0) if (p.cot == 2) { C2_GO(2, 2, 8); }
        if constexpr ((COTS & 16) != 0) if (p.cot == 4) { C2_GO(4, 2, 8); }
    } else {
        // (8 rows per wave at stride 2 would stage a 65 x 65 halo per channel: 17 prefetch registers per channel next to 128
        //  accumulators spilled 237 registers into private memory -- the only such kernel of the 2-D family; 4 rows fit)
        if constexpr ((COTS & 2) != 0) if (p.cot == 1) { if constexpr (STRIDE == 1) { if (c2_nt(cg, Ho, 8) >= 8) { C2_GO(1, 8, 4); } } C2_GO(1, 4, 4); }
        if constexpr ((COTS & 4) != 0) if (p.cot == 2) { if (c2_nt(cg, Ho, 4) >= 4) { C2_GO(2, 4, 4); } C2_GO(2, 2, 4); }
        if constexpr ((COTS & 8) != 0) if (p.cot == 3) { C2_GO(3, 2, 4); }
        if constexpr ((COTS & 16) != 0) if (p.cot == 4) { if (c2_nt(cg, Ho, 2) >= 2) { C2_GO(4, 2, 4); } C2_GO(4, 1, 4); }
    }
#undef C2_GO
    return ECM_EUNSUP;
}

}  // namespace


// one translation unit per (shape, stride, dilation) case keeps the build parallel (conv2d_cases_*.hip)
#define ECM_C2_ARGS const float* x, const float* wp, float* y, int B, int Ci, int Co, int H, int W, int Ho, int Wo, int pad_top, int pad_left, hipStream_t st
int ecm_c2_k33_s1_d1(ECM_C2_ARGS);
int ecm_c2_k33_s1_d2(ECM_C2_ARGS);
int ecm_c2_k33_s1_d4(ECM_C2_ARGS);
int ecm_c2_k33_s2_d1(ECM_C2_ARGS);
int ecm_c2_k35_s1_d1(ECM_C2_ARGS);
int ecm_c2_k11_s1(ECM_C2_ARGS);
int ecm_c2_k11_s2(ECM_C2_ARGS);
