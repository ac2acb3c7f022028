// ConvTranspose3d k=3, stride 2, pad 1, output_padding 1 (reference hourglass conv5/conv6, cmfsm.py:261-281)
// and, with the same arithmetic, the data gradient of the stride-2 Conv3d (hourglass conv1/conv3).
//   y[b,co,o] = sum_{ci,i,k : o = 2i-1+k} x[b,ci,i] * w[ci,co,k]          (per dimension)
// Outputs split into 8 parity classes (pd,ph,pw): an even output o=2m uses tap k=1 of input m; an odd
// output o=2m+1 uses tap k=2 of input m and tap k=0 of input m+1.  Each class is a dense small
// convolution over the INPUT grid, run as an implicit GEMM on the fp32 matrix cores exactly like
// conv3d.hip (A = weights [co][k], B = 32 consecutive input-x voxels of one row, D = [co][voxel]).
// One workgroup = one (1 x 4 x 32) tile of input voxels for ALL 8 classes (27 taps in total), so the MFMA count equals
// that of a stride-1 conv on the input grid and the halo tile is staged once.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TW = 32;

// One workgroup = one (1 x 4 x 32) tile of INPUT voxels and ALL 8 output parity classes: the input halo tile is staged
// once per channel chunk and every one of the 27 taps issues exactly one MFMA per k-step into the accumulator of the
// class it belongs to (so the MFMA count equals a stride-1 conv on the input grid and no class is staging-bound).
// Waves: 4 input rows x CO_TILES output-channel tiles (256 or 512 threads); 8 classes x 16 accumulator registers each.
// KD = 1: the 2-D form (a depth-1 volume without taps along depth): 9 taps, 4 parity classes -- the data gradient of the
// encoder's stride-2 3x3 Conv2d layers (cmfsm.py:141, layer2's first block).
template <int CO_TILES, int CIC, int KD = 3>
struct DeconvCfg {
    static constexpr int TD = 1, TH = 4, NTAPS = 9 * KD;
    static constexpr int ID = KD == 3 ? TD + 1 : 1, IH = TH + 1, IW = TW + 1;
    static constexpr int RS = IW;
    static constexpr int COP = CO_TILES * 32;
    static constexpr int THREADS = 256 * CO_TILES;
    static constexpr int XS_FLOATS = CIC * ID * IH * RS;
    static constexpr int WS_FLOATS = NTAPS * CIC * COP;
    static constexpr int LDS_BYTES = (XS_FLOATS + 2 * WS_FLOATS) * 4;
};

template <int CO_TILES, int CIC, int KD = 3>
__global__ __launch_bounds__(256 * CO_TILES) void deconv3d_k3s2_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                                     float* __restrict__ y, int Ci, int Co, int D, int H,
                                                                     int W, int Do, int Ho, int Wo, int tiles_d, int tiles_h,
                                                                     int tiles_w) {
    using Cfg = DeconvCfg<CO_TILES, CIC, KD>;
    constexpr int TD = Cfg::TD, TH = Cfg::TH, ID = Cfg::ID, IH = Cfg::IH, IW = Cfg::IW, RS = Cfg::RS, COP = Cfg::COP,
                  NTHR = Cfg::THREADS, NTAPS = Cfg::NTAPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                        // [CIC][ID][IH][RS]
    float* Ws = smem + Cfg::XS_FLOATS;       // 2 x [27][CIC][COP]

    int bid = ecm_xcd_tile(blockIdx.x, gridDim.x);       // one contiguous run of tiles per XCD (its own L2), see common.h
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int td = bid % tiles_d;
    const int b = bid / tiles_d;
    const int md0 = td * TD, mh0 = th * TH, mw0 = tw * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int row = wave & 3, ct = wave >> 2;                    // this wave: input row (hy) and output-channel tile
    const int xbase = (half * ID * IH + row) * RS + l31;
    const int wbase = half * COP + ct * 32 + l31;

    f32x16 acc[8];                                               // one per parity class (pd,ph,pw)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

    const size_t HWi = (size_t)H * W, DHWi = (size_t)D * HWi;
    const float* xb = x + (size_t)b * Ci * DHWi;

    constexpr int NPOS = ID * IH * IW;
    constexpr int PP = (NPOS + NTHR - 1) / NTHR;
    constexpr int NWQ = (NTAPS * CIC * COP / 4 + NTHR - 1) / NTHR;
    float xr[CIC * PP];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned posoff[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int p = tid + j * NTHR;
        int t = p;
        const int xx = t % IW; t /= IW;
        const int hy = t % IH;
        const int dz = t / IH;
        const int gz = md0 + dz, gy = mh0 + hy, gx = mw0 + xx;
        const bool ok = p < NPOS && gz < D && gy < H && gx < W;
        posoff[j] = ok ? (unsigned)(gz * (int)HWi + gy * W + gx) * 4u : 0x80000000u;
    }
    const unsigned plane_bytes = (unsigned)DHWi * 4u;
    auto prefetch = [&](int c0, float* wdst) {
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            const int e = tid + i * NTHR;
            if (e < NTAPS * CIC * COP / 4) {
                const int tap = e / (CIC * COP / 4), r = e - tap * (CIC * COP / 4);
                const float* src = wp + ((size_t)tap * Ci + c0) * COP + (size_t)r * 4;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(wdst + (wave_u * 64 + i * NTHR) * 4), 16, 0, 0);
            }
        }
    };
    // the halo loads of a chunk one at a time (i = cc * PP + j), issued a few per tap inside the MFMA loop (see conv3d.hip)
    constexpr int NX = CIC * PP;
    auto prefetch_x = [&](int c0, int i) {
        const int cc = i / PP, j = i % PP;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb + (size_t)(c0 + cc) * DHWi), 0, plane_bytes,
                                                            0x00020000);
        xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, posoff[j], 0, 0));
    };
    prefetch(0, Ws);
#pragma unroll
    for (int i = 0; i < NX; ++i) prefetch_x(0, i);
    int buf = 0;
    for (int c0 = 0; c0 < Ci; c0 += CIC, buf ^= 1) {
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CIC; ++cc)
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const int p = tid + j * NTHR;
                if (p < NPOS) Xs[cc * NPOS + p] = xr[cc * PP + j];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* Wc = Ws + buf * Cfg::WS_FLOATS;
        const bool more = c0 + CIC < Ci;
        if (more) prefetch(c0 + CIC, Ws + (buf ^ 1) * Cfg::WS_FLOATS);
        constexpr int LPT = (NX + NTAPS - 1) / NTAPS;
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            if (more) {
#pragma unroll
                for (int q = 0; q < LPT; ++q)
                    if (tap * LPT + q < NX) prefetch_x(c0 + CIC, tap * LPT + q);
            }
            // tap k of an output of parity p reads input m + (k == 0 ? 1 : 0); k == 1 <-> even output, k in {0,2} <-> odd
            const int kd = KD == 3 ? tap / 9 : 1, kh = (tap / 3) % 3, kw = tap % 3;
            const int pd = kd != 1, ph = kh != 1, pw = kw != 1;
            const int sd = kd == 0, sh = kh == 0, sw = kw == 0;
            const int cls = (pd * 2 + ph) * 2 + pw;
#pragma unroll
            for (int kk = 0; kk < CIC / 2; ++kk) {
                const float a = Wc[wbase + (tap * CIC + kk * 2) * COP];
                const float bv = Xs[xbase + ((kk * 2 * ID + sd) * IH + sh) * RS + sw];
                acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[cls], 0, 0, 0);
            }
        }
    }

    const size_t HWo = (size_t)Ho * Wo, DHWo = (size_t)Do * HWo;
    float* yb = y + (size_t)b * Co * DHWo;
    const int md = md0, mh = mh0 + row, mw = mw0 + l31;
    if (md < D && mh < H && mw < W) {
        // The two width-parity classes of a (depth, height) parity pair are the neighbouring outputs ow = 2 mw, 2 mw + 1: one
        // 8-byte store instead of two stride-2 4-byte ones (32 lanes then write 256 contiguous bytes per instruction) whenever
        // the pair is 8-byte aligned, i.e. for even output widths -- every layer of the registered models.
        const bool pair_ok = (Wo & 1) == 0 && (HWo & 1) == 0 && (DHWo & 1) == 0 && (reinterpret_cast<size_t>(yb) & 7) == 0;
#pragma unroll
        for (int cls = 0; cls < 8; cls += 2) {
            const int od = 2 * md + (cls >> 2), oh = 2 * mh + ((cls >> 1) & 1), ow = 2 * mw;
            if (od >= Do || oh >= Ho || ow >= Wo) continue;
            float* yp = yb + (size_t)od * HWo + (size_t)oh * Wo + ow;
            if (pair_ok) {                                     // (ow + 1 < Wo follows from Wo even)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int co = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    // (plain stores: with the non-temporal hint the quarter-resolution layer, whose 53 MB output the next kernel
                    //  finds in the caches, ran 11 % slower and the large one the same -- round 4 A/B)
                    if (co < Co) *reinterpret_cast<float2*>(yp + (size_t)co * DHWo) = make_float2(acc[cls][i], acc[cls + 1][i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int co = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    if (co < Co) {
                        yp[(size_t)co * DHWo] = acc[cls][i];
                        if (ow + 1 < Wo) yp[(size_t)co * DHWo + 1] = acc[cls + 1][i];
                    }
                }
            }
        }
    }
}

// [A][Bc][taps] (ConvTranspose3d [Ci,Co,27], or a Conv3d / Conv2d weight [Co_f,Ci_f,27 | 9] whose dgrad is wanted)
// -> [taps][A][BcP]
__global__ void pack_deconv_weight(const float* __restrict__ w, float* __restrict__ packed, int A, int Bc, int bcp, int taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= taps * A * bcp) return;
    const int o = i % bcp;
    const int k = (i / bcp) % A;
    const int tap = i / (bcp * A);
    packed[i] = o < Bc ? w[((size_t)k * Bc + o) * taps + tap] : 0.f;
}

template <int CO_TILES, int CIC, int KD = 3>
int launch_deconv(const float* x, const float* wp, float* y, int B, int Ci, int Co, int D, int H, int W, int Do, int Ho,
                  int Wo, hipStream_t st) {
    using Cfg = DeconvCfg<CO_TILES, CIC, KD>;
    const int tiles_d = (D + Cfg::TD - 1) / Cfg::TD, tiles_h = (H + Cfg::TH - 1) / Cfg::TH, tiles_w = (W + TW - 1) / TW;
    const long long nblk = (long long)B * tiles_d * tiles_h * tiles_w;
    if (nblk > 0x7fffffffLL || (long long)D * H * W * 4 >= 0x80000000LL) return ECM_EUNSUP;
    auto kern = deconv3d_k3s2_mfma<CO_TILES, CIC, KD>;
    {
        const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, x, wp, y, Ci, Co, D, H, W, Do, Ho,
                       Wo, tiles_d, tiles_h, tiles_w);
    return ECM_LAUNCH_RESULT();
}

}  // namespace

extern "C" int ecm_deconv3d_pack_weight(const float* w, float* packed, int Ci, int Co, void* stream) {
    ECM_CHECK_ARG(w && packed && Ci > 0 && Co > 0);
    const int cop = ((Co + 31) / 32) * 32;
    const int n = 27 * Ci * cop;
    hipLaunchKernelGGL(pack_deconv_weight, dim3((n + 255) / 256), dim3(256), 0, ecm_stream(stream), w, packed, Ci, Co,
                       cop, 27);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_deconv3d_k3s2_fwd(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int D, int H,
                                     int W, int Do, int Ho, int Wo, void* stream) {
    ECM_CHECK_ARG(x && wpacked && y && B > 0 && D > 0 && H > 0 && W > 0);
    // output extent per dim is 2n (output_padding 1) or 2n-1 (dgrad of a stride-2 conv on an odd extent)
    if (Ci % 4 != 0 || Co < 1 || Co > 64) return ECM_EUNSUP;
    if (Do > 2 * D || Do < 2 * D - 1 || Ho > 2 * H || Ho < 2 * H - 1 || Wo > 2 * W || Wo < 2 * W - 1) return ECM_EUNSUP;
    hipStream_t st = ecm_stream(stream);
    if (Co > 32) return launch_deconv<2, 4>(x, wpacked, y, B, Ci, Co, D, H, W, Do, Ho, Wo, st);
    return launch_deconv<1, 4>(x, wpacked, y, B, Ci, Co, D, H, W, Do, Ho, Wo, st);
}

// ---- 2-D: the data gradient of a stride-2 3x3 Conv2d (pad 1) = ConvTranspose2d(k 3, s 2, p 1) on a depth-1 volume ------
extern "C" int ecm_deconv2d_pack_weight(const float* w, float* packed, int Ci, int Co, void* stream) {
    ECM_CHECK_ARG(w && packed && Ci > 0 && Co > 0);
    const int cop = ((Co + 31) / 32) * 32;
    const int n = 9 * Ci * cop;
    hipLaunchKernelGGL(pack_deconv_weight, dim3((n + 255) / 256), dim3(256), 0, ecm_stream(stream), w, packed, Ci, Co,
                       cop, 9);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_deconv2d_k3s2_fwd(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int H, int W,
                                     int Ho, int Wo, void* stream) {
    ECM_CHECK_ARG(x && wpacked && y && B > 0 && H > 0 && W > 0);
    if (Ci % 4 != 0 || Co < 1 || Co > 64) return ECM_EUNSUP;
    if (Ho > 2 * H || Ho < 2 * H - 1 || Wo > 2 * W || Wo < 2 * W - 1) return ECM_EUNSUP;
    hipStream_t st = ecm_stream(stream);
    if (Co > 32) return launch_deconv<2, 4, 1>(x, wpacked, y, B, Ci, Co, 1, H, W, 1, Ho, Wo, st);
    return launch_deconv<1, 4, 1>(x, wpacked, y, B, Ci, Co, 1, H, W, 1, Ho, Wo, st);
}
