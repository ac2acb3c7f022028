// ConvTranspose3d k=3, stride 2, pad 1, output_padding 1 (reference hourglass conv5/conv6, cmfsm.py:261-281)
// and, with the same arithmetic, the data gradient of the stride-2 Conv3d (hourglass conv1/conv3).
//   y[b,co,o] = sum_{ci,i,k : o = 2i-1+k} x[b,ci,i] * w[ci,co,k]          (per dimension)
// Outputs split into 8 parity classes (pd,ph,pw): an even output o=2m uses tap k=1 of input m; an odd
// output o=2m+1 uses tap k=2 of input m and tap k=0 of input m+1.  Each class is a dense small
// convolution over the INPUT grid, run as an implicit GEMM on the fp32 matrix cores exactly like
// conv3d.hip (A = weights [co][k], B = 32 consecutive input-x voxels of one row, D = [co][voxel]).
// One workgroup = one class x one (TD x TH x 32) tile of m; 27 taps in total over the 8 classes, so
// the MFMA count equals that of a stride-1 conv on the input grid.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TW = 32;

template <int CO_TILES, int TD, int TH, int CIC>
struct DeconvCfg {
    static constexpr int ID = TD + 1, IH = TH + 1, IW = TW + 1;
    static constexpr int RS = IW;
    static constexpr int NT = TD * TH / 4;
    static constexpr int COP = CO_TILES * 32;
    static constexpr int XS_FLOATS = CIC * ID * IH * RS;
    static constexpr int WS_FLOATS = 8 * CIC * COP;                 // one parity class uses at most 8 taps
    static constexpr int LDS_BYTES = (XS_FLOATS + 2 * WS_FLOATS) * 4;
    static_assert((TD * TH) % 4 == 0 && NT <= TH && TH % NT == 0, "tile/wave split");
};

template <int CO_TILES, int TD, int TH, int CIC>
__global__ __launch_bounds__(256, 2) void deconv3d_k3s2_mfma(const float* __restrict__ x, const float* __restrict__ wp,
                                                          float* __restrict__ y, int Ci, int Co, int D, int H, int W,
                                                          int Do, int Ho, int Wo, int tiles_d, int tiles_h,
                                                          int tiles_w) {
    using Cfg = DeconvCfg<CO_TILES, TD, TH, CIC>;
    constexpr int ID = Cfg::ID, IH = Cfg::IH, IW = Cfg::IW, RS = Cfg::RS, NT = Cfg::NT, COP = Cfg::COP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ws = smem + Cfg::XS_FLOATS;

    int bid = blockIdx.x;
    const int cls = bid & 7; bid >>= 3;          // classes of one tile are adjacent workgroups (shared input in L2)
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int td = bid % tiles_d;
    const int b = bid / tiles_d;
    const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    const int md0 = td * TD, mh0 = th * TH, mw0 = tw * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int row0 = wave * NT;
    const int dz0 = row0 / TH, hy0 = row0 % TH;
    const int xbase = ((half * ID + dz0) * IH + hy0) * RS + l31;
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
    const int nd = 1 + pd, nh = 1 + ph, nw = 1 + pw;     // taps per dimension for this class

    // Staging as in conv3d.hip: halo tile through registers with buffer-descriptor loads (hardware zero fill for
    // positions beyond the input), this class's taps only ( <= 8 of the 27) by LDS-DMA into a double buffer.
    constexpr int NPOS = ID * IH * IW;
    constexpr int PP = (NPOS + 255) / 256;
    constexpr int NX = CIC * PP;
    constexpr int WSLICE = CIC * COP;                              // floats per tap per chunk
    constexpr int NWQ = (8 * WSLICE / 4 + 255) / 256;
    float xr[NX];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ntaps = nd * nh * nw;
    unsigned posoff[PP];
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const int p = tid + j * 256;
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
            const int e = tid + i * 256;
            const int lt = e / (WSLICE / 4), r = e - lt * (WSLICE / 4);         // local tap index (a_d, a_h, a_w)
            if (lt < ntaps) {
                const int a_w = lt % nw, a_h = (lt / nw) % nh, a_d = lt / (nw * nh);
                const int kd = pd ? (a_d ? 0 : 2) : 1, kh = ph ? (a_h ? 0 : 2) : 1, kw = pw ? (a_w ? 0 : 2) : 1;
                const int tap = (kd * 3 + kh) * 3 + kw;
                const float* src = wp + ((size_t)tap * Ci + c0) * COP + (size_t)r * 4;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(wdst + (wave_u * 64 + i * 256) * 4), 16, 0, 0);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CIC; ++cc) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb + (size_t)(c0 + cc) * DHWi), 0,
                                                                plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < PP; ++j)
                xr[cc * PP + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, posoff[j], 0, 0));
        }
    };
    prefetch(0, Ws);
    int buf = 0;
    for (int c0 = 0; c0 < Ci; c0 += CIC, buf ^= 1) {
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CIC; ++cc)
#pragma unroll
            for (int j = 0; j < PP; ++j) {
                const int p = tid + j * 256;
                if (p < NPOS) Xs[cc * NPOS + p] = xr[cc * PP + j];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* Wc = Ws + buf * (8 * WSLICE);
        if (c0 + CIC < Ci) prefetch(c0 + CIC, Ws + (buf ^ 1) * (8 * WSLICE));
        int lt = 0;
        for (int a_d = 0; a_d < nd; ++a_d) {
            const int sd = a_d;                                   // input offset (+0 / +1) of this tap
            for (int a_h = 0; a_h < nh; ++a_h) {
                const int sh = a_h;
                for (int a_w = 0; a_w < nw; ++a_w, ++lt) {
                    const int sw = a_w;
                    const float* wt = Wc + wbase + lt * WSLICE;
                    const float* xt = Xs + xbase + (sd * IH + sh) * RS + sw;
#pragma unroll
                    for (int kk = 0; kk < CIC / 2; ++kk) {
                        float a[CO_TILES];
#pragma unroll
                        for (int ct = 0; ct < CO_TILES; ++ct) a[ct] = wt[kk * 2 * COP + ct * 32];
#pragma unroll
                        for (int r = 0; r < NT; ++r) {
                            const float bv = xt[(kk * 2 * ID * IH + r) * RS];
#pragma unroll
                            for (int ct = 0; ct < CO_TILES; ++ct)
                                acc[r][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct], bv, acc[r][ct], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    const size_t HWo = (size_t)Ho * Wo, DHWo = (size_t)Do * HWo;
    float* yb = y + (size_t)b * Co * DHWo;
    const int mw = mw0 + l31;
    const int ow = 2 * mw + pw;
#pragma unroll
    for (int r = 0; r < NT; ++r) {
        const int md = md0 + dz0, mh = mh0 + hy0 + r;
        const int od = 2 * md + pd, oh = 2 * mh + ph;
        if (md >= D || mh >= H || mw >= W || od >= Do || oh >= Ho || ow >= Wo) continue;
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

// [A][Bc][27] (ConvTranspose3d [Ci,Co,27], or a Conv3d weight [Co_f,Ci_f,27] whose dgrad is wanted) -> [27][A][BcP]
__global__ void pack_deconv_weight(const float* __restrict__ w, float* __restrict__ packed, int A, int Bc, int bcp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * A * bcp) return;
    const int o = i % bcp;
    const int k = (i / bcp) % A;
    const int tap = i / (bcp * A);
    packed[i] = o < Bc ? w[((size_t)k * Bc + o) * 27 + tap] : 0.f;
}

template <int CO_TILES, int TD, int TH, int CIC>
int launch_deconv(const float* x, const float* wp, float* y, int B, int Ci, int Co, int D, int H, int W, int Do, int Ho,
                  int Wo, hipStream_t st) {
    using Cfg = DeconvCfg<CO_TILES, TD, TH, CIC>;
    const int tiles_d = (D + TD - 1) / TD, tiles_h = (H + TH - 1) / TH, tiles_w = (W + TW - 1) / TW;
    const long long nblk = 8LL * B * tiles_d * tiles_h * tiles_w;
    if (nblk > 0x7fffffffLL || (long long)D * H * W * 4 >= 0x80000000LL) return ECM_EUNSUP;
    auto kern = deconv3d_k3s2_mfma<CO_TILES, TD, TH, CIC>;
    static bool attr_set = false;
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

extern "C" int ecm_deconv3d_pack_weight(const float* w, float* packed, int Ci, int Co, void* stream) {
    ECM_CHECK_ARG(w && packed && Ci > 0 && Co > 0);
    const int cop = ((Co + 31) / 32) * 32;
    const int n = 27 * Ci * cop;
    hipLaunchKernelGGL(pack_deconv_weight, dim3((n + 255) / 256), dim3(256), 0, ecm_stream(stream), w, packed, Ci, Co,
                       cop);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_deconv3d_k3s2_fwd(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int D, int H,
                                     int W, int Do, int Ho, int Wo, void* stream) {
    ECM_CHECK_ARG(x && wpacked && y && B > 0 && D > 0 && H > 0 && W > 0);
    // output extent per dim is 2n (output_padding 1) or 2n-1 (dgrad of a stride-2 conv on an odd extent)
    if (Ci % 4 != 0 || Co < 1 || Co > 64) return ECM_EUNSUP;
    if (Do > 2 * D || Do < 2 * D - 1 || Ho > 2 * H || Ho < 2 * H - 1 || Wo > 2 * W || Wo < 2 * W - 1) return ECM_EUNSUP;
    hipStream_t st = ecm_stream(stream);
    if (Co > 32) return launch_deconv<2, 2, 8, 4>(x, wpacked, y, B, Ci, Co, D, H, W, Do, Ho, Wo, st);
    return launch_deconv<1, 4, 8, 4>(x, wpacked, y, B, Ci, Co, D, H, W, Do, Ho, Wo, st);
}
