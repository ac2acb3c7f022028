// The classifier's last layer, Conv3d(32 -> 1, k3, pad 1, no bias) (reference classif1/2/3[2], cmfsm.py:624,629,634),
// forward and weight gradient.  With ONE output channel the generic implicit GEMM wastes 31/32 of every MFMA, so the
// contraction is re-associated:
//   forward:  T[tap][u] = sum_ci w[ci,tap] x[ci,u]   (a 27 x Ci x voxels GEMM on the fp32 matrix cores, M = taps),
//             y[v] = sum_tap T[tap][v + off(tap)]     (27 shifted adds out of LDS, fixed order => deterministic),
//             one input plane at a time, marching along the disparity axis
//   wgrad:    gw[ci,tap] = sum_u x[ci,u] * gy[u - off(tap)]  (M = ci, N = taps, K = voxels: one MFMA per two voxels
//             covers all 32 x 27 products; the tap shift is a per-lane LDS offset on the small gy tile)
// Both are then bound by reading x once (212 MB at 576x960) instead of by ~92 GFLOP of padded MFMA work.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ forward, vector-ALU form
// 864 multiply-adds per output voxel is 5.7 G FMA for the batch-4 volume: 0.07 ms at the packed fp32 rate, under the 0.15-0.18 ms
// it takes to read x once -- while the matrix-core form below pads 27 taps to 32 and pays a load -> LDS -> MFMA -> LDS -> 27 reads
// chain per plane (0.31 ms).  Here every thread owns 2 x 2 x 4 outputs and walks the input channels TWO at a time: the halo tile
// of a channel pair sits in LDS interleaved [position][2 channels], so a v_pk_fma_f32 multiplies the pair (x_c0, x_c1)[pos] by
// the pair (w_c0, w_c1)[tap] into a pair of partial sums (added at the end) -- packed FMAs at full rate without any of the
// register-pair alignment problems packing along w would have (a tap shift just selects another pair register).
// Workgroup: 8 x 8 x 4 threads = 32 x 16 x 8 outputs; halo 34 x 18 x 10 per channel (1.49x, served by L2); the next pair's halo
// is prefetched through registers while the current one is multiplied; 49.0 + 3.4 KB of LDS, two workgroups per CU.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int VT_W = 32, VT_H = 16, VT_D = 8;
constexpr int VH_W = VT_W + 2, VH_H = VT_H + 2, VH_D = VT_D + 2;
// LDS row = 34 positions + 2 of padding; rows hy with (hy >> 1) odd are shifted by 2 positions (4 dwords): the 64 lanes of a
// window read (8 w-strips x 8 row pairs, 16 bytes each) then start at 8 different bank groups instead of 4 -- conflict-free
constexpr int VROW = 36;
constexpr int VPOS = VH_D * VH_H * VH_W;                   // 6120 halo positions
constexpr int VSLOTS = (VPOS + 255) / 256;                 // 24 per thread
constexpr int V_X_FLOATS = VH_D * VH_H * VROW * 2;
constexpr int V_W_FLOATS = 16 * 27 * 2;
constexpr int V_LDS_BYTES = (V_X_FLOATS + V_W_FLOATS + 2) * 4;   // + the scratch slot of the unused 24th halo slot
static_assert(2 * V_LDS_BYTES <= 160 * 1024, "two workgroups per CU");
__device__ __forceinline__ int v_slot(int dz, int hy, int wx) { return ((dz * VH_H + hy) * VROW + wx + 2 * ((hy >> 1) & 1)) * 2; }

// GN = true (round 4): x is the RAW output of the classifier's first convolution and the kernel applies GroupNorm(32) + ReLU
// (cmfsm.py:621-634: convbn_3d -> ReLU -> Conv3d 32 -> 1) while it stages the halo tile: h = max(fma(x, a_c, sh_c), 0) with
// a_c = rstd * gamma_c, sh_c = fma(-mean, a_c, beta_c) -- the expression of gn3d.hip's gn_affine, so h has the bits the
// stand-alone GroupNorm kernel would have written -- and positions outside the volume stay 0 (the convolution pads h, not x).
// The normalised tensor (849 MB per head at batch 4) is then never written or read: one read pass of x for the statistics
// (ecm_gn3d_stats) replaces the GroupNorm kernel's read + write.  Ci == 32 only (one channel per group).
template <bool GN>
__global__ __launch_bounds__(256, 2) void conv3d_c1_fwd_v(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int Ci, int D, int H, int W, int tiles_d,
                                                          int tiles_h, int tiles_w, const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [VH_D][VH_H][VROW][2]
    float* Ws = smem + V_X_FLOATS;          // [pair][27][2]
    // depth fastest: the tiles an XCD works on at one time then share their depth halo (2 of 10 planes) through its L2, the most
    // expensive one to re-read; the in-plane halo is shared with the next tiles of the run
    int bid = ecm_xcd_tile(blockIdx.x, gridDim.x);
    const int td = bid % tiles_d; bid /= tiles_d;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h;
    const int b = bid / tiles_h;
    const int w0 = tw * VT_W, h0 = th * VT_H, d0 = td * VT_D;
    const int tid = threadIdx.x;
    const int lw = tid & 7, lh = (tid >> 3) & 7, ld = tid >> 6;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
    const int npairs = Ci >> 1;

    // weights, interleaved by channel pair
    for (int i = tid; i < Ci * 27; i += 256) {
        const int c = i / 27, tap = i - c * 27;
        Ws[((c >> 1) * 27 + tap) * 2 + (c & 1)] = w[i];
    }
    // halo slots of this thread: global byte offset within a channel (out of range: dropped by the descriptor) and LDS slot
    unsigned goff[VSLOTS];
    int lslot[VSLOTS];
#pragma unroll
    for (int k = 0; k < VSLOTS; ++k) {
        const int p = tid + 256 * k;
        const int dz = p / (VH_H * VH_W), r = p - dz * (VH_H * VH_W), hy = r / VH_W, wx = r - hy * VH_W;
        const int gz = d0 - 1 + dz, gy = h0 - 1 + hy, gx = w0 - 1 + wx;
        const bool ok = p < VPOS && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        goff[k] = ok ? (unsigned)(((size_t)gz * H + gy) * W + gx) * 4u : 0x80000000u;
        lslot[k] = p < VPOS ? v_slot(dz, hy, wx) * 4 : (V_X_FLOATS + V_W_FLOATS) * 4;   // bytes; past the end: a scratch slot nobody reads
        asm volatile("" : "+v"(goff[k]), "+v"(lslot[k]));   // keep them in registers: recomputing costs more than the pk_fmas
    }
    const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)b * Ci * DHW, 0,
                                                      (unsigned)((size_t)Ci * DHW * 4), 0x00020000);
    const unsigned cstride = (unsigned)(DHW * 4);
    f32x2 rr[VSLOTS];                                        // (channel 2p, channel 2p+1) of a halo position: one ds_write_b64
    auto fetch = [&](int pair) {
        const unsigned s0 = (unsigned)(2 * pair) * cstride;
#pragma unroll
        for (int k = 0; k < VSLOTS; ++k) {
            rr[k].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, goff[k], s0, 0));
            rr[k].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, goff[k], s0 + cstride, 0));
        }
    };
    fetch(0);
    f32x2 acc[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][c][i] = f32x2{0.f, 0.f};
    // this thread's 4 x 4 x 6 window: rows 2 lh + hy; their shift is 2 positions for (lh + (hy >> 1)) odd
    const char* win0 = reinterpret_cast<const char*>(Xs) + v_slot(2 * ld, 2 * lh, 4 * lw) * 4;       // hy = 0, 1
    const char* win2 = reinterpret_cast<const char*>(Xs) + v_slot(2 * ld, 2 * lh + 2, 4 * lw) * 4;   // hy = 2, 3
    char* xs_c = reinterpret_cast<char*>(Xs);
    for (int pair = 0; pair < npairs; ++pair) {
        if (pair > 0) __syncthreads();                       // everyone has finished with the previous pair's tile
        if (GN) {
            // one channel per group (Ci == 32): group = channel; coefficients are wave-uniform
            const int c0 = 2 * pair;
            const float m0 = mean_rstd[(b * 32 + c0) * 2], r0 = mean_rstd[(b * 32 + c0) * 2 + 1];
            const float m1 = mean_rstd[(b * 32 + c0 + 1) * 2], r1 = mean_rstd[(b * 32 + c0 + 1) * 2 + 1];
            const float a0 = r0 * gamma[c0], a1 = r1 * gamma[c0 + 1];
            const float s0 = __builtin_fmaf(-m0, a0, beta[c0]), s1 = __builtin_fmaf(-m1, a1, beta[c0 + 1]);
#pragma unroll
            for (int k = 0; k < VSLOTS; ++k) {
                const bool in = goff[k] != 0x80000000u;
                rr[k].x = in ? fmaxf(__builtin_fmaf(rr[k].x, a0, s0), 0.f) : 0.f;
                rr[k].y = in ? fmaxf(__builtin_fmaf(rr[k].y, a1, s1), 0.f) : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < VSLOTS; ++k) *reinterpret_cast<f32x2*>(xs_c + lslot[k]) = rr[k];
        __syncthreads();
        if (pair + 1 < npairs) fetch(pair + 1);              // in flight under the multiply-adds below
        f32x2 wv[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) wv[t] = *reinterpret_cast<const f32x2*>(Ws + (pair * 27 + t) * 2);
#pragma unroll
        for (int dz = 0; dz < 4; ++dz)
#pragma unroll
            for (int hy = 0; hy < 4; ++hy) {
                f32x2 row[6];
                const float4* rp = reinterpret_cast<const float4*>((hy < 2 ? win0 : win2) + ((dz * VH_H + (hy & 1)) * VROW) * 8);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float4 v = rp[q];
                    row[2 * q] = f32x2{v.x, v.y};
                    row[2 * q + 1] = f32x2{v.z, v.w};
                }
#pragma unroll
                for (int od = 0; od < 2; ++od) {
                    const int kd = dz - od;
                    if (kd < 0 || kd > 2) continue;
#pragma unroll
                    for (int oh = 0; oh < 2; ++oh) {
                        const int kh = hy - oh;
                        if (kh < 0 || kh > 2) continue;
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                acc[od][oh][i] = __builtin_elementwise_fma(row[i + kw], wv[(kd * 3 + kh) * 3 + kw], acc[od][oh][i]);
                    }
                }
            }
    }
    float* yb = y + (size_t)b * DHW;
    const int ow = w0 + 4 * lw;
#pragma unroll
    for (int od = 0; od < 2; ++od)
#pragma unroll
        for (int oh = 0; oh < 2; ++oh) {
            const int z = d0 + 2 * ld + od, yy = h0 + 2 * lh + oh;
            if (z >= D || yy >= H || ow >= W) continue;
            float* dst = yb + ((size_t)z * H + yy) * W + ow;
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = acc[od][oh][i].x + acc[od][oh][i].y;
            if (ow + 3 < W && (((size_t)z * H + yy) * W + ow) % 4 == 0 && (reinterpret_cast<size_t>(yb) & 15) == 0)
                *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            else
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ow + i < W) dst[i] = o[i];
        }
}

// ------------------------------------------------------------------------------------------------ forward
// Depth-marching: a workgroup owns an MTH x MTW output footprint and walks a range of input planes z.  Per plane:
//   stage x[all 32 ci][(MTH+2) x 32 window] (register-prefetched one plane ahead),
//   T[tap][pos] = sum_ci w[ci,tap] x[ci,pos]              (27 x 32 x 256 GEMM: 8 MFMA row-blocks, 2 per wave),
//   S_kd[h,w] = sum_{kh,kw} T[kd,kh,kw][h+kh, w+kw]        (27 LDS reads per output position),
//   out[z+1] += S_0, out[z] += S_1, out[z-1] += S_2        (rolling registers; out[z-1] is complete and written).
// Only one plane of T lives in LDS (27.6 KB instead of 83 KB for a 3-D tile), so two workgroups share a CU, and the
// halo is re-read in-plane only (1.42x instead of 3.2x).
constexpr int MTH = 6, MTW = 30;                          // output footprint per plane
constexpr int MIH = MTH + 2, MPOS = MIH * 32;             // 8 window rows of 32 columns = 256 positions (1 per thread)
constexpr int MXS = MPOS + 32;                            // channel stride of the x image (== 32 mod 64: the two k halves
                                                          //   of a B-operand read land on different banks)
constexpr int M_LDS_BYTES = (32 * MXS + 27 * MPOS) * 4;   // 36.9 KB + 27.6 KB
static_assert(MPOS == 256, "one window position per thread");

__global__ __launch_bounds__(256, 2) void conv3d_c1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, int Ci, int D, int H, int W, int zchunk,
                                                        int tiles_z, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                 // [32 ci][MXS]
    float* Ts = smem + 32 * MXS;      // [27 taps][MPOS]
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int tz = bid % tiles_z;
    const int b = bid / tiles_z;
    const int z0 = tz * zchunk, z1 = z0 + zchunk < D ? z0 + zchunk : D;      // output planes [z0, z1)
    const int oh0 = th * MTH, ow0 = tw * MTW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;

    // A operand (weights) for every k-step, kept in registers: lane holds w[ci = 2*kk + half][tap = l31]
    float wa[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int ci = 2 * kk + half;
        wa[kk] = (l31 < 27 && ci < Ci) ? w[(size_t)ci * 27 + l31] : 0.f;
    }
    // staging: thread owns window position tid = (row, column) for all channels; in-plane offset is plane-invariant
    const int prow = tid >> 5, pcol = tid & 31;
    const int gyy = oh0 - 1 + prow, gx = ow0 - 1 + pcol;
    const bool inplane = (unsigned)gyy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const unsigned inoff = (unsigned)(gyy * W + gx) * 4u;
    const unsigned plane_bytes = (unsigned)DHW * 4u;
    const int nci = Ci < 32 ? Ci : 32;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * Ci * DHW), 0,
                                                       (unsigned)nci * plane_bytes, 0x00020000);
    float xr[32];
    auto prefetch = [&](int z) {      // input plane z (may be outside [0, D): zeros)
        const bool ok = inplane && (unsigned)z < (unsigned)D;
        const unsigned off = ok ? (unsigned)z * (unsigned)HW * 4u + inoff : 0x80000000u;
#pragma unroll
        for (int cc = 0; cc < 32; ++cc)
            xr[cc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off + (unsigned)cc * plane_bytes, 0, 0));
    };
    // output position of this thread (threads >= MTH*MTW only help with staging and the GEMM)
    const bool owner = tid < MTH * MTW;
    const int ohy = tid / MTW, oxo = tid - ohy * MTW;
    const bool ovalid = owner && oh0 + ohy < H && ow0 + oxo < W;
    float* yp = y + (size_t)b * DHW + (size_t)(oh0 + ohy) * W + (ow0 + oxo);
    float r_prev = 0.f, r_cur = 0.f;                       // partial sums of out[z-1], out[z]

    prefetch(z0 - 1);
    for (int z = z0 - 1; z <= z1; ++z) {                   // input planes z0-1 .. z1 (zero planes outside the volume)
        __syncthreads();                                   // previous plane's T reads and x reads are done
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) Xs[cc * MXS + tid] = xr[cc];
        __syncthreads();
        if (z < z1) prefetch(z + 1);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float bv = Xs[(kk * 2 + half) * MXS + (wave * 2 + r) * 32 + l31];
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[kk], bv, acc[r], 0, 0, 0);
            }
        // T[tap][row][col] -> LDS   (D fragment: column = l31, register i = tap (i&3) + 8*(i>>2) + 4*half)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tap = (i & 3) + 8 * (i >> 2) + 4 * half;
                if (tap < 27) Ts[tap * MPOS + (wave * 2 + r) * 32 + l31] = acc[r][i];
            }
        __syncthreads();
        if (owner) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;            // S_kd of this plane at (ohy, oxo), taps in fixed order
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int kh = t9 / 3, kw = t9 % 3;
                const int o = (ohy + kh) * 32 + oxo + kw;
                s0 += Ts[(0 * 9 + t9) * MPOS + o];
                s1 += Ts[(1 * 9 + t9) * MPOS + o];
                s2 += Ts[(2 * 9 + t9) * MPOS + o];
            }
            // plane z feeds out[z+1] (kd=0), out[z] (kd=1), out[z-1] (kd=2); out[z-1] is now complete
            const float done = r_prev + s2;
            if (ovalid && z - 1 >= z0 && z - 1 < z1) yp[(size_t)(z - 1) * HW] = done;
            r_prev = r_cur + s1;
            r_cur = s0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
// gx[b,ci,d,h,w] = sum_tap w[ci,tap] * gy[b, d-kd+1, h-kh+1, w-kw+1]: 27 values of the single-channel gy per voxel fan out
// to all Ci channels.  Store-bound (Ci x the volume is written, gy is 1/Ci of that): one workgroup stages a 3 x (DTH+2) x
// (DTW+2) gy halo tile in LDS, every thread keeps the 27 neighbours of its 4 consecutive voxels in registers (54 LDS
// reads) and streams out one float4 per channel; the weights come through the scalar cache.
constexpr int DTH = 8, DTW = 128;                         // output tile of one depth plane: 8 rows x 128 columns
constexpr int DGW = DTW + 4;                              // halo row stride (1 left, 1 right, +2 so rows stay 4-aligned)

__global__ __launch_bounds__(256) void conv3d_c1_dgrad(const float* __restrict__ gy, const float* __restrict__ w,
                                                       float* __restrict__ gx, int Ci, int D, int H, int W, int tiles_h,
                                                       int tiles_w) {
    __shared__ float Gs[3 * (DTH + 2) * DGW];
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int d = bid % D;
    const int b = bid / D;
    const int h0 = th * DTH, w0 = tw * DTW;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
    const float* gb = gy + (size_t)b * DHW;
    for (int e = threadIdx.x; e < 3 * (DTH + 2) * (DTW + 2); e += 256) {
        const int xx = e % (DTW + 2), r = e / (DTW + 2);
        const int hy = r % (DTH + 2), dz = r / (DTH + 2);
        const int gz = d - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + xx;
        float v = 0.f;
        if ((unsigned)gz < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
            v = gb[(size_t)gz * HW + (size_t)gh * W + gw];
        Gs[(dz * (DTH + 2) + hy) * DGW + xx] = v;
    }
    __syncthreads();
    const int hy = threadIdx.x >> 5, x4 = (threadIdx.x & 31) * 4;
    const int oh = h0 + hy, ow = w0 + x4;
    if (oh >= H || ow >= W) return;
    // g[j][i]: row j = dz*3 + hh of the 3 x 3 neighbourhood rows, i = 0..5 the columns ow-1 .. ow+4 (halo coordinates x4 .. x4+5)
    float g[9][6];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) g[j][i] = Gs[((j / 3) * (DTH + 2) + hy + (j % 3)) * DGW + x4 + i];
    const bool full = ow + 3 < W && (W & 3) == 0;
    float* op = gx + (size_t)b * Ci * DHW + (size_t)d * HW + (size_t)oh * W + ow;
    for (int ci = 0; ci < Ci; ++ci) {
        const float* wc = w + (size_t)ci * 27;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    // gy index d-kd+1 -> halo plane (2-kd); h-kh+1 -> halo row offset (2-kh); w-kw+1 -> column offset (2-kw)
                    const float wv = wc[kd * 9 + kh * 3 + kw];
                    const int j = (2 - kd) * 3 + (2 - kh), i0 = 2 - kw;
                    a0 = fmaf(wv, g[j][i0 + 0], a0);
                    a1 = fmaf(wv, g[j][i0 + 1], a1);
                    a2 = fmaf(wv, g[j][i0 + 2], a2);
                    a3 = fmaf(wv, g[j][i0 + 3], a3);
                }
        float* o = op + (size_t)ci * DHW;
        if (full) ecm_st_stream(o, make_float4(a0, a1, a2, a3));
        else {
            o[0] = a0;
            if (ow + 1 < W) o[1] = a1;
            if (ow + 2 < W) o[2] = a2;
            if (ow + 3 < W) o[3] = a3;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int GTD = 2, GTH = 8, GTW = 32, GNV = GTD * GTH * GTW;       // 512 voxels of x per tile (no halo on x)
constexpr int GXSTR = GNV + 1;                                           // odd channel stride: 32 lanes (ci) -> 32 banks
constexpr int GRS = 35, GPS = 361;                                       // gy halo row / plane strides: 3 and 9 (mod 32)
constexpr int G_GYF = (GTD + 2) * GPS;                                   //   => the 27 tap offsets hit 27 distinct banks
constexpr int G_LDS_BYTES = (32 * GXSTR + G_GYF) * 4;                   // 71 KB: two workgroups per CU
static_assert((GTH + 2) * GRS <= GPS, "plane stride too small");

// GN = true: x is the raw conv output, normalised + ReLU'd on its way into LDS as in conv3d_c1_fwd_v<true> (positions outside the
// volume stay 0: they pair with in-volume gy values of the halo).
template <bool GN>
__global__ __launch_bounds__(256) void conv3d_c1_wgrad(const float* __restrict__ x, const float* __restrict__ gy,
                                                       float* __restrict__ partial, int B, int Ci, int D, int H, int W,
                                                       int tiles_d, int tiles_h, int tiles_w, const float* __restrict__ mean_rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [32 ci][GXSTR]
    float* Gs = smem + 32 * GXSTR;          // gy halo tile [(GTD+2)][GPS] (rows of GRS)
    float* Ps = Xs;                         // [4 waves][1024] end-of-kernel reduction, reuses the x image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const size_t HW = (size_t)H * W, DHW = (size_t)D * HW;
    // lane's tap (B operand column) and its offset inside the gy halo tile relative to the voxel's own position
    const int tap = l31 < 27 ? l31 : 0;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    // x voxel u=(dz,hy,xx) pairs with gy[u - (tap-1)]; the halo tile starts one voxel before the tile, so that is
    // halo index (dz+2-kd, hy+2-kh, xx+2-kw)
    const int toff = (2 - kd) * GPS + (2 - kh) * GRS + (2 - kw);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const unsigned xplane = (unsigned)DHW * 4u;
    const int ntiles = B * tiles_d * tiles_h * tiles_w;       // < 2^31 (checked by the host)
    // Register-pipelined staging: the NEXT tile's 64 x values and 6 gy-halo values per thread are requested through
    // buffer descriptors (out-of-volume positions carry the offset 0x80000000 and read as 0) before this tile's MFMAs,
    // so a tile costs max(load, MFMA) instead of 10 exposed memory round trips (the first version: 85K cycles per tile).
    constexpr int NGH = (GTD + 2) * (GTH + 2) * (GTW + 2);                 // gy halo elements
    constexpr int PG = (NGH + 255) / 256;
    float xr[64], gr[PG];
    bool xin[2] = {false, false};                                         // GN: is the prefetched x position inside the volume?
    int xb_next = 0;                                                      // GN: sample index of the prefetched tile
    int gdst[PG];                                                         // tile-invariant LDS slot of each halo element
#pragma unroll
    for (int j = 0; j < PG; ++j) {
        const int e = tid + j * 256;
        const int xx = e % (GTW + 2), hy = (e / (GTW + 2)) % (GTH + 2), dz = e / ((GTW + 2) * (GTH + 2));
        gdst[j] = e < NGH ? dz * GPS + hy * GRS + xx : -1;
    }
    const int nci = Ci < 32 ? Ci : 32;
    auto prefetch = [&](unsigned tile) {
        const bool live = tile < (unsigned)ntiles;
        unsigned r = live ? tile : 0u;
        const int tw = (int)(r % (unsigned)tiles_w); r /= (unsigned)tiles_w;
        const int th = (int)(r % (unsigned)tiles_h); r /= (unsigned)tiles_h;
        const int td = (int)(r % (unsigned)tiles_d);
        const int b = (int)(r / (unsigned)tiles_d);
        const int d0 = td * GTD, h0 = th * GTH, w0 = tw * GTW;
        unsigned off[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int v = tid + j * 256;
            const int xx = v % GTW, hy = (v / GTW) % GTH, dz = v / (GTW * GTH);
            const int gz = d0 + dz, gyy = h0 + hy, gx = w0 + xx;
            const bool ok = live && gz < D && gyy < H && gx < W;
            off[j] = ok ? (unsigned)(gz * (int)HW + gyy * W + gx) * 4u : 0x80000000u;
            if (GN) xin[j] = ok;
        }
        if (GN) xb_next = b;
        const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * Ci * DHW), 0,
                                                           (unsigned)nci * xplane, 0x00020000);
#pragma unroll
        for (int cc = 0; cc < 32; ++cc)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                xr[cc * 2 + j] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off[j] + (unsigned)cc * xplane, 0, 0));
        const auto grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy + (size_t)b * DHW), 0, xplane, 0x00020000);
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int e = tid + j * 256;
            const int xx = e % (GTW + 2), hy = (e / (GTW + 2)) % (GTH + 2), dz = e / ((GTW + 2) * (GTH + 2));
            const int gz = d0 - 1 + dz, gyy = h0 - 1 + hy, gx = w0 - 1 + xx;
            const bool ok = live && e < NGH && (unsigned)gz < (unsigned)D && (unsigned)gyy < (unsigned)H && (unsigned)gx < (unsigned)W;
            gr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  grs, ok ? (unsigned)(gz * (int)HW + gyy * W + gx) * 4u : 0x80000000u, 0, 0));
        }
    };
    prefetch(blockIdx.x);
    for (unsigned tile = blockIdx.x; tile < (unsigned)ntiles; tile += gridDim.x) {
        __syncthreads();
        if (GN) {
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) {
                const float m = mean_rstd[(xb_next * 32 + cc) * 2], r = mean_rstd[(xb_next * 32 + cc) * 2 + 1];
                const float a = r * gamma[cc], sh = __builtin_fmaf(-m, a, beta[cc]);
#pragma unroll
                for (int j = 0; j < 2; ++j) xr[cc * 2 + j] = xin[j] ? fmaxf(__builtin_fmaf(xr[cc * 2 + j], a, sh), 0.f) : 0.f;
            }
        }
#pragma unroll
        for (int cc = 0; cc < 32; ++cc)
#pragma unroll
            for (int j = 0; j < 2; ++j) Xs[cc * GXSTR + tid + j * 256] = xr[cc * 2 + j];
#pragma unroll
        for (int j = 0; j < PG; ++j)
            if (gdst[j] >= 0) Gs[gdst[j]] = gr[j];
        __syncthreads();
        prefetch(tile + gridDim.x);
        // k-steps: two x-adjacent voxels; wave handles rows {wave, wave+4, ...} of the 16 tile rows
        const float* xa = Xs + l31 * GXSTR + half;
        const float* gb = Gs + toff + half;
#pragma unroll
        for (int rr = 0; rr < GTD * GTH / 4; ++rr) {
            const int row = wave + rr * 4;
            const int dz = row / GTH, hy = row % GTH;
#pragma unroll
            for (int xx = 0; xx < GTW; xx += 2) {
                const float a = xa[(dz * GTH + hy) * GTW + xx];
                const float bv = l31 < 27 ? gb[dz * GPS + hy * GRS + xx] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
            }
        }
    }
    // D[ci][tap]: column = tap = l31, rows ci = (i&3) + 8*(i>>2) + 4*half.  Sum the 4 waves, then one partial per block.
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2) + 4 * half;
        Ps[wave * 1024 + ci * 32 + l31] = acc[i];
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const int ci = e >> 5, t = e & 31;
        if (t < 27 && ci < Ci)
            partial[(size_t)blockIdx.x * Ci * 27 + ci * 27 + t] = (Ps[e] + Ps[1024 + e]) + (Ps[2048 + e] + Ps[3072 + e]);
    }
}

// gw[i] = sum_p partial[p][i] in a fixed order: 32 outputs per workgroup, 8 lane groups each sum every 8th partial
// (same scheme as wgrad_reduce in conv3d_wgrad.hip; one thread per output walking 512 partials took ~0.1 ms).
__global__ __launch_bounds__(256) void c1_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ gw, int n, int P) {
    __shared__ float sm[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    float s = 0.f;
    if (i < n) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = grp;
        for (; p + 24 < P; p += 32) {
            s0 += partial[(size_t)p * n + i];
            s1 += partial[(size_t)(p + 8) * n + i];
            s2 += partial[(size_t)(p + 16) * n + i];
            s3 += partial[(size_t)(p + 24) * n + i];
        }
        for (; p < P; p += 8) s0 += partial[(size_t)p * n + i];
        s = (s0 + s1) + (s2 + s3);
    }
    sm[grp][o] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sm[g][o];
        gw[i] = t;
    }
}

inline int c1_workers(long long ntiles) { return (int)(ntiles < 512 ? ntiles : 512); }
inline long long c1_tiles(int B, int D, int H, int W) {
    return (long long)B * ((D + GTD - 1) / GTD) * ((H + GTH - 1) / GTH) * ((W + GTW - 1) / GTW);
}

}  // namespace

namespace {
template <bool GN>
int launch_c1_fwd_v(const float* x, const float* w, float* y, int B, int Ci, int D, int H, int W, const float* mean_rstd,
                    const float* gamma, const float* beta, void* stream) {
    const int td = (D + VT_D - 1) / VT_D, th = (H + VT_H - 1) / VT_H, tw = (W + VT_W - 1) / VT_W;
    const long long nb = (long long)B * td * th * tw;
    if (nb > 0x7fffffffLL) return ECM_EUNSUP;
    const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(conv3d_c1_fwd_v<GN>), V_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv3d_c1_fwd_v<GN>, dim3((unsigned)nb), dim3(256), V_LDS_BYTES, ecm_stream(stream), x, w, y, Ci, D, H, W, td,
                       th, tw, mean_rstd, gamma, beta);
    return ECM_LAUNCH_RESULT();
}
}  // namespace

extern "C" int ecm_conv3d_c1_gn_fwd(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* w,
                                    float* y, int B, int Ci, int D, int H, int W, void* stream) {
    ECM_CHECK_ARG(x && mean_rstd && gamma && beta && w && y && B > 0 && D > 0 && H > 0 && W > 0);
    if (Ci != 32 || (long long)D * H * W * 4 * 32 >= 0x7fffffffLL) return ECM_EUNSUP;      // one channel per GroupNorm group
    return launch_c1_fwd_v<true>(x, w, y, B, Ci, D, H, W, mean_rstd, gamma, beta, stream);
}

extern "C" int ecm_conv3d_c1_fwd(const float* x, const float* w, float* y, int B, int Ci, int D, int H, int W, void* stream) {
    ECM_CHECK_ARG(x && w && y && B > 0 && Ci > 0 && D > 0 && H > 0 && W > 0);
    if (Ci > 32 || (long long)D * H * W * 4 * 32 >= 0x7fffffffLL) return ECM_EUNSUP;
    static const bool valu = [] { const char* v = getenv("ECM_C1_VALU"); return !(v && v[0] == '0'); }();
    if (valu && Ci % 2 == 0) return launch_c1_fwd_v<false>(x, w, y, B, Ci, D, H, W, nullptr, nullptr, nullptr, stream);
    const int tiles_h = (H + MTH - 1) / MTH, tiles_w = (W + MTW - 1) / MTW;
    // split the disparity axis only as far as needed to give the chip ~3 rounds of workgroups (each chunk re-reads 2 planes)
    const long long cols = (long long)B * tiles_h * tiles_w;
    int tiles_z = (int)((3 * 512 + cols - 1) / cols);
    if (tiles_z < 1) tiles_z = 1;
    if (tiles_z > (D + 7) / 8) tiles_z = (D + 7) / 8;
    const int zchunk = (D + tiles_z - 1) / tiles_z;
    tiles_z = (D + zchunk - 1) / zchunk;
    const long long nblk = cols * tiles_z;
    if (nblk > 0x7fffffffLL) return ECM_EUNSUP;
    {
        const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(conv3d_c1_fwd), M_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(conv3d_c1_fwd, dim3((unsigned)nblk), dim3(256), M_LDS_BYTES, ecm_stream(stream), x, w, y, Ci, D, H, W,
                       zchunk, tiles_z, tiles_h, tiles_w);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_conv3d_c1_dgrad(const float* gy, const float* w, float* gx, int B, int Ci, int D, int H, int W, void* stream) {
    ECM_CHECK_ARG(gy && w && gx && B > 0 && Ci > 0 && D > 0 && H > 0 && W > 0);
    const int tiles_h = (H + DTH - 1) / DTH, tiles_w = (W + DTW - 1) / DTW;
    const long long nblk = (long long)B * D * tiles_h * tiles_w;
    if (nblk > 0x7fffffffLL) return ECM_EUNSUP;
    hipLaunchKernelGGL(conv3d_c1_dgrad, dim3((unsigned)nblk), dim3(256), 0, ecm_stream(stream), gy, w, gx, Ci, D, H, W, tiles_h,
                       tiles_w);
    return ECM_LAUNCH_RESULT();
}

extern "C" long long ecm_conv3d_c1_wgrad_scratch_bytes(int B, int Ci, int D, int H, int W) {
    if (B <= 0 || Ci <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return (long long)c1_workers(c1_tiles(B, D, H, W)) * Ci * 27 * (long long)sizeof(float);
}

namespace {
template <bool GN>
int launch_c1_wgrad(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes, int B, int Ci, int D, int H,
                    int W, const float* mean_rstd, const float* gamma, const float* beta, void* stream) {
    if (Ci > 32 || (long long)D * H * W * 4 * 32 >= 0x7fffffffLL || c1_tiles(B, D, H, W) >= 0x7fffffffLL) return ECM_EUNSUP;
    if (scratch_bytes < ecm_conv3d_c1_wgrad_scratch_bytes(B, Ci, D, H, W)) return ECM_ESCRATCH;
    const int tiles_d = (D + GTD - 1) / GTD, tiles_h = (H + GTH - 1) / GTH, tiles_w = (W + GTW - 1) / GTW;
    const int P = c1_workers(c1_tiles(B, D, H, W));
    hipStream_t st = ecm_stream(stream);
    {
        const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(conv3d_c1_wgrad<GN>), G_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    float* partial = static_cast<float*>(scratch);
    hipLaunchKernelGGL(conv3d_c1_wgrad<GN>, dim3(P), dim3(256), G_LDS_BYTES, st, x, gy, partial, B, Ci, D, H, W, tiles_d, tiles_h,
                       tiles_w, mean_rstd, gamma, beta);
    const int n = Ci * 27;
    hipLaunchKernelGGL(c1_wgrad_reduce, dim3((n + 31) / 32), dim3(256), 0, st, partial, gw, n, P);
    return ECM_LAUNCH_RESULT();
}
}  // namespace

extern "C" int ecm_conv3d_c1_wgrad(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes, int B,
                                   int Ci, int D, int H, int W, void* stream) {
    ECM_CHECK_ARG(x && gy && gw && scratch && B > 0 && Ci > 0 && D > 0 && H > 0 && W > 0);
    return launch_c1_wgrad<false>(x, gy, gw, scratch, scratch_bytes, B, Ci, D, H, W, nullptr, nullptr, nullptr, stream);
}

extern "C" int ecm_conv3d_c1_gn_wgrad(const float* x, const float* mean_rstd, const float* gamma, const float* beta, const float* gy,
                                      float* gw, void* scratch, long long scratch_bytes, int B, int Ci, int D, int H, int W,
                                      void* stream) {
    ECM_CHECK_ARG(x && mean_rstd && gamma && beta && gy && gw && scratch && B > 0 && D > 0 && H > 0 && W > 0);
    if (Ci != 32) return ECM_EUNSUP;
    return launch_c1_wgrad<true>(x, gy, gw, scratch, scratch_bytes, B, Ci, D, H, W, mean_rstd, gamma, beta, stream);
}
