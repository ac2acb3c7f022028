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

#ifdef WG_PROFILE
__device__ unsigned long long wg_prof[4 * 8];     // [wave][phase] cycles of workgroup (0,0); debugging aid only
#define WG_T(i) do { const unsigned long long now_ = clock64(); prof[i] += now_ - last; last = now_; } while (0)
#else
#define WG_T(i) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CT = 32;        // channel tile (both co and ci)
constexpr int TWV = 16;       // output voxels per tile row

// KD = 3: the 3x3x3 Conv3d.  KD = 1: a 3x3 Conv2d seen as a depth-1 volume with no taps and no padding along depth
// (the encoder's convbn, cmfsm.py:37-47) -- same staging, 9 taps.
// KH x KW taps with dilation DIL in the plane (KD = 1 only: the encoder's dilated / strided / 1x1 layers and the sheared
// 3x5 class convolution of the collapsed cost volume, conv2d.hip); the 3-D layers are always 3x3x3, dilation 1.
template <int STRIDE, int TD, int TH, int KD, int KH = 3, int KW = 3, int DIL = 1>
struct WgCfg {
    static_assert(KD == 1 || (KH == 3 && KW == 3 && DIL == 1), "3-D layers are 3x3x3, dilation 1");
    static constexpr int NTAPS = KH * KW * KD, PADD = KD / 2;
    // wave plan: 27 taps -> 4 tap groups of 7 (the last has 6), every wave runs all k-steps;
    //            15 taps -> 4 tap groups of 4 (the last has 3);  9 taps -> 2 tap groups (5 + 4) x 2 halves of the k-steps
    //            (each half writes its own partial);
    //            1 tap -> 4 quarters of the k-steps.
    static constexpr int TG = NTAPS >= 15 ? 4 : NTAPS >= 2 ? 2 : 1, KG = 4 / TG, MAXNT = (NTAPS + TG - 1) / TG;
    static constexpr int ID = (TD - 1) * STRIDE + KD, IH = (TH - 1) * STRIDE + (KH - 1) * DIL + 1,
                         IW = (TWV - 1) * STRIDE + (KW - 1) * DIL + 1;
    static constexpr int RS = IW;
    static constexpr int XCH = ID * IH * RS;
    static constexpr int XSTR = (XCH % 2 == 0) ? XCH + 1 : XCH + 2; // odd => 32 lanes (ci) hit 32 banks; slot XCH is a pad
    static constexpr int NV = TD * TH * TWV;
    static constexpr int GSTR = NV + 1;                              // odd
    static constexpr int LDS_FLOATS = CT * XSTR + CT * GSTR;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};


// One staged tile: TD*TH*TWV/2 k-steps (two x-adjacent voxels each), taps [T0, T0+7) of this wave.
// With one wave per SIMD nothing else hides LDS latency, so the operands of k-step i+1 are read while the MFMAs of
// k-step i execute: explicit register double buffer, with a sched_barrier per k-step so that hipcc cannot sink the
// reads back next to their use (left to itself it waits lgkmcnt(0) in front of each MFMA group).
// The NEXT tile's global loads are spread over the k-steps too (issue(i), i < NLOADS): issued in one burst they
// stall the wave on the 64-entry vmcnt window for ~13K cycles per tile with the matrix core idle (measured 17.6 %).
template <int STRIDE, int TD, int TH, int KD, int KH, int KW, int DIL, int TGI, int KGI, int NLOADS, class Issue>
__device__ __forceinline__ void wg_tile(const float* __restrict__ ga, const float* __restrict__ xb,
                                        f32x16 (&acc)[WgCfg<STRIDE, TD, TH, KD, KH, KW, DIL>::MAXNT], Issue&& issue) {
    using Cfg = WgCfg<STRIDE, TD, TH, KD, KH, KW, DIL>;
    constexpr int IH = Cfg::IH, RS = Cfg::RS;
    constexpr int T0 = TGI * Cfg::MAXNT;                                         // this wave's taps [T0, T0+NT)
    constexpr int NT = (T0 + Cfg::MAXNT <= Cfg::NTAPS) ? Cfg::MAXNT : Cfg::NTAPS - T0;
    static_assert(NT >= 1, "every wave owns at least one tap");
    constexpr int KSA = TD * TH * TWV / 2;
    constexpr int KS0 = KGI * (KSA / Cfg::KG), KS = KS0 + KSA / Cfg::KG;         // this wave's k-steps [KS0, KS)
    constexpr int KSI = (KS - KS0) * 3 / 4;                   // all issued within the first 3/4 of the k-steps, so that
    constexpr int LPK = (NLOADS + KSI - 1) / KSI;             // their latency is not exposed at the LDS stores that follow
    auto b_off = [](int ks, int t) constexpr {
        const int xx = (ks * 2) % TWV, hy = ((ks * 2) / TWV) % TH, dz = (ks * 2) / (TWV * TH);
        const int tap = T0 + t, kd = KD == 3 ? tap / 9 : 0, kh = (tap / KW) % KH, kw = tap % KW;
        return ((dz * STRIDE + kd) * IH + hy * STRIDE + kh * DIL) * RS + xx * STRIDE + kw * DIL;
    };
    float a_cur = ga[KS0 * 2], b_cur[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b_cur[t] = xb[b_off(KS0, t)];
#pragma unroll
    for (int ks = KS0; ks < KS; ++ks) {
        float a_nxt = 0.f, b_nxt[NT];
        if (ks + 1 < KS) {
            a_nxt = ga[(ks + 1) * 2];                          // voxel index of the pair == 2*ks
#pragma unroll
            for (int t = 0; t < NT; ++t) b_nxt[t] = xb[b_off(ks + 1, t)];
        }
#pragma unroll
        for (int q = 0; q < LPK; ++q)
            if ((ks - KS0) * LPK + q < NLOADS) issue((ks - KS0) * LPK + q);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);     // nothing moves across k-steps: next step's reads stay ahead of these MFMAs
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < NT; ++t) b_cur[t] = b_nxt[t];
    }
}

// ---- Winograd F(2x2,3x3) form of the stride-1 weight gradient (direct along depth) ---------------------------------------
//   gw = G^T [ sum_tiles (A gy_tile A^T) (.) (B^T x_patch B) ] G
// per 2x2 output tile: 16 multiplies per (co, ci, kd) instead of 36.  Frequency (i,j): wave i owns row i, i.e. the four
// accumulators j = 0..3 per depth tap.  The operands are NOT staged in transformed form: every lane builds them on the fly
// from the SAME raw LDS tiles the direct kernel uses (x halo tile [ci][dz][hy][xx], gy tile [co][voxels]) -- for its
// frequency row a patch contributes two of its four rows, so a B fragment costs 8 LDS reads + 8 adds per four operands, an A
// fragment 4 reads + 4 adds -- so staging, prefetch pipeline and LDS footprint are exactly the direct kernel's.
// GEMM per frequency: D[co][ci] += sum_tile A[co][tile] B[tile][ci]; a k-step = two x-adjacent tiles (lane half = tile).
template <int TD, int TH, int KD, int I, int NLOADS, class Issue>
__device__ __forceinline__ void wg_tile_wino(const float* __restrict__ gaw, const float* __restrict__ xbw,
                                             f32x16 (&acc)[4 * KD], Issue&& issue) {
    using Cfg = WgCfg<1, TD, TH, KD>;
    constexpr int IH = Cfg::IH, RS = Cfg::RS, NP = TD + KD - 1;
    constexpr int PAIRS = TWV / 4, KS = (TH / 2) * PAIRS;                 // k-steps: tile rows x pairs of tiles
    // frequency row I of B^T d: rows (RA, RB) of the patch, RA - RB (I = 1: RA + RB)
    constexpr int RA = I == 0 ? 0 : I == 2 ? 2 : 1, RB = I == 0 ? 2 : I == 1 ? 2 : I == 2 ? 1 : 3;
    // Register budget (one wave per SIMD, 512 registers): 4*KD accumulators (192) + the next tile's prefetch registers (104)
    // leave room for ONE set of raw operands: the reads of k-step ks+1 are issued before the MFMAs of ks and turned into
    // operands after them, in the shadow of the last MFMAs.
    float xr_[NP][2][4], gr_[TD][2][2];                                   // raw operands of the NEXT k-step: [plane][row][column]
    float bq[NP][4], aq[TD][4];                                           // operands of the current k-step
    auto read_raw = [&](int ks) {                                         // ks is wave-uniform: scalar address arithmetic
        const int tr = ks / PAIRS, pr = ks % PAIRS;
        const float* xk = xbw + 2 * tr * RS + 4 * pr;
        const float* gk = gaw + 2 * tr * TWV + 4 * pr;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xr_[p][0][j] = xk[(p * IH + RA) * RS + j];
                xr_[p][1][j] = xk[(p * IH + RB) * RS + j];
            }
#pragma unroll
        for (int d = 0; d < TD; ++d)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) gr_[d][r][c] = gk[(d * TH + r) * TWV + c];
    };
    auto make_operands = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = I == 1 ? xr_[p][0][j] + xr_[p][1][j] : xr_[p][0][j] - xr_[p][1][j];
            bq[p][0] = e[0] - e[2]; bq[p][1] = e[1] + e[2]; bq[p][2] = e[2] - e[1]; bq[p][3] = e[1] - e[3];
        }
#pragma unroll
        for (int d = 0; d < TD; ++d) {
            // row I of A g: g0 | g0 + g1 | g0 - g1 | -g1; then columns m0 | m0 + m1 | m0 - m1 | -m1
            float m[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
                m[c] = I == 0 ? gr_[d][0][c] : I == 1 ? gr_[d][0][c] + gr_[d][1][c]
                     : I == 2 ? gr_[d][0][c] - gr_[d][1][c] : -gr_[d][1][c];
            aq[d][0] = m[0]; aq[d][1] = m[0] + m[1]; aq[d][2] = m[0] - m[1]; aq[d][3] = -m[1];
        }
    };
    // The k-step loop stays ROLLED (fully unrolled, hipcc's register allocation of 16 x 24 MFMAs with the interleaved
    // prefetch spills ~200 registers); the next tile's global loads are issued in NPH bursts between groups of k-steps.
    // Burst length matters: a long burst fills the memory pipeline's queue and the wave sits on it with the matrix core idle
    // (3-D: 3 bursts of 43 loads -> 7 of 19: 1.95 -> 1.69 ms; one per k-step would need the loop unrolled, which spills).
    constexpr int NPH = KD == 1 ? 16 : 8, KPP = KS / NPH, LPP = (NLOADS + NPH - 2) / (NPH - 1);   // bursts in the first NPH-1 phases
    constexpr int KUNROLL = KD == 1 ? 8 : 1;   // 2-D: 4 MFMAs per k-step -- a whole phase unrolled (loop control and address
                                                // arithmetic gone: -6..-12 %); 3-D: 24 per k-step, and unrolling by 2 already spills
    static_assert(KS % NPH == 0, "k-steps split into phases");
    read_raw(0);
    make_operands();
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph < NPH - 1) {
#pragma unroll
            for (int q = 0; q < LPP; ++q)
                if (ph * LPP + q < NLOADS) issue(ph * LPP + q);
        }
#pragma unroll KUNROLL
        for (int ks = ph * KPP; ks < (ph + 1) * KPP; ++ks) {
            if (ks + 1 < KS) read_raw(ks + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kd = 0; kd < KD; ++kd)
#pragma unroll
                    for (int d = 0; d < TD; ++d)
                        acc[j * KD + kd] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[d][j], bq[d + kd][j], acc[j * KD + kd], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);     // the reads above stay ahead of these MFMAs, the transform below behind them
            if (ks + 1 < KS) make_operands();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int STRIDE, int TD, int TH, int KD, int KH = 3, int KW = 3, int DIL = 1, bool WINO = false>
__global__ __launch_bounds__(256, KD == 3 ? 1 : 2) void conv3d_wgrad_mfma(const float* __restrict__ x, const float* __restrict__ gy,
                                                         float* __restrict__ partial, int B, int Ci, int Co, int D,
                                                         int H, int W, int Do, int Ho, int Wo, int tiles_d, int tiles_h,
                                                         int tiles_w, int ci_tiles, int pad_top, int pad_left) {
    using Cfg = WgCfg<STRIDE, TD, TH, KD, KH, KW, DIL>;
    constexpr int ID = Cfg::ID, IH = Cfg::IH, IW = Cfg::IW, XSTR = Cfg::XSTR, NV = Cfg::NV, GSTR = Cfg::GSTR,
                  NTAPS = Cfg::NTAPS, MAXNT = Cfg::MAXNT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                     // [32 ci][XSTR]
    float* Gs = smem + CT * XSTR;         // [32 co][GSTR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ci0 = (blockIdx.y % ci_tiles) * CT, co0 = (blockIdx.y / ci_tiles) * CT;
    const int tgi = wave % Cfg::TG, kgi = wave / Cfg::TG;      // tap group, k-step group of this wave
    const int t0 = tgi * MAXNT;                                // this wave's taps [t0, t0+nt)
    constexpr int NACC = WINO ? 4 * KD : MAXNT;                // Winograd: frequencies j = 0..3 x depth taps, wave = frequency row
    static_assert(!WINO || (STRIDE == 1 && KH == 3 && KW == 3 && DIL == 1 && TH % 2 == 0), "Winograd form: 3x3, stride 1");
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const size_t HWi = (size_t)H * W, DHWi = (size_t)D * HWi;
    const size_t HWo = (size_t)Ho * Wo, DHWo = (size_t)Do * HWo;
    const int ntiles = B * tiles_d * tiles_h * tiles_w;       // < 2^31 (checked by the host)

    // Register-pipelined staging through buffer descriptors (see conv3d.hip): the loads of the NEXT tile are issued
    // inside this tile's MFMA loop; positions outside the volume carry offset 0x80000000 and read back as 0.
    constexpr int NPOSX = ID * IH * IW;
    constexpr int PPX = (NPOSX + 255) / 256;                 // x positions per thread per channel
    constexpr int GPL = 256 / NV;                            // gy channels covered by one 256-thread pass
    static_assert(256 % NV == 0 && CT % GPL == 0, "gy tile must divide the workgroup");
    constexpr int NGRP = CT / GPL;
    constexpr int NLOADS = CT * PPX + NGRP;
    float xr[CT * PPX], gr[NGRP];
    const unsigned xplane = (unsigned)DHWi * 4u, gplane = (unsigned)DHWo * 4u;
    const int nci = Ci - ci0 < CT ? Ci - ci0 : CT, nco = Co - co0 < CT ? Co - co0 : CT;
    unsigned xoff[PPX], goff;
    auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0, 0x00020000);
    auto grs = xrs;
    // Tile-invariant part of the staging map: this thread's PPX halo positions (dz,hy,xx) and their linear offset.
    int pdz[PPX], phy[PPX], pxx[PPX], prel[PPX];
#pragma unroll
    for (int j = 0; j < PPX; ++j) {
        int t = tid + j * 256;
        pxx[j] = t % IW; t /= IW;
        phy[j] = t % IH;
        const bool in_tile = tid + j * 256 < NPOSX;
        prel[j] = in_tile ? (t / IH) * (int)HWi + phy[j] * W + pxx[j] : 0;
        pdz[j] = in_tile ? t / IH : -(1 << 20);                      // past the halo tile: never inside the volume
    }
    const int gv = tid % NV, gccl = tid / NV;
    const int gxx = gv % TWV, ghy = (gv / TWV) % TH, gdz = gv / (TWV * TH);
    const int grel = gdz * (int)HWo + ghy * Wo + gxx;
    // Per-tile addressing of the next tile (tile >= ntiles: every offset out of range, the loads return 0 untouched).
    auto prefetch_setup = [&](unsigned tile) {
        const bool live = tile < (unsigned)ntiles;
        unsigned r = live ? tile : 0u;                         // depth fastest: depth neighbours share most of their halo
        const int td = (int)(r % (unsigned)tiles_d); r /= (unsigned)tiles_d;
        const int tw = (int)(r % (unsigned)tiles_w); r /= (unsigned)tiles_w;
        const int th = (int)(r % (unsigned)tiles_h);
        const int b = (int)(r / (unsigned)tiles_h);
        const int od0 = td * TD, oh0 = th * TH, ow0 = tw * TWV;
        const int id0 = od0 * STRIDE - Cfg::PADD, ih0 = oh0 * STRIDE - pad_top, iw0 = ow0 * STRIDE - pad_left;
        const int xbase = id0 * (int)HWi + ih0 * W + iw0;
#pragma unroll
        for (int j = 0; j < PPX; ++j) {
            const bool ok = live && (unsigned)(id0 + pdz[j]) < (unsigned)D && (unsigned)(ih0 + phy[j]) < (unsigned)H &&
                            (unsigned)(iw0 + pxx[j]) < (unsigned)W;
            xoff[j] = ok ? (unsigned)(xbase + prel[j]) * 4u : 0x80000000u;
        }
        // ONE descriptor per tensor and sample; the (uniform) channel offset is added to the per-lane offset.  The
        // out-of-volume marker 0x80000000 stays >= num_records after adding any channel offset (< 2^31 bytes per sample).
        xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((size_t)b * Ci + ci0) * DHWi), 0,
                                                (unsigned)nci * xplane, 0x00020000);
        // gy: thread -> (channel within group = tid / NV, voxel = tid % NV)
        const bool vok = live && od0 + gdz < Do && oh0 + ghy < Ho && ow0 + gxx < Wo;
        goff = vok ? (unsigned)(od0 * (int)HWo + oh0 * Wo + ow0 + grel) * 4u + (unsigned)gccl * gplane : 0x80000000u;
        grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy + ((size_t)b * Co + co0) * DHWo), 0,
                                                (unsigned)nco * gplane, 0x00020000);
    };
    auto prefetch_issue = [&](int i) {          // i is a compile-time constant at every (unrolled) call site
        if (i < CT * PPX) {
            const int cc = i / PPX, j = i % PPX;
            xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff[j] + (unsigned)cc * xplane, 0, 0));
        } else {
            const int g = i - CT * PPX;
            gr[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, goff + (unsigned)(g * GPL) * gplane, 0, 0));
        }
    };
#ifdef WG_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = clock64();
#endif
    // XCD-aware persistent schedule: the workers of one XCD (blockIdx.x % 8) walk one contiguous run of the tiles together
    // (see ecm_xcd_tile), so that halo voxels shared by neighbouring tiles are found in that XCD's L2.
    unsigned tile0 = blockIdx.x, tile_end = (unsigned)ntiles, tile_step = gridDim.x;
    if ((gridDim.x & 7u) == 0u) {
        const unsigned xcd = blockIdx.x & 7u, q = (unsigned)ntiles >> 3, rr = (unsigned)ntiles & 7u;
        const unsigned start = xcd * q + (xcd < rr ? xcd : rr);
        tile0 = start + (blockIdx.x >> 3);
        tile_end = start + q + (xcd < rr ? 1u : 0u);
        tile_step = gridDim.x >> 3;
    }
    prefetch_setup(tile0 < tile_end ? tile0 : (unsigned)ntiles);
#pragma unroll
    for (int i = 0; i < NLOADS; ++i) prefetch_issue(i);
    WG_T(0);
    for (unsigned tile = tile0; tile < tile_end; tile += tile_step) {
        __syncthreads();
        WG_T(1);
        // Unconditional stores (positions past the halo tile land in the per-channel pad slot): a guarded store costs
        // a saveexec/branch pair each, and there are CT*PPX of them on the critical path between two MFMA loops.
#pragma unroll
        for (int j = 0; j < PPX; ++j) {
            const int p = tid + j * 256 < NPOSX ? tid + j * 256 : NPOSX;
#pragma unroll
            for (int cc = 0; cc < CT; ++cc) Xs[cc * XSTR + p] = xr[cc * PPX + j];   // [ci][dz][hy][xx], odd channel stride
        }
#pragma unroll
        for (int g = 0; g < NGRP; ++g) Gs[(g * GPL + tid / NV) * GSTR + tid % NV] = gr[g];
        WG_T(2);
        __syncthreads();
        WG_T(3);
        prefetch_setup(tile + tile_step < tile_end ? tile + tile_step : (unsigned)ntiles);
        WG_T(4);
        if constexpr (WINO) {
            const float* gaw = Gs + l31 * GSTR + half * 2;       // lane half = which of the k-step's two tiles
            const float* xbw = Xs + l31 * XSTR + half * 2;
            switch (wave) {      // wave-uniform: the frequency row is a compile-time constant
                case 0: wg_tile_wino<TD, TH, KD, 0, NLOADS>(gaw, xbw, acc, prefetch_issue); break;
                case 1: wg_tile_wino<TD, TH, KD, 1, NLOADS>(gaw, xbw, acc, prefetch_issue); break;
                case 2: wg_tile_wino<TD, TH, KD, 2, NLOADS>(gaw, xbw, acc, prefetch_issue); break;
                default: wg_tile_wino<TD, TH, KD, 3, NLOADS>(gaw, xbw, acc, prefetch_issue); break;
            }
        } else {
        const float* ga = Gs + l31 * GSTR + half;
        const float* xb = Xs + l31 * XSTR + half * STRIDE;
        switch (wave) {          // wave-uniform: makes every tap offset a compile-time immediate
            case 0: wg_tile<STRIDE, TD, TH, KD, KH, KW, DIL, 0 % Cfg::TG, 0 / Cfg::TG, NLOADS>(ga, xb, acc, prefetch_issue); break;
            case 1: wg_tile<STRIDE, TD, TH, KD, KH, KW, DIL, 1 % Cfg::TG, 1 / Cfg::TG, NLOADS>(ga, xb, acc, prefetch_issue); break;
            case 2: wg_tile<STRIDE, TD, TH, KD, KH, KW, DIL, 2 % Cfg::TG, 2 / Cfg::TG, NLOADS>(ga, xb, acc, prefetch_issue); break;
            default: wg_tile<STRIDE, TD, TH, KD, KH, KW, DIL, 3 % Cfg::TG, 3 / Cfg::TG, NLOADS>(ga, xb, acc, prefetch_issue); break;
        }
        }
        WG_T(5);
    }
#ifdef WG_PROFILE
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) wg_prof[wave * 8 + i] = prof[i];
#endif
    if constexpr (WINO) {
        // partial[blockIdx.x][xi = 4*wave + j][kd][co][ci]   (Winograd domain; wgrad_wino_finish applies G^T . G)
        float* pw = partial + (size_t)blockIdx.x * 16 * KD * Co * Ci;
        const int ci = ci0 + l31;
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            const int j = t / KD, kd = t % KD;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * half;
                if (co < Co && ci < Ci) pw[((size_t)((wave * 4 + j) * KD + kd) * Co + co) * Ci + ci] = acc[t][i];
            }
        }
        return;
    }
    // ---- write this wave's partial: partial[blockIdx.x * KG + k-group][co][ci][tap] -------------------
    float* pp = partial + ((size_t)blockIdx.x * Cfg::KG + kgi) * Co * Ci * NTAPS;
#pragma unroll
    for (int t = 0; t < MAXNT; ++t) {
        const int tap = t0 + t;
        if (tap >= NTAPS) continue;
        const int ci = ci0 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (co < Co && ci < Ci) pp[((size_t)co * Ci + ci) * NTAPS + tap] = acc[t][i];
        }
    }
}

// gw[i] = sum_p partial[p][i] in a FIXED order (deterministic): a workgroup owns 32 outputs, its 8 lane groups each sum
// every 8th partial, and the 8 group sums are added in order.  (One thread per output walking all P partials was
// latency-bound: up to 1024 partials x few thousand outputs for the 2-D layers.)
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ partial, float* __restrict__ gw, int n, int P) {
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

// gU [16][KD][Co][Ci] (summed partials, Winograd domain) -> gw [Co][Ci][KD][3][3] = G^T gU G,
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void wgrad_wino_finish(const float* __restrict__ gU, float* __restrict__ gw, int Co, int Ci, int KD) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Co * Ci * KD) return;
    const int kd = idx % KD, ci = (idx / KD) % Ci, co = idx / (KD * Ci);
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[i][j] = gU[((size_t)((i * 4 + j) * KD + kd) * Co + co) * Ci + ci];
    float t[3][4];                                   // G^T u: rows a = 0..2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        t[1][j] = 0.5f * (u[1][j] - u[2][j]);
        t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
    float* o = gw + (((size_t)co * Ci + ci) * KD + kd) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a * 3 + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        o[a * 3 + 1] = 0.5f * (t[a][1] - t[a][2]);
        o[a * 3 + 2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
    }
}

inline int wgrad_workers(int Ci, int Co, long long ntiles, int occ = 1) {
    const int ytiles = ((Ci + CT - 1) / CT) * ((Co + CT - 1) / CT);
    long long p = 256 * occ / ytiles;           // `occ` persistent workgroups per CU in total (3-D: 1, 2-D: 2)
    if (p < 1) p = 1;
    if (p > ntiles) p = ntiles;
    return (int)p;
}

template <int STRIDE, int TD, int TH, int KD, int KH = 3, int KW = 3, int DIL = 1>
int launch_wgrad(const float* x, const float* gy, float* gw, float* partial, int B, int Ci, int Co, int D, int H, int W,
                 hipStream_t st, int Ho = 0, int Wo = 0, int pad_top = 1, int pad_left = 1) {
    using Cfg = WgCfg<STRIDE, TD, TH, KD, KH, KW, DIL>;
    const int Do = (D - 1) / STRIDE + 1;
    if (Ho <= 0) Ho = (H - 1) / STRIDE + 1;
    if (Wo <= 0) Wo = (W - 1) / STRIDE + 1;
    const int tiles_d = (Do + TD - 1) / TD, tiles_h = (Ho + TH - 1) / TH, tiles_w = (Wo + TWV - 1) / TWV;
    const long long ntiles = (long long)B * tiles_d * tiles_h * tiles_w;
    const int ci_tiles = (Ci + CT - 1) / CT, co_tiles = (Co + CT - 1) / CT;
    const int P = wgrad_workers(Ci, Co, ntiles, KD == 3 ? 1 : 2);
    auto kern = conv3d_wgrad_mfma<STRIDE, TD, TH, KD, KH, KW, DIL>;
    {
        const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(P, ci_tiles * co_tiles), dim3(256), Cfg::LDS_BYTES, st, x, gy, partial, B, Ci, Co, D, H,
                       W, Do, Ho, Wo, tiles_d, tiles_h, tiles_w, ci_tiles, pad_top, pad_left);
    const int n = Co * Ci * Cfg::NTAPS;
    hipLaunchKernelGGL(wgrad_reduce, dim3((n + 31) / 32), dim3(256), 0, st, partial, gw, n, P * Cfg::KG);
    return ECM_LAUNCH_RESULT();
}

template <int TD, int TH, int KD>
int launch_wgrad_wino(const float* x, const float* gy, float* gw, float* scratch, int B, int Ci, int Co, int D, int H, int W,
                      hipStream_t st) {
    using Cfg = WgCfg<1, TD, TH, KD>;
    const int tiles_d = (D + TD - 1) / TD, tiles_h = (H + TH - 1) / TH, tiles_w = (W + TWV - 1) / TWV;
    const long long ntiles = (long long)B * tiles_d * tiles_h * tiles_w;
    const int ci_tiles = (Ci + CT - 1) / CT, co_tiles = (Co + CT - 1) / CT;
    const int P = wgrad_workers(Ci, Co, ntiles, KD == 3 ? 1 : 2);
    auto kern = conv3d_wgrad_mfma<1, TD, TH, KD, 3, 3, 1, true>;
    {
        const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    const int n = 16 * KD * Co * Ci;
    float* partial = scratch;
    float* gU = scratch + (size_t)P * n;
    hipLaunchKernelGGL(kern, dim3(P, ci_tiles * co_tiles), dim3(256), Cfg::LDS_BYTES, st, x, gy, partial, B, Ci, Co, D, H,
                       W, D, H, W, tiles_d, tiles_h, tiles_w, ci_tiles, 1, 1);
    hipLaunchKernelGGL(wgrad_reduce, dim3((n + 31) / 32), dim3(256), 0, st, partial, gU, n, P);
    hipLaunchKernelGGL(wgrad_wino_finish, dim3((Co * Ci * KD + 255) / 256), dim3(256), 0, st, gU, gw, Co, Ci, KD);
    return ECM_LAUNCH_RESULT();
}

inline long long ntiles_wino(int B, int D, int H, int W, int kd) {
    const int TD = kd == 3 ? 2 : 1, TH = kd == 3 ? 8 : 16;
    return (long long)B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TWV - 1) / TWV);
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
    if (ntiles_for(B, D, H, W, stride) >= 0x7fffffffLL) return ECM_EUNSUP;   // 32-bit tile counter
    if (scratch_bytes < ecm_conv3d_wgrad_scratch_bytes(B, Ci, Co, D, H, W, stride)) return ECM_ESCRATCH;
    float* partial = static_cast<float*>(scratch);
    hipStream_t st = ecm_stream(stream);
    if (stride == 1) return launch_wgrad<1, 2, 8, 3>(x, gy, gw, partial, B, Ci, Co, D, H, W, st);
    return launch_wgrad<2, 1, 4, 3>(x, gy, gw, partial, B, Ci, Co, D, H, W, st);
}

// ---- general 2-D weight gradient: KH x KW in {3x3, 3x5, 1x1}, stride 1|2, dilation 1|2|4, explicit padding / output size
namespace {
inline long long ntiles2d_ex(int B, int Ho, int Wo, int stride) {
    const int th = stride == 1 ? 16 : 8;
    return (long long)B * ((Ho + th - 1) / th) * ((Wo + TWV - 1) / TWV);
}
}  // namespace

extern "C" long long ecm_conv2d_wgrad_ex_scratch_bytes(int B, int Ci, int Co, int Ho, int Wo, int kh, int kw, int stride) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || kw <= 0) return 0;
    // every worker writes KG <= 4 partials of [Co][Ci][taps]
    return (long long)wgrad_workers(Ci, Co, ntiles2d_ex(B, Ho, Wo, stride), 2) * 4 * Co * Ci * kh * kw * (long long)sizeof(float);
}

extern "C" int ecm_conv2d_wgrad_ex(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes, int B,
                                   int Ci, int Co, int H, int W, int kh, int kw, int stride, int dil, int pad_top,
                                   int pad_left, int Ho, int Wo, void* stream) {
    ECM_CHECK_ARG(x && gy && gw && scratch && B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0);
    if ((long long)H * W * 4 * 32 >= 0x7fffffffLL || (long long)Ho * Wo * 4 * 32 >= 0x7fffffffLL) return ECM_EUNSUP;
    if (ntiles2d_ex(B, Ho, Wo, stride) >= 0x7fffffffLL) return ECM_EUNSUP;
    if (scratch_bytes < ecm_conv2d_wgrad_ex_scratch_bytes(B, Ci, Co, Ho, Wo, kh, kw, stride)) return ECM_ESCRATCH;
    float* partial = static_cast<float*>(scratch);
    hipStream_t st = ecm_stream(stream);
#define WG2(KH, KW, S, DL, TH) if (kh == KH && kw == KW && stride == S && dil == DL) \
        return launch_wgrad<S, 1, TH, 1, KH, KW, DL>(x, gy, gw, partial, B, Ci, Co, 1, H, W, st, Ho, Wo, pad_top, pad_left)
    WG2(3, 3, 1, 1, 16);
    WG2(3, 3, 1, 2, 16);
    WG2(3, 3, 1, 4, 16);
    WG2(3, 3, 2, 1, 8);
    WG2(3, 5, 1, 1, 16);
    WG2(1, 1, 1, 1, 16);
    WG2(1, 1, 2, 1, 8);
#undef WG2
    return ECM_EUNSUP;
}

// ---- Winograd F(2x2,3x3) weight gradient of the stride-1 3x3x3 Conv3d (kd = 3) / 3x3 Conv2d (kd = 1, D = 1) ------------------
extern "C" long long ecm_conv_wino_wgrad_scratch_bytes(int B, int Ci, int Co, int D, int H, int W, int kd) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0 || (kd != 1 && kd != 3)) return 0;
    const long long P = wgrad_workers(Ci, Co, ntiles_wino(B, D, H, W, kd), kd == 3 ? 1 : 2);
    return (P + 1) * 16LL * kd * Co * Ci * (long long)sizeof(float);
}

extern "C" int ecm_conv_wino_wgrad(const float* x, const float* gy, float* gw, void* scratch, long long scratch_bytes, int B,
                                   int Ci, int Co, int D, int H, int W, int kd, void* stream) {
    ECM_CHECK_ARG(x && gy && gw && scratch && B > 0 && Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0);
    if (kd != 1 && kd != 3) return ECM_EUNSUP;
    if ((long long)D * H * W * 4 * 32 >= 0x7fffffffLL || ntiles_wino(B, D, H, W, kd) >= 0x7fffffffLL) return ECM_EUNSUP;
    if (scratch_bytes < ecm_conv_wino_wgrad_scratch_bytes(B, Ci, Co, D, H, W, kd)) return ECM_ESCRATCH;
    hipStream_t st = ecm_stream(stream);
    if (kd == 3) return launch_wgrad_wino<2, 8, 3>(x, gy, gw, static_cast<float*>(scratch), B, Ci, Co, D, H, W, st);
    return launch_wgrad_wino<1, 16, 1>(x, gy, gw, static_cast<float*>(scratch), B, Ci, Co, D, H, W, st);   // D independent planes
}
