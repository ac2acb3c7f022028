// 3x3 / stride 1 / pad 1 Conv2d (no bias) by Winograd F(4x4, 3x3) on the fp32 matrix cores: the encoder's convbn layers
// (cmfsm.py:36-46, 126-236) and -- with flipped / transposed weights -- their data gradients.  Y = A^T [ sum_ci U (.) V ] A
// with U = G g G^T (6x6 per filter) and V = B^T d B (6x6 per input patch): 36 multiplies per 16 outputs per (ci, co)
// instead of the 64 of F(2x2,3x3) (conv_wino.hip) and the 144 of the direct form -- 1.78x fewer MFMAs than F(2x2).
// fp32 throughout; the transforms multiply by 2, 4, 5, 8 and the packed weights carry 1/4, 1/6, 1/12, 1/24, so results
// differ from the direct kernel by a few ulps more than F(2x2)'s (stated and tested in tests/test_hip_wino44.py).
//
// GEMM per frequency xi in [0,36):  M_xi[co][tile] += sum_ci U_xi[ci][co] * V_xi[ci][tile]   (v_mfma_f32_32x32x2_f32)
//   A = U_xi : lane l holds U[k = l>>5][co = l&31]      LDS image [xi][cc][32 co]   (global -> LDS DMA, double buffered)
//   B = V_xi : lane l holds V[k = l>>5][tile = l&31]    LDS image [xi][cc][32 tiles] (double buffered)
// A workgroup = 8 waves owns 32 consecutive 4x4 tiles (numbered row-major over one plane, wrapping over row ends) x 32
// output channels.  Waves 0-5 are COMPUTE waves: wave i owns the frequency row xi = 6i .. 6i+5 (6 accumulators of 16
// registers); waves 6-7 are STAGING waves: per chunk of 4 input channels each of their 128 threads fetches one 6x6 patch
// (hardware zero padding through a buffer descriptor), transforms it (2 x 6 x 13 VALU operations) and stores the 36
// frequencies; the patch of the chunk after next is in flight meanwhile.  The matrix pipe and the vector ALU are separate
// issue ports, so the transform of chunk c+1 runs under the MFMAs of chunk c on the same SIMDs; one barrier per chunk.
// Epilogue: the column half of A^T M A in the compute waves' registers (a wave owns a frequency row), the row half after an
// exchange through LDS in two passes (output columns 0-1, then 2-3), 8-byte stores; optional addend (see conv_wino.hip).
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int W44_CIC = 4;                                  // input channels per chunk
constexpr int W44_VF = 36 * W44_CIC * 32;                   // floats of one V (or U) image: 18 KB
constexpr int W44_STAGE_FLOATS = 4 * W44_VF;                // V x2, U x2
constexpr int W44_EPI_FLOATS = 6 * 2 * 32 * 32;             // T[6 i][2 b][32 co][32 t] per pass
constexpr int W44_LDS_BYTES = (W44_STAGE_FLOATS > W44_EPI_FLOATS ? W44_STAGE_FLOATS : W44_EPI_FLOATS) * 4;
static_assert(2 * W44_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

// one 6-vector through B^T (input transform; also used along the other axis)
__device__ __forceinline__ void bt6(float d0, float d1, float d2, float d3, float d4, float d5, float (&t)[6]) {
    const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
    const float c = d4 - d2, e = 2.f * (d3 - d1);
    t[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}
// one 6-vector through A^T (output transform)
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float (&y)[4]) {
    const float s0 = m1 + m2, s1 = m1 - m2, s2 = m3 + m4, s3 = m3 - m4;
    y[0] = m0 + s0 + s2;
    y[1] = fmaf(2.f, s3, s1);
    y[2] = fmaf(4.f, s2, s0);
    y[3] = fmaf(8.f, s3, s1) + m5;
}

// packed weights: [co group][chunk][xi 36][cc 4][32 co]   (zero rows / columns beyond Ci / Co)
__global__ __launch_bounds__(512, 2) void conv_wino44_mfma(const float* __restrict__ x, const float* __restrict__ up,
                                                           const float* __restrict__ addend, float* __restrict__ y, int Ci,
                                                           int nchunks, int Co, int P, int H, int W, int tiles_wt, int ntile,
                                                           int tblocks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                       // 2 x [36][4][32]
    float* Us = smem + 2 * W44_VF;          // 2 x [36][4][32]
    int bid = ecm_xcd_tile(blockIdx.x, gridDim.x);
    const int tb = bid % tblocks; bid /= tblocks;
    const int pl = bid % P;
    const int b = bid / P;
    const int grp = blockIdx.y;
    const int n0 = tb * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const size_t HWi = (size_t)H * W, PHW = (size_t)P * HWi;
    const float* xb = x + (size_t)b * Ci * PHW + (size_t)pl * HWi;
    const unsigned plane_bytes = (unsigned)HWi * 4u;
    const float* ug = up + (size_t)grp * nchunks * W44_VF;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    const bool stager = wave >= 6;
    const int st = tid - 384;                                // staging thread id 0..127: (channel-in-chunk, tile)
    const int pc = st >> 5;                                  // wave-uniform per half-wave pair: st>>5 in 0..3
    // ---- staging threads: the 6x6 patch of tile n0 + l31 --------------------------------------------------------------
    unsigned roff[6];
    bool colok[6];
    {
        const int n = n0 + l31;
        const int trow = n / tiles_wt;
        const int oh = 4 * trow, ow = n < ntile ? 4 * (n - trow * tiles_wt) : W + 8;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int gy = oh - 1 + r;
            roff[r] = (unsigned)gy < (unsigned)H && n < ntile ? (unsigned)(gy * W + ow - 1) * 4u : 0x80000000u;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) colok[c] = (unsigned)(ow - 1 + c) < (unsigned)W;
    }
    // The staging waves never touch the accumulators and the compute waves never hold a patch, but one kernel is one register
    // allocation: the patch registers ARE accumulator registers (set 0 = acc[0..2], set 1 = acc[3..5]; 36 of 48 each).
    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
#define W44_RAW(set, k) acc[(set) * 3 + ((k) >> 4)][(k) & 15]
    auto load_raw = [&](int chunk, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int c = chunk * W44_CIC + pc;
        const bool live = c < Ci;                            // channel padding: an empty descriptor reads zeros
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb + (size_t)(live ? c : 0) * PHW), 0,
                                                          live ? plane_bytes : 0u, 0x00020000);
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
                // a column outside the image is an in-range address of the neighbouring row: masked through the offset
                const unsigned off = colok[cc] ? roff[r] + 4u * cc : 0x80000000u;
                W44_RAW(SET, r * 6 + cc) = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            }
    };
    auto transform_store = [&](float* dst, auto set_c) {     // V = B^T d B -> dst[xi][pc][l31]
        constexpr int SET = decltype(set_c)::value;
        float t[36];
#pragma unroll
        for (int c = 0; c < 6; ++c) {                        // B^T d : down the columns
            float o[6];
            bt6(W44_RAW(SET, c), W44_RAW(SET, 6 + c), W44_RAW(SET, 12 + c), W44_RAW(SET, 18 + c), W44_RAW(SET, 24 + c),
                W44_RAW(SET, 30 + c), o);
#pragma unroll
            for (int r = 0; r < 6; ++r) t[r * 6 + c] = o[r];
        }
        float* vp = dst + pc * 32 + l31;
#pragma unroll
        for (int r = 0; r < 6; ++r) {                        // (.) B : along the rows
            float o[6];
            bt6(t[r * 6], t[r * 6 + 1], t[r * 6 + 2], t[r * 6 + 3], t[r * 6 + 4], t[r * 6 + 5], o);
#pragma unroll
            for (int c = 0; c < 6; ++c) vp[(r * 6 + c) * (W44_CIC * 32)] = o[c];
        }
    };
    auto dma_u = [&](int chunk, float* dst) {                // 18 KB = 1152 float4: 9 per staging thread
        const float* src = ug + (size_t)chunk * W44_VF;
#pragma unroll
        for (int i = 0; i < 9; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + (size_t)(st + i * 128) * 4),
                                             (lds_ptr_t)(dst + ((wave - 6) * 64 + i * 128) * 4), 16, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // one chunk: compute waves multiply chunk c (buffer c & 1); staging waves fetch chunk c+2's patch into the register set
    // chunk c held, then transform chunk c+1's patch (fetched a chunk ago, other set) into the other buffer.  The barrier at the
    // end also waits for the staging waves' outstanding loads (its fence drains VMEM), which have had the whole transform to land.
    auto chunk_body = [&](int c, auto set_next, auto set_after) {
        const int buf = c & 1;
        if (stager) {
            if (c + 1 < nchunks) {
                dma_u(c + 1, Us + (buf ^ 1) * W44_VF);
                if (c + 2 < nchunks) load_raw(c + 2, set_after);
                transform_store(Vs + (buf ^ 1) * W44_VF, set_next);
            }
        } else {
            const float* Vc = Vs + buf * W44_VF;
            const float* Uc = Us + buf * W44_VF;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int xi = wave * 6 + j;
#pragma unroll
                for (int kk = 0; kk < W44_CIC / 2; ++kk) {
                    const float a = Uc[(xi * W44_CIC + kk * 2 + half) * 32 + l31];
                    const float bq = Vc[(xi * W44_CIC + kk * 2 + half) * 32 + l31];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    };
    // ---- prologue: chunk 0 staged (set 0), chunk 1's patch in flight (set 1) -----------------------------------------------
    if (stager) {
        load_raw(0, S0{});
        dma_u(0, Us);
        if (nchunks > 1) load_raw(1, S1{});
        transform_store(Vs, S0{});
    }
    __syncthreads();                                         // (drains the DMA of chunk 0's weights as well)
    {
        int c = 0;
        for (; c + 1 < nchunks; c += 2) {
            chunk_body(c, S1{}, S0{});                       // chunk c+1 sits in set 1; chunk c+2 goes to set 0
            chunk_body(c + 1, S0{}, S1{});
        }
        if (c < nchunks) chunk_body(c, S1{}, S0{});
    }
    if (stager) {                                            // the staging waves' "accumulators" held patches: not results
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    }

    // ---- epilogue: Y = A^T M A.  Compute wave i owns frequency ROW i, so T[i][b] = sum_j M[i][j] A[j][b] is formed in
    // registers; T crosses the waves through LDS in two passes (b = 0,1 then b = 2,3): Ts[6 i][2 b][32 co][32 t].
    float* Ts = smem;
    const int nco = Co - grp * 32 < 32 ? Co - grp * 32 : 32;
    const size_t obase = ((size_t)b * Co + (size_t)grp * 32) * PHW + (size_t)pl * HWi;
    // this thread's (co, t) pairs in the output pass: t = l31, co = (tid >> 5) + 16 * e, e = 0, 1
    const int n = n0 + l31, trow = n / tiles_wt;
    const int oh = 4 * trow, ow = 4 * (n - trow * tiles_wt);
    const bool tile_ok = n < ntile;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();                           // pass 0's reads done before pass 1 overwrites
        if (!stager) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = (i & 3) + 8 * (i >> 2) + 4 * half;
                float t4[4];
                at6(acc[0][i], acc[1][i], acc[2][i], acc[3][i], acc[4][i], acc[5][i], t4);
                Ts[((wave * 2 + 0) * 32 + co) * 32 + l31] = t4[pass * 2];
                Ts[((wave * 2 + 1) * 32 + co) * 32 + l31] = t4[pass * 2 + 1];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int co = (tid >> 5) + 16 * e;
            float y0[4], y1[4];
            {
                float m[6][2];
#pragma unroll
                for (int i = 0; i < 6; ++i) { m[i][0] = Ts[((i * 2 + 0) * 32 + co) * 32 + l31]; m[i][1] = Ts[((i * 2 + 1) * 32 + co) * 32 + l31]; }
                at6(m[0][0], m[1][0], m[2][0], m[3][0], m[4][0], m[5][0], y0);
                at6(m[0][1], m[1][1], m[2][1], m[3][1], m[4][1], m[5][1], y1);
            }
            if (!tile_ok || co >= nco) continue;
            const int oc = ow + pass * 2;                    // output columns oc, oc + 1
            float* yp = y + obase + (size_t)co * PHW;
            const float* ap = addend ? addend + obase + (size_t)co * PHW : nullptr;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (oh + a >= H || oc >= W) continue;
                const size_t o = (size_t)(oh + a) * W + oc;
                float v0 = y0[a], v1 = y1[a];
                if (oc + 1 < W) {
                    if (ap) { v0 += ap[o]; v1 += ap[o + 1]; }
                    if ((W & 1) == 0) *reinterpret_cast<float2*>(yp + o) = make_float2(v0, v1);     // oc even, W even: 8-byte aligned
                    else { yp[o] = v0; yp[o + 1] = v1; }
                } else {
                    if (ap) v0 += ap[o];
                    yp[o] = v0;
                }
            }
        }
    }
}

// w [Co][Ci][3][3] (or, flip_transpose: the data-gradient operator w'[ci][co][flipped taps]) -> U = G g G^T (6x6),
// packed [co group][chunk][xi][cc][32]
__global__ void pack_wino44_weight(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int nchunks,
                                   int flip_transpose, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int Kin = flip_transpose ? Co : Ci, Kout = flip_transpose ? Ci : Co;
    const int o = (int)(idx % 32);
    long long r = idx / 32;
    const int cc = (int)(r % W44_CIC); r /= W44_CIC;
    const int xi = (int)(r % 36); r /= 36;
    const int chunk = (int)(r % nchunks);
    const int grp = (int)(r / nchunks);
    const int oc = grp * 32 + o, k = chunk * W44_CIC + cc;
    float v = 0.f;
    if (oc < Kout && k < Kin) {
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) {
                if (!flip_transpose) g[a][bq] = w[((size_t)oc * Ci + k) * 9 + a * 3 + bq];
                else g[a][bq] = w[((size_t)k * Ci + oc) * 9 + (2 - a) * 3 + (2 - bq)];
            }
        // G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]];  U[i][j] = sum G[i][a] g[a][b] G[j][b]
        const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                               {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
        const int i = xi / 6, j = xi % 6;
        double s = 0.0;                                      // the pack runs once per weight: double costs nothing here
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) s += (double)G[i][a] * (double)g[a][bq] * (double)G[j][bq];
        v = (float)s;
    }
    packed[idx] = v;
}

}  // namespace

extern "C" long long ecm_conv_wino44_packed_floats(int Ci, int Co) {
    if (Ci <= 0 || Co <= 0) return 0;
    return (long long)((Co + 31) / 32) * ((Ci + W44_CIC - 1) / W44_CIC) * W44_VF;
}

extern "C" int ecm_conv_wino44_pack_weight(const float* w, float* packed, int Co, int Ci, int flip_transpose, void* stream) {
    ECM_CHECK_ARG(w && packed && Co > 0 && Ci > 0);
    const int Kin = flip_transpose ? Co : Ci, Kout = flip_transpose ? Ci : Co;
    const int nchunks = (Kin + W44_CIC - 1) / W44_CIC;
    const long long n = ecm_conv_wino44_packed_floats(Kin, Kout);
    hipLaunchKernelGGL(pack_wino44_weight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ecm_stream(stream), w, packed, Co, Ci,
                       nchunks, flip_transpose, n);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_conv_wino44_fwd(const float* x, const float* upacked, const float* addend, float* y, int B, int Ci, int Co,
                                   int P, int H, int W, void* stream) {
    ECM_CHECK_ARG(x && upacked && y && B > 0 && Ci > 0 && Co > 0 && P > 0 && H > 0 && W > 0);
    const int tiles_wt = (W + 3) / 4, ntile = ((H + 3) / 4) * tiles_wt;
    const int tblocks = (ntile + 31) / 32;
    const long long nblk = (long long)B * P * tblocks;
    const int groups = (Co + 31) / 32, nchunks = (Ci + W44_CIC - 1) / W44_CIC;
    if (nblk > 0x7fffffffLL || groups > 65535 || (long long)H * W * 4 >= 0x7fffffffLL) return ECM_EUNSUP;
    const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(conv_wino44_mfma), W44_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv_wino44_mfma, dim3((unsigned)nblk, (unsigned)groups), dim3(512), W44_LDS_BYTES, ecm_stream(stream), x,
                       upacked, addend, y, Ci, nchunks, Co, P, H, W, tiles_wt, ntile, tblocks);
    return ECM_LAUNCH_RESULT();
}
