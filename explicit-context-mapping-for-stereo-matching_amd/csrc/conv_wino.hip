// Stride-1 3x3x3 Conv3d / 3x3 Conv2d (pad 1, no bias) by Winograd F(2x2, 3x3) in the (h, w) plane -- direct along depth --
// on the fp32 matrix cores.  Reference ops: nn.Conv3d inside convbn_3d (cmfsm.py:49-58; dres0/1 604-613, hourglass
// 244-259, classif 621-634) and the encoder's 3x3 convbn layers (cmfsm.py:36-46); with flipped / transposed weights the
// same kernel is their data gradient.
//
// For a 2x2 output tile, Y = A^T [ sum_{kd,ci} U[kd,co,ci] (.) V[ci,d+kd-1] ] A with U = G g G^T (4x4 per filter plane) and
// V = B^T x B (4x4 per input patch): 16 multiplies per 4 outputs per (kd,ci,co) instead of 36 -- 2.25x fewer MFMAs than
// the implicit GEMM of conv3d.hip, all in fp32 (v_mfma_f32_32x32x2_f32; the transforms only add / subtract and halve).
//
// GEMM per frequency xi in [0,16):  M_xi[co][p][r][t] += sum_{kd,ci} U_xi[kd][ci][co] * V_xi[ci][p+kd][r][t]
//   A = U_xi : lane l holds U[k = l>>5][co = l&31]            (LDS image [xi][kd][ci][co], global->LDS DMA, double buffered)
//   B = V_xi : lane l holds V[k = l>>5][tile column t = l&31] (LDS image [xi][ci][plane][tile row][32], double buffered)
// One workgroup (4 waves) owns TD planes x 32*TR consecutive 2x2 tiles (numbered row-major over the plane) for 32
// output channels; wave w owns the frequencies 4w..4w+3 (4 x TD*TR accumulators of 16 registers).  Per chunk of CIC input
// channels every thread transforms ONE 4x4 input patch (8 eight-byte buffer loads, contiguous across the lanes, with hardware
// zero padding -> 32 adds -> 16 LDS stores), prefetched TWO chunks ahead through registers while the previous chunks' MFMAs
// run; one barrier per chunk; two workgroups per CU so that one's transform / barrier sits under the other's matrix work.
// The patch loads are inline asm with hand-counted s_waitcnt (the compiler would drain the queue at every transform, and its
// raw_buffer_load_b64 builtin is lowered to a one-dword load): tools/check_wino_isa.py proves on the built code object that
// no register is touched while its load is in flight.  Epilogue: the column half of A^T M A in registers (a wave owns one
// frequency row), the row half after an exchange through LDS.
#include "common.h"
#include <type_traits>

#ifdef WINO_PROFILE
__device__ unsigned long long wino_prof[4 * 8];   // [wave][phase] cycles of one workgroup; debugging aid (tools/micro/wino_prof.hip)
#define WN_T(i) do { const unsigned long long now_ = clock64(); prof[i] += now_ - last; last = now_; } while (0)
#else
#define WN_T(i) do { } while (0)
#endif

#ifndef ECM_WINO_ST_AUX
#define ECM_WINO_ST_AUX 2            // cache policy of the epilogue's stores: nt (1-3 % on the 3-D layers, nothing on the 2-D ones;
                                     // profiles/r04_gn_store_policy.txt, tools/experiments/README.md)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KD, int TD, int TR, int CIC>
struct WinoCfg {
    static constexpr int NP = TD + KD - 1;                 // input planes per tile
    static constexpr int NPR = TD * TR;                    // (plane, tile row) pairs = accumulators per frequency
    static constexpr int V_FLOATS = 16 * CIC * NP * TR * 32;
    static constexpr int U_FLOATS = 16 * KD * CIC * 32;
    static constexpr int STAGE_FLOATS = 2 * (V_FLOATS + U_FLOATS);
    static constexpr int EPI_FLOATS = NPR * 4 * 2 * 32 * 32;         // epilogue exchange: T[q][4 i][32 co][32 t][2 b]
    static constexpr int LDS_FLOATS = STAGE_FLOATS > EPI_FLOATS ? STAGE_FLOATS : EPI_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(CIC * NP * TR * 32 == 256, "one input patch per thread per chunk");
    static_assert(CIC % 2 == 0 && (256 / (NP * TR * 32)) == CIC, "channel of a patch must be wave-uniform");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

// packed weights: [co group][chunk][xi][kd][cc][32 co]  (chunk = CIC input channels; zero rows / columns beyond Ci / Co)
template <int KD, int TD, int TR, int CIC>
__global__ __launch_bounds__(256, 2) void conv_wino_mfma(const float* __restrict__ x, const float* __restrict__ up,
                                                      const float* __restrict__ addend, float* __restrict__ y, int Ci, int nchunks, int Co, int D, int H,
                                                      int W, int tiles_d, int tiles_wt, int ntile, int tblocks) {
    using Cfg = WinoCfg<KD, TD, TR, CIC>;
    constexpr int NP = Cfg::NP, NPR = Cfg::NPR, VF = Cfg::V_FLOATS, UF = Cfg::U_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;                       // 2 x [16][CIC][NP][TR][32]
    float* Us = smem + 2 * VF;              // 2 x [16][KD][CIC][32]

    // The 2x2 output tiles of a plane are numbered row-major (tiles_wt per row, ntile in all) and a workgroup takes 32*TR
    // CONSECUTIVE ones, wrapping over the row ends: no column padding whatever the image width (240 columns = 120 tiles per
    // row: 3.75 blocks of 32 when cut per row, 270 exact blocks when flattened).
    int bid = ecm_xcd_tile(blockIdx.x, gridDim.x);
    const int td = bid % tiles_d; bid /= tiles_d;
    const int tb = bid % tblocks;
    const int b = bid / tblocks;
    const int grp = blockIdx.y;
    const int od0 = td * TD, n0 = tb * (32 * TR);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // ---- this thread's input patch: (channel-in-chunk pc, plane pz, tile row pr, tile column l31) ----------------------
    const int rest = tid >> 5;                               // 0..7
    const int pc = rest / (NP * TR);                         // wave-uniform (static_assert above)
    const int pz = (rest / TR) % NP, pr = rest % TR;
    const size_t HWi = (size_t)H * W, DHWi = (size_t)D * HWi;
    // The 4x4 patch is fetched as 8 pairs of neighbouring columns (8-byte loads, contiguous across the lanes): pair A =
    // columns (ow-1, ow), pair B = (ow+1, ow+2) of rows oh-1..oh+2.  Rows / planes outside the volume get the out-of-range
    // offset (hardware zero fill).  A pair never straddles the end of a row: at the left border pair A is read one column to
    // the right, at the right border pair B one column to the left, and transform_rows moves the values into place.
    unsigned poff[8];
#ifdef ECM_WINO_ALIAS       // timing experiment only (results invalid): every workgroup reads the SAME 64 tiles -> cache-resident input
    const int n_t = (pr * 32 + l31) + 0 * n0;
#else
    const int n_t = n0 + pr * 32 + l31;                      // this thread's tile; past the last one: everything out of range
#endif
    const int trow_t = n_t / tiles_wt;
    const int oh_t = 2 * trow_t, ow_t = n_t < ntile ? 2 * (n_t - trow_t * tiles_wt) : W;   // first output row / column
    const bool edge_l = ow_t == 0, edge_r = ow_t + 2 >= W && ow_t + 1 < W;
    {
#ifdef ECM_WINO_ALIAS
        const int gz = (KD == 3 ? 1 : 0) + pz - KD / 2 + 0 * od0;
#else
        const int gz = od0 - KD / 2 + pz;
#endif
        const int gy0 = oh_t - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gy = gy0 + i;
            const bool ok = (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H;
            const int row = gz * (int)HWi + gy * W;
            poff[i * 2 + 0] = ok && ow_t < W ? (unsigned)(row + ow_t - 1 + (edge_l ? 1 : 0)) * 4u : 0x80000000u;
            poff[i * 2 + 1] = ok && ow_t + 1 < W ? (unsigned)(row + ow_t + 1 - (edge_r ? 1 : 0)) * 4u : 0x80000000u;
        }
    }
    const bool wg_edge = __builtin_amdgcn_ballot_w64(edge_l || edge_r) != 0;   // some tile of this wave touches a row end
#ifdef ECM_WINO_ALIAS
    const float* xb = x;
#else
    const float* xb = x + (size_t)b * Ci * DHWi;
#endif
    const unsigned plane_bytes = (unsigned)DHWi * 4u;
    const int pc_u = __builtin_amdgcn_readfirstlane(pc);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const float* ug = up + (size_t)grp * nchunks * UF;

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    constexpr int NUQ = UF / 4 / 256;                        // weight float4s per thread per chunk
    static_assert(UF % (4 * 256) == 0, "weight chunk moves as whole float4 rounds of the workgroup (hand-counted vmcnt)");

    // Patches are fetched TWO chunks ahead of their transform (two register buffers, used alternately by even / odd chunks):
    // the 8 loads then have a whole chunk's worth of MFMAs to land, so the transform never waits for memory.
    // The loads are issued through inline asm with hand-counted waits: the compiler cannot count VMEM operations around the
    // loop and would drain the queue (s_waitcnt vmcnt(0)) before every transform -- waiting for the patch issued a moment
    // ago instead of the one issued two chunks ago.  (Loads it does not know of only make ITS waits stricter, never unsafe.)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 rawA[8], rawB[8];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_raw = [&](int chunk, f32x2 (&raw)[8]) {
        const int c = chunk * CIC + pc_u;
        const bool live = c < Ci;                            // channel padding: an empty descriptor reads zeros
        const unsigned long long base = reinterpret_cast<unsigned long long>(xb + (size_t)(live ? c : 0) * DHWi);
        u32x4 rsrc;
        rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)base);
        rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
        rsrc.z = live ? plane_bytes : 0u;
        rsrc.w = 0x00020000u;
        // s_nop: the descriptor may have been written by v_readfirstlane a moment ago (VALU-writes-SGPR -> VMEM needs 5 wait
        // states, and the hazard recogniser does not look into asm)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(raw[0]) : "v"(poff[0]), "s"(rsrc) : "memory");
#pragma unroll
        for (int k = 1; k < 8; ++k)
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(raw[k]) : "v"(poff[k]), "s"(rsrc) : "memory");
    };
    // A patch has landed when at most N VMEM operations issued after it are outstanding.  The compiler believes the asm
    // outputs valid from the moment of issue, so every use must be ordered behind the wait by hand: the empty asm statements
    // (volatile, hence after the wait) redefine the registers in place and all uses hang off them.  That it also inserted no
    // copies in between is checked in the disassembly of the built library (tools/check_wino_isa.py, run by tests/test_abi.py).
    auto wait_raw = [&](f32x2 (&raw)[8], auto n) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n)::value) : "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(raw[k]));
    };
    auto dma_u = [&](int chunk, float* dst) {
        const float* src = ug + (size_t)chunk * UF;
#pragma unroll
        for (int i = 0; i < NUQ; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + (size_t)(tid + i * 256) * 4),
                                             (lds_ptr_t)(dst + (wave_u * 64 + i * 256) * 4), 16, 0, 0);
    };
    // V = B^T d B of the patch in `raw` -> Vs image `dst`: 16 frequency planes, this thread's (pc, pz, pr, t) slot.
    // Two halves so that each fits into the shadow of one group of MFMAs in the main loop.
    // The transform in packed fp32 arithmetic (v_pk_add_f32: two adds per instruction): a patch row arrives as two 8-byte
    // pairs, which ARE the even-aligned register pairs the packed instructions want, so the row pass B^T d is 8 packed
    // instructions instead of 16 scalar ones; the column pass (.) B mixes the halves of a pair and is written so that each
    // output pair is one packed add with half-select / negate modifiers.  (Instruction issue next to the MFMAs is what bounds
    // this kernel: DESIGN.md section 3.)
    f32x2 tmp[8];                                            // tmp[2 i], tmp[2 i + 1] = columns (0,1), (2,3) of row i of B^T d
    auto rows_pass = [&](const f32x2 (&r)[8]) {              // rows: B^T d, on column pairs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 d0 = r[h], d1 = r[2 + h], d2 = r[4 + h], d3 = r[6 + h];
            tmp[h] = d0 - d2; tmp[2 + h] = d1 + d2; tmp[4 + h] = d2 - d1; tmp[6 + h] = d1 - d3;
        }
    };
    auto transform_rows = [&](const f32x2 (&rawp)[8]) {
        if (wg_edge) {                                       // pairs read shifted at a row end: move into place, zero the pad
            f32x2 raw[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a0 = rawp[i * 2].x, a1 = rawp[i * 2].y, b0 = rawp[i * 2 + 1].x, b1 = rawp[i * 2 + 1].y;
                raw[i * 2].x = edge_l ? 0.f : a0;
                raw[i * 2].y = edge_l ? a0 : a1;
                raw[i * 2 + 1].x = edge_r ? b1 : b0;
                raw[i * 2 + 1].y = edge_r ? 0.f : b1;
            }
            rows_pass(raw);
        } else {
            rows_pass(rawp);                                 // (straight from the load registers: no copies)
        }
    };
    auto transform_cols_store = [&](float* dst) {
        float* vp = dst + ((pc * NP + pz) * TR + pr) * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                        // columns: (.) B
            const f32x2 A = tmp[i * 2], Bq = tmp[i * 2 + 1];     // (e0, e1), (e2, e3)
            f32x2 o01, o23;
            // (e0 - e2, e1 + e2): low = A.lo - B.lo, high = A.hi + B.lo;   (e2 - e1, e1 - e3): low = -A.hi + B.lo, high = A.hi - B.hi
            // (hipcc lowers the same expressions to scalar adds plus register-pair moves: spelled out)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(o01) : "v"(A), "v"(Bq));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(o23) : "v"(A), "v"(Bq));
            vp[(i * 4 + 0) * (CIC * NP * TR * 32)] = o01.x;
            vp[(i * 4 + 1) * (CIC * NP * TR * 32)] = o01.y;
            vp[(i * 4 + 2) * (CIC * NP * TR * 32)] = o23.x;
            vp[(i * 4 + 3) * (CIC * NP * TR * 32)] = o23.y;
        }
    };

    // No zero fill: the first MFMA into each accumulator (chunk 0, kk = 0, kd = 0) takes a constant-zero C operand -- an inline
    // constant of the instruction -- instead of 16 v_mov per accumulator (128 VALU instructions per wave in the prologue of a
    // kernel whose non-matrix instructions cost matrix-pipe time by their count).
    f32x16 acc[4][NPR];

#ifdef WINO_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = clock64();
#endif
    // ---- prologue: chunk 0 staged, its first two frequencies' operands in registers, chunks 1 / 2 in flight -------------
    // chunk c's patch lives in rawA for even c, rawB for odd c.
    // Patch loads and weight DMAs are issued for every chunk index up to nchunks + 2 -- beyond the last chunk through an
    // empty descriptor / the last chunk's weights again into the buffer nobody reads -- so that every wait below is a
    // constant count on every path.
    float av[4][CIC / 2][KD], bw[4][CIC / 2][NP][TR];
    // MFMA operands of frequency 4*wave + f of the chunk staged in buffer `buf` -> registers
    auto read_ops = [&](int buf, int f) {
        const float* Vc = Vs + buf * VF;
        const float* Uc = Us + buf * UF;
        const int xi = wave * 4 + f;
#pragma unroll
        for (int kk = 0; kk < CIC / 2; ++kk) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < TR; ++r)
                    bw[f][kk][p][r] = Vc[(((xi * CIC + kk * 2 + half) * NP + p) * TR + r) * 32 + l31];
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) av[f][kk][kd] = Uc[((xi * KD + kd) * CIC + kk * 2 + half) * 32 + l31];
        }
    };
    const int last_chunk = nchunks - 1;
    load_raw(0, rawA);
    dma_u(0, Us);
    load_raw(1, rawB);
    wait_raw(rawA, std::integral_constant<int, NUQ + 8>{});   // younger than chunk 0's patch: its weights, chunk 1's patch
    transform_rows(rawA);
    transform_cols_store(Vs);
    dma_u(last_chunk < 1 ? last_chunk : 1, Us + UF);
    load_raw(2, rawA);
    // weights of chunk 0 landed: younger are chunk 1's patch, chunk 1's weights and chunk 2's patch
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 + NUQ) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // raw barrier: see the main loop
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_ops(0, 0);
    read_ops(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    WN_T(0);

    // One chunk = four groups of MFMAs (one per frequency of this wave).  Each group carries a slice of the staging work,
    // small enough to issue in the group's shadow; the operands of a group are read from LDS TWO groups ahead, so neither
    // their latency nor the rendezvous below is on the critical path of the matrix pipe -- a wave keeps its pipe fed even
    // while the other workgroup of the CU is in its prologue / epilogue:
    //   group 0: operands of group 2; transform chunk c+1's patch (loaded two chunks ago) into the other V buffer
    //   group 1: operands of group 3; wait for chunk c+1's weights (DMA issued a chunk ago) and meet the other waves -- the
    //            ONLY rendezvous per chunk, taken while every wave still has half a chunk of MFMAs to issue
    //   group 2: operands of group 0 of chunk c+1; weight DMA of chunk c+2 into the U buffer just vacated
    //   group 3: operands of group 1 of chunk c+1; patch loads of chunk c+3 into the registers just transformed
    // No wave overwrites what another may still read: every read of V/U[c] is issued before the rendezvous of chunk c (and
    // complete at it: lgkmcnt(0)); V[c+2] and U[c+2] are written after it.
    auto chunk_body = [&](int c, f32x2 (&raw_next)[8], auto is_last, auto is_first) {
        constexpr bool LAST = decltype(is_last)::value;         // the odd chunk out at the end: nothing left to stage
        constexpr bool FIRST = decltype(is_first)::value;       // chunk 0: the accumulators start from the constant 0
        const int buf = c & 1;
        const bool more = !LAST && c + 1 < nchunks;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
#pragma unroll
            for (int kk = 0; kk < CIC / 2; ++kk)
#pragma unroll
                for (int kd = 0; kd < KD; ++kd)
#pragma unroll
                    for (int p = 0; p < TD; ++p)
#pragma unroll
                        for (int r = 0; r < TR; ++r) {
                            const f32x16 zero = {};
                            acc[f][p * TR + r] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                av[f][kk][kd], bw[f][kk][p + kd][r], (FIRST && kk == 0 && kd == 0) ? zero : acc[f][p * TR + r], 0, 0, 0);
                        }
#ifdef ECM_WINO_ABLATE_OPREADS
#define read_ops(a, b) do { } while (0)
#endif
            if (f == 0) {
                read_ops(buf, 2);
                // chunk c+1's patch, loaded two chunks ago; younger: weights of c+1, patch of c+2.  Waited for even when it
                // is past the end and unused: until then its registers must not be handed to anything else.
                wait_raw(raw_next, std::integral_constant<int, NUQ + 8>{});
#ifndef ECM_WINO_ABLATE_TRANSFORM
                if (more) {
                    transform_rows(raw_next);
                    transform_cols_store(Vs + (buf ^ 1) * VF);
                }
#endif
            }
            if (f == 1) {
                read_ops(buf, 3);
                if (!LAST) {
                    // chunk c+1's weights landed: the only VMEM operations younger than their DMA are chunk c+2's 8 patch loads.
                    // NOT __syncthreads(): its fence waits vmcnt(0), i.e. for those loads too -- a memory round trip per chunk
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef ECM_WINO_ABLATE_BARRIER     // timing experiments only (results invalid): tools/experiments/README.md, round 4
                    __builtin_amdgcn_s_barrier();
#endif
                    asm volatile("" ::: "memory");
                }
            }
            if (f == 2 && !LAST) {
                if (more) read_ops(buf ^ 1, 0);
                dma_u(c + 2 < nchunks ? c + 2 : last_chunk, Us + buf * UF);
            }
            if (f == 3 && !LAST) {
                if (more) read_ops(buf ^ 1, 1);
                // the registers just transformed get chunk c+3's patch.  (Round 4 tried issuing these loads in group 0, right
                // behind the transform -- a two-chunk window instead of 1.25 chunks: 1-2 % SLOWER on every layer; with the
                // input made cache-resident the kernel gains 3 % (3-D) to 11 % (2-D 32 -> 32), so memory latency is not
                // what it mostly waits for.  tools/experiments/README.md)
                load_raw(c + 3, raw_next);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef ECM_WINO_ABLATE_OPREADS
#undef read_ops
#endif
        WN_T(2);
    };
    {
        int c = 0;
        if (nchunks >= 2) {                                  // the first pair, peeled: chunk 0 starts the accumulators
            chunk_body(0, rawB, std::false_type{}, std::true_type{});
            chunk_body(1, rawA, std::false_type{}, std::false_type{});
            c = 2;
            for (; c + 1 < nchunks; c += 2) {
                chunk_body(c, rawB, std::false_type{}, std::false_type{});          // chunk c even: chunk c+1's patch is in rawB
                chunk_body(c + 1, rawA, std::false_type{}, std::false_type{});
            }
            if (c < nchunks) chunk_body(c, rawB, std::true_type{}, std::false_type{});
        } else {
            chunk_body(0, rawB, std::true_type{}, std::true_type{});              // a single chunk: first and last
        }
    }
    // Before the epilogue reuses the staging buffers: (1) this wave's VMEM queue drained -- the weight DMA issued past the
    // last chunk still WRITES LDS when it lands, and the patch loads issued past the last chunk (empty descriptor) still
    // write their registers, which must stay allocated until then; (2) all operand reads done; (3) every wave has got
    // that far (a DMA of wave B landing after wave A's first exchange write would corrupt it).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(rawA[k]), "v"(rawB[k]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WN_T(1);

    // ---- epilogue: Y = A^T M A.  Wave w owns frequency ROW i = w (xi = 4w + j), so the column half of the transform,
    // T[i][b] = sum_j M[i][j] A[j][b], is done in registers; only T (2 of 4 values) crosses the waves through LDS:
    // Ts[q][4 i][32 co][32 t][2 b] for all (plane, tile row) pairs q at once -- ONE rendezvous -- then
    // Y[a][b] = sum_i A^T[a][i] T[i][b].  The stores go through a buffer descriptor over this workgroup's 32 output channels:
    // positions outside the volume (and channels beyond Co) carry the out-of-range offset and are dropped by the hardware,
    // so the store loop has no branches.
    float* Ts = smem;
    const size_t DHWo = DHWi;                                // stride 1, pad 1: output volume == input volume
    const bool w_even = (W & 1) == 0;
    // exchange image Ts[q][4 i][32 co][32 t][2 b]: the pair (b = 0, 1) of a (co, t) sits in one 8-byte slot, so a lane writes
    // it with ONE ds_write_b64 and the reader fetches a frequency row's pair with one ds_read_b64 -- half the LDS
    // instructions of the [b][co][t] image of rounds 2-3 (the phase profile of round 4 showed the epilogue's two LDS passes
    // at 45-95 cycles per instruction with the CU's other workgroup in its main loop: instruction count, not bytes);
    // 32 consecutive lanes touch 256 consecutive bytes on either side: conflict-free
    typedef float f32x2e __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < NPR; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = (i & 3) + 8 * (i >> 2) + 4 * half;
            const float m0 = acc[0][q][i], m1 = acc[1][q][i], m2 = acc[2][q][i], m3 = acc[3][q][i];
            const f32x2e tb = {m0 + m1 + m2, m1 - m2 - m3};
            *reinterpret_cast<f32x2e*>(Ts + ((((q * 4 + wave) * 32 + co) * 32 + l31) * 2)) = tb;
        }
    WN_T(3);
    const int nco = Co - grp * 32 < 32 ? Co - grp * 32 : 32;
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y + ((size_t)b * Co + (size_t)grp * 32) * DHWo, 0,
                                                       (unsigned)nco * plane_bytes, 0x00020000);
    unsigned yoff[NPR][2];                                   // this thread's tile column t = l31: byte offset of its two rows
#pragma unroll
    for (int q = 0; q < NPR; ++q) {
        const int n = n0 + (q % TR) * 32 + l31, trow = n / tiles_wt;
        const int od = od0 + q / TR, oh = 2 * trow, ow = 2 * (n - trow * tiles_wt);
        const bool ok = n < ntile && od < D;
        const unsigned base = (unsigned)((od * H + oh) * W + ow) * 4u;
        yoff[q][0] = ok ? base : 0x80000000u;
        yoff[q][1] = ok && oh + 1 < H ? base + (unsigned)W * 4u : 0x80000000u;
    }
    const bool has_col1 = w_even;                            // W odd: the second column of the last tile of a row is outside
    // y = conv(x) + addend (optional; same shape as y): fetched here, ahead of the rendezvous, through the same offsets --
    // this is how a data gradient is accumulated onto the gradient arriving over a skip connection without a separate pass
    float ad[NPR][4][4];
    if (addend) {
        const auto ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(addend) + ((size_t)b * Co + (size_t)grp * 32) * DHWo, 0,
                                                           (unsigned)nco * plane_bytes, 0x00020000);
#pragma unroll
        for (int q = 0; q < NPR; ++q)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const unsigned choff = (unsigned)((tid >> 5) + 8 * e4) * plane_bytes;
                const int n = n0 + (q % TR) * 32 + l31;
                const bool two = 2 * (n % tiles_wt) + 1 < W;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    ad[q][e4][r * 2 + 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, yoff[q][r] + choff, 0, 0));
                    ad[q][e4][r * 2 + 1] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(ars, two ? yoff[q][r] + choff + 4u : 0x80000000u, 0, 0));
                }
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own LDS writes done + raw barrier (no need to drain VMEM)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WN_T(4);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < NPR; ++q)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {                     // 1024 (co, t) pairs per q: 4 per thread, t == l31 for all of them
            const int col = (tid >> 5) + 8 * e4;
            f32x2e tv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) tv[i] = *reinterpret_cast<const f32x2e*>(Ts + ((((q * 4 + i) * 32 + col) * 32 + l31) * 2));
            const f32x2e yr0 = tv[0] + tv[1] + tv[2], yr1 = tv[1] - tv[2] - tv[3];        // same order of additions as before
            float y00 = yr0.x, y01 = yr0.y, y10 = yr1.x, y11 = yr1.y;
            if (addend) { y00 += ad[q][e4][0]; y01 += ad[q][e4][1]; y10 += ad[q][e4][2]; y11 += ad[q][e4][3]; }
            const unsigned choff = (unsigned)col * plane_bytes;          // channel beyond Co: >= num_records, dropped
            if (has_col1) {
                u32x2 r0 = {__builtin_bit_cast(unsigned, y00), __builtin_bit_cast(unsigned, y01)};
                u32x2 r1 = {__builtin_bit_cast(unsigned, y10), __builtin_bit_cast(unsigned, y11)};
                __builtin_amdgcn_raw_buffer_store_b64(r0, yrs, yoff[q][0] + choff, 0, ECM_WINO_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(r1, yrs, yoff[q][1] + choff, 0, ECM_WINO_ST_AUX);
            } else {
                const int n = n0 + (q % TR) * 32 + l31;
                const bool two = 2 * (n % tiles_wt) + 1 < W;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), yrs, yoff[q][0] + choff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), yrs, yoff[q][1] + choff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), yrs, two ? yoff[q][0] + choff + 4u : 0x80000000u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), yrs, two ? yoff[q][1] + choff + 4u : 0x80000000u, 0, 0);
            }
        }
#ifdef WINO_PROFILE
    WN_T(5);
    if (blockIdx.x == 2000 && blockIdx.y == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) wino_prof[wave * 8 + i] = prof[i];
#endif
}

// w [Co][Ci][KD][3][3] (or, flip_transpose: the data-gradient operator w'[ci][co][flipped taps]) -> U = G g G^T,
// packed [co group][chunk][xi][kd][cc][32]
// `both`: one launch produces the forward layout (n elements) followed by the data-gradient layout (n2 elements, nchunks2
// chunks) -- a training step needs both and would otherwise pay two dependent 5 us launches per layer
__global__ void pack_wino_weight(const float* __restrict__ w, float* __restrict__ packed, int Co, int Ci, int KD, int CIC,
                                 int nchunks, int flip_transpose, long long n, int both = 0, int nchunks2 = 0, long long n2 = 0) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (both) {
        if (idx >= n + n2) return;
        if (idx >= n) { idx -= n; packed += n; flip_transpose = 1; nchunks = nchunks2; n = n2; }
        else flip_transpose = 0;
    }
    if (idx >= n) return;
    const int Kin = flip_transpose ? Co : Ci, Kout = flip_transpose ? Ci : Co;
    const int o = (int)(idx % 32);
    long long r = idx / 32;
    const int cc = (int)(r % CIC); r /= CIC;
    const int kd = (int)(r % KD); r /= KD;
    const int xi = (int)(r % 16); r /= 16;
    const int chunk = (int)(r % nchunks);
    const int grp = (int)(r / nchunks);
    const int oc = grp * 32 + o, k = chunk * CIC + cc;
    float v = 0.f;
    if (oc < Kout && k < Kin) {
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) {
                if (!flip_transpose) g[a][bq] = w[(((size_t)oc * Ci + k) * KD + kd) * 9 + a * 3 + bq];
                else g[a][bq] = w[(((size_t)k * Ci + oc) * KD + (KD - 1 - kd)) * 9 + (2 - a) * 3 + (2 - bq)];
            }
        // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  U[i][j] = sum_ab G[i][a] g[a][b] G[j][b]
        const int i = xi >> 2, j = xi & 3;
        float row[3];                                        // (G g)[i][:]
#pragma unroll
        for (int bq = 0; bq < 3; ++bq)
            row[bq] = i == 0 ? g[0][bq] : i == 3 ? g[2][bq] : i == 1 ? 0.5f * (g[0][bq] + g[1][bq] + g[2][bq])
                                                                      : 0.5f * (g[0][bq] - g[1][bq] + g[2][bq]);
        v = j == 0 ? row[0] : j == 3 ? row[2] : j == 1 ? 0.5f * (row[0] + row[1] + row[2]) : 0.5f * (row[0] - row[1] + row[2]);
    }
    packed[idx] = v;
}

template <int KD, int TD, int TR, int CIC>
int launch_wino(const float* x, const float* up, const float* addend, float* y, int B, int Ci, int Co, int D, int H, int W, hipStream_t st) {
    using Cfg = WinoCfg<KD, TD, TR, CIC>;
    const int tiles_d = (D + TD - 1) / TD, tiles_wt = (W + 1) / 2, ntile = ((H + 1) / 2) * tiles_wt;
    const int tblocks = (ntile + 32 * TR - 1) / (32 * TR);
    const long long nblk = (long long)B * tiles_d * tblocks;
    const int groups = (Co + 31) / 32, nchunks = (Ci + CIC - 1) / CIC;
    // the epilogue's store / addend descriptors span 32 channels (num_records = 32 planes' bytes, offsets col*plane_bytes + yoff,
    // 0x80000000 as the dropped-store sentinel): all 32-bit arithmetic, so 32 planes must fit below 2^31 bytes -- the same
    // bound as ecm_conv_wino_wgrad
    if (nblk > 0x7fffffffLL || groups > 65535 || (long long)D * H * W * 4 * 32 > 0x80000000LL) return ECM_EUNSUP;
    auto kern = conv_wino_mfma<KD, TD, TR, CIC>;
    const hipError_t e = ecm_allow_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)groups), dim3(256), Cfg::LDS_BYTES, st, x, up, addend, y, Ci, nchunks, Co, D,
                       H, W, tiles_d, tiles_wt, ntile, tblocks);
    return ECM_LAUNCH_RESULT();
}

constexpr int WINO_CIC3 = 2, WINO_CIC2 = 4;

}  // namespace

extern "C" long long ecm_conv_wino_packed_floats(int Ci, int Co, int kd) {
    if (Ci <= 0 || Co <= 0 || (kd != 1 && kd != 3)) return 0;
    const int cic = kd == 3 ? WINO_CIC3 : WINO_CIC2;
    return (long long)((Co + 31) / 32) * ((Ci + cic - 1) / cic) * 16 * kd * cic * 32;
}

extern "C" int ecm_conv_wino_pack_weight(const float* w, float* packed, int Co, int Ci, int kd, int flip_transpose, void* stream) {
    ECM_CHECK_ARG(w && packed && Co > 0 && Ci > 0 && (kd == 1 || kd == 3));
    const int Kin = flip_transpose ? Co : Ci, Kout = flip_transpose ? Ci : Co;
    const int cic = kd == 3 ? WINO_CIC3 : WINO_CIC2;
    const int nchunks = (Kin + cic - 1) / cic;
    const long long n = ecm_conv_wino_packed_floats(Kin, Kout, kd);
    hipLaunchKernelGGL(pack_wino_weight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ecm_stream(stream), w, packed, Co, Ci, kd,
                       cic, nchunks, flip_transpose, n);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_conv_wino_pack_weight2(const float* w, float* packed, int Co, int Ci, int kd, void* stream) {
    ECM_CHECK_ARG(w && packed && Co > 0 && Ci > 0 && (kd == 1 || kd == 3));
    const int cic = kd == 3 ? WINO_CIC3 : WINO_CIC2;
    const long long nf = ecm_conv_wino_packed_floats(Ci, Co, kd), nb = ecm_conv_wino_packed_floats(Co, Ci, kd);
    hipLaunchKernelGGL(pack_wino_weight, dim3((unsigned)((nf + nb + 255) / 256)), dim3(256), 0, ecm_stream(stream), w, packed, Co, Ci,
                       kd, cic, (Ci + cic - 1) / cic, 0, nf, 1, (Co + cic - 1) / cic, nb);
    return ECM_LAUNCH_RESULT();
}

namespace {
int wino_dispatch(const float* x, const float* upacked, const float* addend, float* y, int B, int Ci, int Co, int D, int H, int W,
                  int kd, void* stream) {
    ECM_CHECK_ARG(x && upacked && y && B > 0 && Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0);
    if (W < 2) return ECM_EUNSUP;                            // patches are read as pairs of neighbouring columns
    hipStream_t st = ecm_stream(stream);
    if (kd == 3) return launch_wino<3, 2, 1, WINO_CIC3>(x, upacked, addend, y, B, Ci, Co, D, H, W, st);
    if (kd == 1) return launch_wino<1, 1, 2, WINO_CIC2>(x, upacked, addend, y, B, Ci, Co, D, H, W, st);   // D independent planes
    return ECM_EUNSUP;
}
}  // namespace

extern "C" int ecm_conv_wino_fwd(const float* x, const float* upacked, float* y, int B, int Ci, int Co, int D, int H, int W,
                                 int kd, void* stream) {
    return wino_dispatch(x, upacked, nullptr, y, B, Ci, Co, D, H, W, kd, stream);
}

extern "C" int ecm_conv_wino_fwd_add(const float* x, const float* upacked, const float* addend, float* y, int B, int Ci, int Co,
                                     int D, int H, int W, int kd, void* stream) {
    ECM_CHECK_ARG(addend);
    return wino_dispatch(x, upacked, addend, y, B, Ci, Co, D, H, W, kd, stream);
}
