// What a split-bf16 ("bf16x3") matrix path could reach on gfx950 (numbers in tools/experiments/README.md):
//   x = x1 + x2 + x3 with three bf16 terms (24 mantissa bits), products x1w1, x1w2, x2w1, x1w3, x3w1, x2w2 accumulated in fp32
//   on v_mfma_f32_32x32x16_bf16 -- six instructions of 8 passes per K = 16, against eight v_mfma_f32_32x32x2_f32 of 16 passes.
// Measured here, per wave and with 1 / 2 waves per SIMD:
//   A  the fp32 instruction stream the direct conv kernel issues (register operands only): cycles per K = 16 block
//   B  the six bf16 products into ONE accumulator (dependent chain) and into two interleaved accumulators
//   C  B with the operand traffic of the best tiling found on paper: per (tap, 16-channel block) 27 ds_read_b128 per 72 MFMAs
//   D  the split itself: fp32 -> three bf16 terms, VALU instructions per element
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/mfma_bf16x3.hip -o tools/micro/mfma_bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fp32(float* out, int iters, float a0, float b0) {
    f32x16 acc0 = {}, acc1 = {};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {                        // K = 16: eight 32x32x2 (four per accumulator)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3];
}

template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters, float seed) {
    f32x16 acc[2] = {};
    bf16x8 a[3], b[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[t][e] = (__bf16)(seed + t + e + threadIdx.x); b[t][e] = (__bf16)(seed - t - e); }
    for (int i = 0; i < iters; ++i) {
        // a1b1, a1b2, a2b1, a1b3, a3b1, a2b2 : one K = 16 block per accumulator in turn
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x16& c = acc[NACC == 1 ? 0 : r];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][5];
}

// C: per (tap, 16-channel block): A terms for 3 kh (9 x b128), B terms of 6 input rows (18 x b128), 4 output rows x 3 kh x 6 products
__global__ __launch_bounds__(256) void k_bf16_lds(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 12288; i += 256) smem[i] = seed + i;
    __syncthreads();
    f32x16 acc[4] = {};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* base = reinterpret_cast<const f32x4*>(smem) + wave * 64 + lane;
    for (int i = 0; i < iters; ++i) {
        bf16x8 A[3][3], B[6][3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int t = 0; t < 3; ++t) { f32x4 v = base[((kh * 3 + t) * 256 + (i & 1) * 64) % 3000]; A[kh][t] = __builtin_bit_cast(bf16x8, v); }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int t = 0; t < 3; ++t) { f32x4 v = base[((9 + r * 3 + t) * 256 + (i & 1) * 64) % 3000]; B[r][t] = __builtin_bit_cast(bf16x8, v); }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x16& c = acc[r];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kh][0], B[r + kh][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kh][0], B[r + kh][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kh][1], B[r + kh][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kh][0], B[r + kh][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kh][2], B[r + kh][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kh][1], B[r + kh][1], c, 0, 0, 0);
            }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][5] + acc[2][7] + acc[3][9];
}

// D: the split of 8 fp32 values per thread per iteration into three bf16 terms (what staging would do per element)
__global__ __launch_bounds__(256) void k_split(const float* in, unsigned* out, int iters) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = in[threadIdx.x * 8 + e];
    unsigned accu = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 h = (__bf16)v[e];
            const float r1 = v[e] - (float)h;
            const __bf16 m = (__bf16)r1;
            const float r2 = r1 - (float)m;
            const __bf16 l = (__bf16)r2;
            accu += (unsigned)__builtin_bit_cast(unsigned short, h) + ((unsigned)__builtin_bit_cast(unsigned short, m) << 3) + ((unsigned)__builtin_bit_cast(unsigned short, l) << 7);
            v[e] = v[e] * 1.0001f + 0.5f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = accu;
}

template <class F>
static float time_ms(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    float* out; (void)hipMalloc(&out, 1 << 24);
    float* in; (void)hipMalloc(&in, 1 << 16); (void)hipMemset(in, 0x3f, 1 << 16);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void*)k_bf16_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    printf("%d CUs, nominal %.2f GHz; %d iterations per wave\n", cus, khz / 1e6, iters);
    for (int wps = 1; wps <= 2; ++wps) {                     // waves per SIMD: one or two workgroups of 4 waves per CU
        const int grid = cus * wps;
        const float f32 = time_ms([&] { hipLaunchKernelGGL(k_fp32, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        const float b1 = time_ms([&] { hipLaunchKernelGGL(k_bf16<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); });
        const float b2 = time_ms([&] { hipLaunchKernelGGL(k_bf16<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); });
        const float bl = time_ms([&] { hipLaunchKernelGGL(k_bf16_lds, dim3(grid), dim3(256), 49152, 0, out, iters / 6, 1.f); });
        // per wave and iteration: fp32 8 MFMAs = one K=16 block of a 32x32 tile; bf16 12 MFMAs = two K=16 blocks; lds variant 72 MFMAs = 12 blocks
        const double blk = (double)iters * grid * 4;         // K=16 blocks issued by the fp32 kernel (per launch)
        const double flop = 2.0 * 32 * 32 * 16;
        printf("%d wave(s)/SIMD: fp32 32x32x2   %.3f ms = %6.1f TFLOP/s\n", wps, f32, blk * flop / f32 / 1e9);
        printf("               bf16x3, one accumulator chain      %.3f ms = %6.1f TFLOP/s fp32-equivalent (x%.2f)\n", b1, 2 * blk * flop / b1 / 1e9, 2 * f32 / b1);
        printf("               bf16x3, two interleaved chains     %.3f ms = %6.1f TFLOP/s fp32-equivalent (x%.2f)\n", b2, 2 * blk * flop / b2 / 1e9, 2 * f32 / b2);
        printf("               bf16x3 + 27 ds_read_b128 per 72 MFMAs %.3f ms = %6.1f TFLOP/s fp32-equivalent (x%.2f)\n", bl,
               12.0 * (iters / 6) * grid * 4 * flop / bl / 1e9, (12.0 * (iters / 6) / (double)iters) * f32 / bl);
    }
    const float sp = time_ms([&] { hipLaunchKernelGGL(k_split, dim3(cus * 2), dim3(256), 0, 0, in, (unsigned*)out, 4000); });
    printf("split fp32 -> 3 x bf16: %.3f ms for %d elements per thread = %.1f ns per 1000 elements per CU\n", sp, 8 * 4000, sp * 1e6 / (8.0 * 4000 * 512 / 1000) );
    return 0;
}
