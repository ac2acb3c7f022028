// Microbenchmark: sustained fp32 MFMA rate on gfx950 (calibrates the "peak" used in the rooflines).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak   Run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long clk[2];
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 4; ++i) s += acc[t][i];
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <typename K>
static void run(const char* name, K kern, int blocks, int threads, int iters, double flops_per_wave_iter) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double tf = waves * iters * flops_per_wave_iter / (ms * 1e-3) / 1e12;
    unsigned long long c[2] = {0, 0};
    hipMemcpyFromSymbol(c, HIP_SYMBOL(clk), sizeof(c));
    printf("%-40s blocks=%d thr=%d  %.3f ms  %.1f TFLOP/s   clock64 %.3f GHz  wall_clock64 %.1f MHz\n", name, blocks, threads, ms, tf,
           c[0] / (ms * 1e6), c[1] / (ms * 1e3));
    hipFree(out);
}
int main() {
    const int it = 20000;
    run("32x32x2 4acc 1wave/SIMD", k32<4>, 256, 256, it, 8.0 * 4 * 4096);
    run("32x32x2 7acc 1wave/SIMD", k32<7>, 256, 256, it, 8.0 * 7 * 4096);
    run("32x32x2 4acc 2wave/SIMD", k32<4>, 512, 256, it, 8.0 * 4 * 4096);
    run("32x32x2 1acc 1wave/SIMD (dependent)", k32<1>, 256, 256, it, 8.0 * 1 * 4096);
    run("32x32x2 1acc 2wave/SIMD (dependent)", k32<1>, 512, 256, it, 8.0 * 1 * 4096);
    run("16x16x4 4acc 1wave/SIMD", k16<4>, 256, 256, it, 8.0 * 4 * 2048);
    run("16x16x4 4acc 2wave/SIMD", k16<4>, 512, 256, it, 8.0 * 4 * 2048);
    run("32x32x2 4acc 2wave/SIMD long", k32<4>, 512, 256, it * 10, 8.0 * 4 * 4096);
    return 0;
}
