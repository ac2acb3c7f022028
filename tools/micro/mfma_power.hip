// Microbenchmark: does the sustained fp32 MFMA rate on gfx950 depend on the DATA?  tools/micro/mfma_peak.hip feeds every MFMA the
// constants (1.0, 0.5) and holds 2.37 GHz / 155.6 TFLOP/s for 345 ms; the convolution kernels of this library measure 1.9-2.0 GHz
// under the counters.  Here the same loop runs on (a) constant operands, (b) per-lane pseudo-random operands that differ from
// one MFMA to the next, (c) random operands with LDS reads and VALU work between the MFMAs (the mix of a Winograd main loop),
// each for ~0.3 s so that power management settles.  Prints TFLOP/s and the shader clock (clock64 ticks / wall time).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o tools/micro/mfma_power   Run: ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned long long clk[2];

__device__ __forceinline__ float rnd(unsigned x) {              // [-1, 1), per (lane, slot)
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ float lds[256 * 9];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    const int tid = threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = MODE == 0 ? 1.0f : rnd(tid * 16 + u + blockIdx.x * 4096);
        b[u] = MODE == 0 ? 0.5f : rnd(tid * 16 + 8 + u + blockIdx.x * 4096);
    }
    for (int i = 0; i < 9; ++i) lds[tid * 9 + i] = rnd(tid * 9 + i + 77);
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float side = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + t) & 7], acc[t], 0, 0, 0);
            if (MODE == 2) {                                     // per 4 MFMAs: 4 LDS reads, 8 VALU (the Winograd loop's mix)
                const float l0 = lds[((tid + it) & 255) * 9 + u], l1 = lds[((tid + 64 + it) & 255) * 9 + ((u + 1) & 7)];
                const float l2 = lds[((tid + 128 + it) & 255) * 9 + ((u + 2) & 7)], l3 = lds[((tid + 192 + it) & 255) * 9 + ((u + 3) & 7)];
                side = side * 0.5f + (l0 - l1) * (l2 + l3);
                a[u] = a[u] * 0.999f + side * 1e-6f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = side;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.f) out[tid] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <typename K>
static void run(const char* name, K kern, int iters) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 512.0 * 4 * iters * 8.0 * 4 * 4096 / (ms * 1e-3) / 1e12;
    unsigned long long c[2] = {0, 0};
    hipMemcpyFromSymbol(c, HIP_SYMBOL(clk), sizeof(c));
    printf("%-58s %8.3f ms  %6.1f TFLOP/s   shader clock %.3f GHz\n", name, ms, tf, c[0] / (ms * 1e6));
    hipFree(out);
}

int main() {
    const int it = 200000;                                       // ~0.35 s at full rate, two waves per SIMD
    run("constant operands (1.0, 0.5)", k<0>, it);
    run("random operands, different for every MFMA", k<1>, it);
    run("random operands + 4 LDS reads + 8 VALU per 4 MFMAs", k<2>, it);
    run("constant operands again", k<0>, it);
    return 0;
}
