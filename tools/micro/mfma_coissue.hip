// Microbenchmark: how much non-matrix work hides under fp32 MFMAs on gfx950?
//   per iteration: 4 independent v_mfma_f32_32x32x2_f32 (256 matrix-pipe cycles) + NV VALU adds (+ NL ds_read_b32) that
//   do not depend on them.  If the extra work hides, the time stays at the MFMA time; if issue serialises, it grows.
//   Variants: everything in ONE wave per SIMD;  TWO waves per SIMD both doing the mix (twice the work).
//   Result on MI355X (profiles/r02_mfma_coissue.txt): nothing hides within a wave (an MFMA holds the wave's issue for its
//   64 cycles; each VALU op adds ~4.7 cycles, each ds_read_b32 ~12); with two waves per SIMD about half of it hides
//   (~2.5 / ~6 cycles).  Non-matrix instructions therefore cost matrix-pipe time roughly in proportion to their COUNT --
//   which is why the Winograd kernels (48 VALU + 24 LDS + 11 VMEM instructions per 24 MFMAs) top out near 80 % busy.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_coissue.hip -o tools/micro/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL, int MODE>   // MODE 0: mix in every wave; 1: even waves MFMA only, odd waves VALU/LDS only
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = (float)i;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    const float* sp = sm + threadIdx.x;
    const bool do_mfma = MODE == 0 || ((threadIdx.x >> 6) + blockIdx.x) % 2 == 0;   // blocks alternate so each SIMD has one of each
    const bool do_other = MODE == 0 || !do_mfma;
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < NV / 4; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(t * 2 + q) & 7]) : "v"(b));
#pragma unroll
                    for (int q = 0; q < NL / 4; ++q) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)(size_t)0 + threadIdx.x * 4), "n"(q * 1024)); v[(t + q) & 7] += 0.f * r; }
                }
            }
        }
        if (MODE == 1 && do_other) {
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(b));
#pragma unroll
            for (int q = 0; q < NL; ++q) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)(size_t)0 + threadIdx.x * 4), "n"((q & 3) * 1024)); v[q & 7] += 0.f * r; }
        }
    }
    float s = sp[0];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <typename K>
static float run(K kern, int blocks, int iters) {
    float* out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}

#define ROW(NV, NL)                                                                                                        \
    printf("%2d VALU + %2d LDS per 4 MFMAs:  1 wave/SIMD %7.3f ms | 2 waves/SIMD (twice the work) %7.3f ms\n", NV, NL,      \
           run(k<NV, NL, 0>, 256, it), run(k<NV, NL, 0>, 512, it))
int main() {
    const int it = 20000;
    printf("4 MFMAs = 256 matrix-pipe cycles per iteration; %d iterations; MFMA-only time is the first row\n", it);
    ROW(0, 0);
    ROW(8, 0);
    ROW(16, 0);
    ROW(32, 0);
    ROW(48, 0);
    ROW(64, 0);
    ROW(0, 8);
    ROW(0, 16);
    ROW(16, 8);
    ROW(32, 16);
    return 0;
}
