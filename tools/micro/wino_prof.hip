// Phase-level cycle profile of the Winograd conv kernel (workgroup 2000): build with
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DWINO_PROFILE -Iinclude -Iexplicit-context-mapping-for-stereo-matching_amd/csrc tools/micro/wino_prof.hip -o tools/micro/wino_prof
#include "conv_wino.hip"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    const int Ci = argc > 1 ? atoi(argv[1]) : 32, Co = argc > 2 ? atoi(argv[2]) : 32, kd = argc > 3 ? atoi(argv[3]) : 3;
    const int B = kd == 3 ? 4 : 8, D = kd == 3 ? 48 : 1, H = kd == 3 ? 144 : 576, W = kd == 3 ? 240 : 960;
    const size_t nx = (size_t)B * Ci * D * H * W, ny = (size_t)B * Co * D * H * W;
    float *x, *y, *up;
    const long long nu = ecm_conv_wino_packed_floats(Ci, Co, kd);
    if (hipMalloc(&x, nx * 4) != hipSuccess || hipMalloc(&y, ny * 4) != hipSuccess || hipMalloc(&up, nu * 4) != hipSuccess) return 1;
    // random-ish data (zeros raise the clock): fill with a pattern through a host buffer of one plane
    {
        const size_t n1 = (size_t)H * W;
        float* h = (float*)malloc(n1 * 4);
        for (size_t i = 0; i < n1; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
        for (size_t o = 0; o < nx; o += n1) (void)hipMemcpy(x + o, h, n1 * 4, hipMemcpyHostToDevice);
        for (long long o = 0; o < nu; o += (long long)n1) (void)hipMemcpy(up + o, h, (size_t)((nu - o) < (long long)n1 ? (nu - o) : (long long)n1) * 4, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        int rc = ecm_conv_wino_fwd(x, up, y, B, Ci, Co, D, H, W, kd, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * 9 * kd * Ci * Co * (double)D * H * W * B;
        printf("rc=%d  %.3f ms  %.1f TFLOP/s (direct-conv flops)\n", rc, ms, fl / ms / 1e9);
    }
    unsigned long long prof[32];
    (void)hipMemcpyFromSymbol(prof, HIP_SYMBOL(wino_prof), sizeof(prof));
    const char* names[6] = {"setup + prologue", "drain + rendezvous before the epilogue", "main loop (MFMA groups with staging slices, rendezvous)",
                            "epilogue: column transform + exchange writes", "epilogue: offsets, addend loads, rendezvous", "epilogue: exchange reads, row transform, stores"};
    for (int w = 0; w < 4; w += 3) {
        unsigned long long tot = 0; for (int i = 0; i < 6; ++i) tot += prof[w * 8 + i];
        printf("wave %d total %llu cycles\n", w, tot);
        for (int i = 0; i < 6; ++i) printf("   %-30s %10llu  %5.1f%%\n", names[i], prof[w * 8 + i], 100.0 * prof[w * 8 + i] / tot);
    }
    return 0;
}
