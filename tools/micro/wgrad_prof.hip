// Phase-level cycle profile of the wgrad kernel (workgroup 0): build with
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DWG_PROFILE -Iinclude -Iexplicit-context-mapping-for-stereo-matching_amd/csrc tools/micro/wgrad_prof.hip -o tools/micro/wgrad_prof
#include "conv3d_wgrad.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    // 1 | 2: direct 3-D weight gradient with that stride;  3: Winograd form of the stride-1 3-D layer;
    // 0: Winograd 2-D (encoder) variant on 8 x 32 x 576 x 960
    const int mode = argc > 1 ? atoi(argv[1]) : 3;
    const int stride = mode == 2 ? 2 : 1;
    const bool two_d = mode == 0, wino = mode == 0 || mode == 3;
    const int B = two_d ? 8 : 1, Ci = 32, Co = 32, D = two_d ? 1 : 48, H = two_d ? 576 : 144, W = two_d ? 960 : 240;
    const int st_ = two_d ? 1 : stride;
    const int Do = (D - 1) / st_ + 1, Ho = (H - 1) / st_ + 1, Wo = (W - 1) / st_ + 1;
    const size_t nx = (size_t)B * Ci * D * H * W, ng = (size_t)B * Co * Do * Ho * Wo;
    float *x, *g, *gw; void* scratch;
    hipMalloc(&x, nx * 4); hipMalloc(&g, ng * 4); hipMalloc(&gw, (size_t)Co * Ci * 27 * 4);
    hipMemset(x, 0, nx * 4); hipMemset(g, 0, ng * 4);
    const long long sb = wino ? ecm_conv_wino_wgrad_scratch_bytes(B, Ci, Co, D, H, W, two_d ? 1 : 3)
                              : ecm_conv3d_wgrad_scratch_bytes(B, Ci, Co, D, H, W, stride);
    hipMalloc(&scratch, sb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        int rc = wino ? ecm_conv_wino_wgrad(x, g, gw, scratch, sb, B, Ci, Co, D, H, W, two_d ? 1 : 3, nullptr)
                      : ecm_conv3d_k3_wgrad(x, g, gw, scratch, sb, B, Ci, Co, D, H, W, stride, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rc=%d  %.3f ms\n", rc, ms);
    }
    unsigned long long prof[32];
    hipMemcpyFromSymbol(prof, HIP_SYMBOL(wg_prof), sizeof(prof));
    const char* names[6] = {"first prefetch", "barrier A (wait others)", "LDS stores (+vmcnt)", "barrier B", "prefetch issue", "MFMA loop"};
    for (int w = 0; w < 4; ++w) {
        unsigned long long tot = 0; for (int i = 0; i < 6; ++i) tot += prof[w * 8 + i];
        printf("wave %d total %llu cycles\n", w, tot);
        for (int i = 0; i < 6; ++i) printf("   %-26s %10llu  %5.1f%%\n", names[i], prof[w * 8 + i], 100.0 * prof[w * 8 + i] / tot);
    }
    return 0;
}
