// A torch-free host of the C ABI: plain HIP runtime + libecm_hip.so, the way a C/C++ maintainer would bind it
// (INTEGRATION.md section 2).  Builds the reference's concat cost volume (cmfsm.py:667-682) on the GPU, checks it
// bit for bit against the loop restated on the host, and round-trips the backward.  Returns 0 = pass.
// Built as a shared object (tests/c_host/Makefile) and entered through `costvol_host_main()` -- from `main()` when
// linked as a program, or in-process via ctypes from the GPU test (a GPU-initialised test process must not exec).
//   hipcc -O2 -fPIC -shared --offload-arch=gfx950 -Iinclude tests/c_host/costvol_host.cpp -L<csrc> -lecm_hip -Wl,-rpath,<csrc> -o libcostvol_host.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ecm_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)

extern "C" int costvol_host_main() {
    const int B = 2, C = 8, h = 6, w = 28, D = 10;
    const size_t nf = (size_t)B * C * h * w, nc = (size_t)B * 2 * C * D * h * w;
    std::vector<float> L(nf), R(nf), cost(nc), ref(nc, 0.f);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : L) v = rnd();
    for (auto& v : R) v = rnd();
    // cmfsm.py:678-681: cost[:, :C, d, :, d:] = L[..., d:];  cost[:, C:, d, :, d:] = R[..., :w-d]
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                for (int y = 0; y < h; ++y)
                    for (int x = d; x < w; ++x) {
                        ref[((((size_t)b * 2 * C + c) * D + d) * h + y) * w + x] = L[(((size_t)b * C + c) * h + y) * w + x];
                        ref[((((size_t)b * 2 * C + C + c) * D + d) * h + y) * w + x] = R[(((size_t)b * C + c) * h + y) * w + x - d];
                    }
    if (ecm_abi_version() <= 0) { std::printf("bad ABI version\n"); return 1; }
    float *dL, *dR, *dC, *dgL, *dgR;
    HIP_OK(hipMalloc(&dL, nf * 4)); HIP_OK(hipMalloc(&dR, nf * 4)); HIP_OK(hipMalloc(&dC, nc * 4));
    HIP_OK(hipMalloc(&dgL, nf * 4)); HIP_OK(hipMalloc(&dgR, nf * 4));
    HIP_OK(hipMemcpy(dL, L.data(), nf * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dR, R.data(), nf * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    int rc = ecm_costvol_concat_fwd(dL, dR, dC, B, C, h, w, D, st);
    if (rc) { std::printf("fwd rc=%d (%s)\n", rc, ecm_error_string(rc)); return 1; }
    // backward of a volume of ones: gL[x] = #{d <= x}, gR[x'] = #{d : x'+d < w}
    std::vector<float> ones(nc, 1.f), gL(nf), gR(nf);
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(cost.data(), dC, nc * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dC, ones.data(), nc * 4, hipMemcpyHostToDevice));
    rc = ecm_costvol_concat_bwd(dC, dgL, dgR, B, C, h, w, D, st);
    if (rc) { std::printf("bwd rc=%d (%s)\n", rc, ecm_error_string(rc)); return 1; }
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(gL.data(), dgL, nf * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gR.data(), dgR, nf * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < nc; ++i) bad += cost[i] != ref[i];
    for (size_t i = 0; i < nf; ++i) {
        const int x = (int)(i % w);
        const float eL = (float)((x < D - 1 ? x : D - 1) + 1), eR = (float)((w - x) < D ? (w - x) : D);
        bad += gL[i] != eL;
        bad += gR[i] != eR;
    }
    // argument errors come back as codes, not crashes
    if (ecm_costvol_concat_fwd(nullptr, dR, dC, B, C, h, w, D, st) != ECM_EINVAL) ++bad;
    std::printf("costvol_host: %zu mismatches\n", bad);
    return bad ? 1 : 0;
}

#ifdef COSTVOL_HOST_PROGRAM
int main() { return costvol_host_main(); }
#endif
