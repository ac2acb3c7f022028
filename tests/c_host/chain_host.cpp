// A torch-free host driving a CHAIN of entry points with packing, scratch and asynchronous-error contracts, the way a
// C/C++ maintainer would bind the library:
//   ecm_conv_wino_pack_weight -> ecm_conv_wino_fwd -> ecm_gn3d_fwd (+ReLU) -> ecm_conv3d_c1_fwd -> ecm_softargmin_heads_fwd
//   ecm_weights9_fwd ------------------------------------------------------------------------------> ecm_aggregate9_fwd
// = classif1 of cmfsm + soft-argmin + ECM weights + 9-neighbour aggregation (cmfsm.py:621-634, 703-706, 431-593, 709-723).
// Inputs, reference-layout weights and the expected outputs of the reference's own modules come from
// tests/golden/chain_classif_heads.bin (written by tests/golden/make_golden_chain.py).  Only the HIP runtime and
// include/ecm_hip.h are used.  Returns 0 = pass.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "ecm_hip.h"

namespace {
struct Tensor { std::vector<unsigned> dims; std::vector<float> data; size_t numel() const { return data.size(); } };

bool load_blob(const char* path, std::map<std::string, Tensor>& out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    unsigned count = 0;
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "ECMBLOB1", 8) == 0 && std::fread(&count, 4, 1, f) == 1;
    for (unsigned i = 0; ok && i < count; ++i) {
        unsigned nl = 0, nd = 0;
        ok = std::fread(&nl, 4, 1, f) == 1 && nl < 256;
        std::string name(nl, '\0');
        ok = ok && std::fread(&name[0], 1, nl, f) == nl && std::fread(&nd, 4, 1, f) == 1 && nd <= 8;
        Tensor t;
        t.dims.resize(nd);
        ok = ok && std::fread(t.dims.data(), 4, nd, f) == nd;
        size_t n = 1;
        for (unsigned d : t.dims) n *= d;
        t.data.resize(n);
        ok = ok && std::fread(t.data.data(), 4, n, f) == n;
        out[name] = std::move(t);
    }
    std::fclose(f);
    return ok;
}

struct Dev {                      // device buffers owned by the host, freed at exit
    std::vector<void*> all;
    template <class T> T* alloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n * sizeof(T) + 16) != hipSuccess) return nullptr; all.push_back(p); return (T*)p; }
    float* upload(const Tensor& t) { float* p = alloc<float>(t.numel()); if (p && hipMemcpy(p, t.data.data(), t.numel() * 4, hipMemcpyHostToDevice) != hipSuccess) p = nullptr; return p; }
    ~Dev() { for (void* p : all) (void)hipFree(p); }
};

double max_abs_diff(const float* dptr, const Tensor& want, hipStream_t st) {
    std::vector<float> got(want.numel());
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(got.data(), dptr, got.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return NAN;
    double m = 0.0;
    for (size_t i = 0; i < got.size(); ++i) {
        const double d = std::fabs((double)got[i] - (double)want.data[i]);
        if (!(d <= m)) m = d;                       // NaN propagates
    }
    return m;
}
}  // namespace

#define RC(call) do { int rc_ = (call); if (rc_ != 0) { std::printf("%s -> %d (%s)\n", #call, rc_, ecm_error_string(rc_)); return 1; } } while (0)
#define CHECK(name, dptr, tol) do { const double e_ = max_abs_diff(dptr, T[name], st); std::printf("  %-12s max |diff| %.3e (tol %.1e)\n", name, e_, (double)(tol)); if (!(e_ <= (tol))) ++bad; } while (0)

extern "C" int chain_host_main(const char* blob_path) {
    std::map<std::string, Tensor> T;
    if (!load_blob(blob_path, T)) { std::printf("cannot read %s\n", blob_path); return 2; }
    for (const char* k : {"x", "lr", "hr", "conv_w", "gn_gamma", "gn_beta", "c1_w", "W0", "W1", "W2", "W3", "exp_hidden", "exp_logits",
                          "exp_disp", "exp_w9", "exp_pred"})
        if (!T.count(k)) { std::printf("fixture lacks %s\n", k); return 2; }
    const int B = (int)T["x"].dims[0], C = (int)T["x"].dims[1], D = (int)T["x"].dims[2], h = (int)T["x"].dims[3], w = (int)T["x"].dims[4];
    const int H = (int)T["hr"].dims[2], W = (int)T["hr"].dims[3], s = W / w;
    if (ecm_abi_version() < 3) { std::printf("ABI version %d < 3\n", ecm_abi_version()); return 1; }
    Dev dev;
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return 2;
    float *x = dev.upload(T["x"]), *lr = dev.upload(T["lr"]), *hr = dev.upload(T["hr"]), *cw = dev.upload(T["conv_w"]);
    float *gam = dev.upload(T["gn_gamma"]), *bet = dev.upload(T["gn_beta"]), *c1w = dev.upload(T["c1_w"]);
    float *W0 = dev.upload(T["W0"]), *W1 = dev.upload(T["W1"]), *W2 = dev.upload(T["W2"]), *W3 = dev.upload(T["W3"]);
    if (!(x && lr && hr && cw && gam && bet && c1w && W0 && W1 && W2 && W3)) { std::printf("device allocation / upload failed\n"); return 2; }
    const size_t vol = (size_t)B * C * D * h * w;
    int bad = 0;

    // 1. Conv3d 32 -> 32 (k3, stride 1) on the Winograd kernel: pack the checkpoint-layout weight, then run
    const long long npk = ecm_conv_wino_packed_floats(C, C, 3);
    if (npk <= 0) { std::printf("packed_floats = %lld\n", npk); return 1; }
    float* packed = dev.alloc<float>((size_t)npk);
    float* y = dev.alloc<float>(vol);
    RC(ecm_conv_wino_pack_weight(cw, packed, C, C, 3, 0, st));
    RC(ecm_conv_wino_fwd(x, packed, y, B, C, C, D, h, w, 3, st));
    // 2. GroupNorm(32) + ReLU: caller-provided scratch, sized by the query; an undersized scratch is refused, not overrun
    const long long gnb = ecm_gn3d_scratch_bytes(B, C, (long long)D * h * w);
    void* gscr = dev.alloc<char>((size_t)gnb);
    float* stats = dev.alloc<float>((size_t)B * 32 * 2);
    float* hid = dev.alloc<float>(vol);
    if (gnb > 16 && ecm_gn3d_fwd(y, gam, bet, nullptr, hid, stats, gscr, gnb - 16, B, C, (long long)D * h * w, 1, 1e-5f, st) != ECM_ESCRATCH) {
        std::printf("undersized GroupNorm scratch was not refused\n"); ++bad;
    }
    RC(ecm_gn3d_fwd(y, gam, bet, nullptr, hid, stats, gscr, gnb, B, C, (long long)D * h * w, 1, 1e-5f, st));
    CHECK("exp_hidden", hid, 2e-4);
    // 3. classifier Conv3d 32 -> 1 (reference weight layout, no packing)
    float* logits = dev.alloc<float>((size_t)B * D * h * w);
    RC(ecm_conv3d_c1_fwd(hid, c1w, logits, B, C, D, h, w, st));
    CHECK("exp_logits", logits, 2e-4);
    // 4. soft-argmin over D (one head)
    float* disp = dev.alloc<float>((size_t)B * h * w);
    RC(ecm_softargmin_heads_fwd(logits, (long long)B * D * h * w, disp, 1, B, D, h * w, st));
    CHECK("exp_disp", disp, 1e-3);
    // 5. ECM weights: nine softmax planes per full-resolution pixel
    const long long wb = ecm_weights9_scratch_bytes(B, h, w);
    void* wscr = dev.alloc<char>((size_t)wb);
    float* w9 = dev.alloc<float>((size_t)B * 9 * H * W);
    RC(ecm_weights9_fwd(lr, hr, W0, W1, W2, W3, w9, wscr, wb, B, h, w, s, st));
    CHECK("exp_w9", w9, 2e-5);
    // 6. NN-upsample x s + 9-neighbour aggregation -> full-resolution disparity
    float* pred = dev.alloc<float>((size_t)B * H * W);
    RC(ecm_aggregate9_fwd(disp, w9, pred, 1, B, h, w, s, st));
    CHECK("exp_pred", pred, 2e-2);                       // the stated end-to-end tolerance, in pixels
    // asynchronous device-side failures (GroupNorm cluster time-outs) are reported here, never silently
    if (hipStreamSynchronize(st) != hipSuccess) ++bad;
    if (ecm_async_status(1) != 0) { std::printf("asynchronous error reported\n"); ++bad; }
    // argument errors come back as codes
    if (ecm_conv_wino_fwd(nullptr, packed, y, B, C, C, D, h, w, 3, st) != ECM_EINVAL) ++bad;
    if (ecm_conv_wino_fwd(x, packed, y, B, C, C, D, h, w, 2, st) != ECM_EUNSUP) ++bad;
    (void)hipStreamDestroy(st);
    std::printf("chain_host: %d failures\n", bad);
    return bad ? 1 : 0;
}

#ifdef CHAIN_HOST_PROGRAM
int main(int argc, char** argv) { return chain_host_main(argc > 1 ? argv[1] : "tests/golden/chain_classif_heads.bin"); }
#endif
