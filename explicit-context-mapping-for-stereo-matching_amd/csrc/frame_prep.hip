// Input side of the harness: the reference stores a stereo sample as one float32 frame [H,W,7] =
// (left RGB, right RGB, disparity) (flying3ddata.py:34-39) and turns it into network inputs on the CPU, per sample, in
// Flying3d.__getitem__ (cmf/loader/Flying3d.py:49-99): crop (train: random 256x512 window; eval: rows [0,540) followed
// by the frame's last 36 rows, 576 in all), /255, HWC->CHW, (x-mean)/std per channel, disparity plane passed through.
// Here a batch of resident frames goes to the three NCHW tensors in one pass: out row r < split reads frame row y0+r,
// r >= split reads frame row H-tail+(r-split); columns x0 .. x0+tw.  Arithmetic is the loader's, operation for operation
// in fp32 -- (v/255 - mean)/std with IEEE division -- so the result is bit-identical to the numpy/torch path.
#include "common.h"

namespace {

constexpr int FP_MAXB = 32;
struct FrameCrops { int y0[FP_MAXB], x0[FP_MAXB]; };
struct FrameNorm { float mean[3], stdv[3]; };

__global__ __launch_bounds__(256) void frame_prep_kernel(const float* __restrict__ frames, float* __restrict__ left,
                                                         float* __restrict__ right, float* __restrict__ disp,
                                                         float* __restrict__ image, FrameCrops crops, FrameNorm nrm, int H,
                                                         int W, int th, int tw, int split, int tail, int b0) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    if (x >= tw) return;
    const int sy = r < split ? crops.y0[bl] + r : H - tail + (r - split);
    const int sx = crops.x0[bl] + x;
    const float* p = frames + (((size_t)b * H + sy) * W + sx) * 7;
    const size_t plane = (size_t)th * tw, o = (size_t)r * tw + x;
    float* lp = left + (size_t)b * 3 * plane + o;
    float* rp = right + (size_t)b * 3 * plane + o;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lv = p[c], rv = p[3 + c];
        lp[(size_t)c * plane] = (lv / 255.0f - nrm.mean[c]) / nrm.stdv[c];
        rp[(size_t)c * plane] = (rv / 255.0f - nrm.mean[c]) / nrm.stdv[c];
        if (image) image[(size_t)b * 3 * plane + (size_t)c * plane + o] = lv;      // Flying3d.py:74-75: raw left, CHW
    }
    disp[(size_t)b * plane + o] = p[6];
}

}  // namespace

extern "C" int ecm_frame_prep(const float* frames, float* left, float* right, float* disp, float* image, int B, int H, int W,
                              const int* crop_y0, const int* crop_x0, int th, int tw, int split, int tail, const float* mean3,
                              const float* std3, void* stream) {
    ECM_CHECK_ARG(frames && left && right && disp && crop_y0 && crop_x0 && mean3 && std3 && B > 0 && H > 0 && W > 0);
    ECM_CHECK_ARG(th > 0 && tw > 0 && split >= 0 && split <= th && tail >= 0 && th - split <= tail && tail <= H);
    FrameNorm nrm;
    for (int c = 0; c < 3; ++c) { nrm.mean[c] = mean3[c]; nrm.stdv[c] = std3[c]; }
    for (int b = 0; b < B; ++b) {           // crop windows must lie inside the frame (the loader's randint bounds)
        ECM_CHECK_ARG(crop_y0[b] >= 0 && crop_x0[b] >= 0 && crop_y0[b] + split <= H && crop_x0[b] + tw <= W);
    }
    if (th > 65535) return ECM_EUNSUP;
    hipStream_t st = ecm_stream(stream);
    for (int b0 = 0; b0 < B; b0 += FP_MAXB) {
        const int nb = B - b0 < FP_MAXB ? B - b0 : FP_MAXB;
        FrameCrops crops;
        for (int i = 0; i < nb; ++i) { crops.y0[i] = crop_y0[b0 + i]; crops.x0[i] = crop_x0[b0 + i]; }
        hipLaunchKernelGGL(frame_prep_kernel, dim3((tw + 255) / 256, th, nb), dim3(256), 0, st, frames, left, right, disp, image,
                           crops, nrm, H, W, th, tw, split, tail, b0);
    }
    return ECM_LAUNCH_RESULT();
}
