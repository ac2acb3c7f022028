// Input side of the harness: the reference stores a stereo sample as one float32 frame [H,W,7] =
// (left RGB, right RGB, disparity) (flying3ddata.py:34-39: uint8 images and a float32 PFM disparity concatenated, so the
// six colour channels only ever hold the integers 0..255) and turns it into network inputs on the CPU, per sample:
//   Flying3d.__getitem__ (cmf/loader/Flying3d.py:49-99): crop (train: random 256x512 window; eval: rows [0,540) followed
//     by the frame's last 36 rows, 576 in all), /255, HWC->CHW, (x-mean)/std per channel, disparity passed through;
//   KITTI.__getitem__ eval branch (cmf/loader/KITTI.py:98-108): pad to 384x1248 at the TOP and LEFT by repeating the
//     frame's first rows / columns, with the disparity zeroed in the repeated part -- and, because `padding_h` /
//     `padding_w` are numpy VIEWS, also in the first th-h rows / tw-w columns of the frame itself (reproduced).
// Here a batch of resident frames goes to the three NCHW tensors in one pass.  Two frame encodings:
//   * the reference's float32 [B,H,W,7] frames (ecm_frame_prep);
//   * packed shards (ecm_frame_prep_packed): uint8 [B,H,W,6] colour + fp16 or fp32 [B,H,W] disparity -- 8 (or 10) bytes per
//     pixel instead of 28, the form that keeps 8 GPUs fed at the new step time.  uint8 -> float is exact, so the colour
//     outputs are bit-identical to the float32-frame path; an fp16 disparity is the fp32 value rounded to fp16.
// Arithmetic is the loader's, operation for operation in fp32 -- (v/255 - mean)/std with IEEE division.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

constexpr int FP_MAXB = 32;
struct FrameCrops { int y0[FP_MAXB], x0[FP_MAXB]; };
struct FrameNorm { float mean[3], stdv[3]; };

struct SrcF32 {                       // float32 [B,H,W,7]
    const float* frames;
    __device__ __forceinline__ void load(size_t pix, float (&c)[6], float& d) const {
        const float* p = frames + pix * 7;
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = p[k];
        d = p[6];
    }
};
template <bool HALF>
struct SrcPacked {                    // uint8 [B,H,W,6] + fp16|fp32 [B,H,W]
    const unsigned char* rgb;
    const void* disp;
    __device__ __forceinline__ void load(size_t pix, float (&c)[6], float& d) const {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(rgb + pix * 6);      // 6 B per pixel: 2-B aligned
        const unsigned a = p[0], b = p[1], e = p[2];
        c[0] = (float)(a & 0xff); c[1] = (float)(a >> 8); c[2] = (float)(b & 0xff);
        c[3] = (float)(b >> 8);   c[4] = (float)(e & 0xff); c[5] = (float)(e >> 8);
        d = HALF ? __half2float(static_cast<const __half*>(disp)[pix]) : static_cast<const float*>(disp)[pix];
    }
};

// MODE 0: window / split-tail (Flying3d): out row r < split reads frame row y0+r, r >= split reads H-tail+(r-split);
//         columns x0 .. x0+tw.
// MODE 1: top-left repeat padding (KITTI eval): ph = th-H, pw = tw-W; out (r,c) reads frame (r<ph ? r : r-ph,
//         c<pw ? c : c-pw); disparity = 0 where the SOURCE row < ph or the source column < pw.
template <int MODE, class Src>
__global__ __launch_bounds__(256) void frame_prep_kernel(Src src, float* __restrict__ left, float* __restrict__ right,
                                                         float* __restrict__ disp, float* __restrict__ image,
                                                         FrameCrops crops, FrameNorm nrm, int H, int W, int th, int tw,
                                                         int split, int tail, int b0) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    if (x >= tw) return;
    int sy, sx;
    bool zero_disp = false;
    if (MODE == 0) {
        sy = r < split ? crops.y0[bl] + r : H - tail + (r - split);
        sx = crops.x0[bl] + x;
    } else {
        const int ph = th - H, pw = tw - W;
        sy = r < ph ? r : r - ph;
        sx = x < pw ? x : x - pw;
        zero_disp = sy < ph || sx < pw;
    }
    float c[6], d;
    src.load(((size_t)b * H + sy) * W + sx, c, d);
    const size_t plane = (size_t)th * tw, o = (size_t)r * tw + x;
    float* lp = left + (size_t)b * 3 * plane + o;
    float* rp = right + (size_t)b * 3 * plane + o;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lp[(size_t)k * plane] = (c[k] / 255.0f - nrm.mean[k]) / nrm.stdv[k];
        rp[(size_t)k * plane] = (c[3 + k] / 255.0f - nrm.mean[k]) / nrm.stdv[k];
        if (image) image[(size_t)b * 3 * plane + (size_t)k * plane + o] = c[k];      // Flying3d.py:74-75: raw left, CHW
    }
    disp[(size_t)b * plane + o] = zero_disp ? 0.f : d;
}

template <class Src>
int frame_prep_launch(Src src, float* left, float* right, float* disp, float* image, int B, int H, int W, const int* crop_y0,
                      const int* crop_x0, int th, int tw, int split, int tail, const float* mean3, const float* std3,
                      int mode, hipStream_t st) {
    if (!(left && right && disp && mean3 && std3 && B > 0 && H > 0 && W > 0 && th > 0 && tw > 0)) return ECM_EINVAL;
    if (th > 65535) return ECM_EUNSUP;
    FrameNorm nrm;
    for (int c = 0; c < 3; ++c) { nrm.mean[c] = mean3[c]; nrm.stdv[c] = std3[c]; }
    if (mode == 0) {
        if (!(crop_y0 && crop_x0 && split >= 0 && split <= th && tail >= 0 && th - split <= tail && tail <= H)) return ECM_EINVAL;
        for (int b = 0; b < B; ++b)             // crop windows must lie inside the frame (the loader's randint bounds)
            if (!(crop_y0[b] >= 0 && crop_x0[b] >= 0 && crop_y0[b] + split <= H && crop_x0[b] + tw <= W)) return ECM_EINVAL;
    } else if (mode == 1) {
        if (!(th >= H && tw >= W && th - H <= H && tw - W <= W)) return ECM_EINVAL;      // the pad repeats existing rows / columns
    } else {
        return ECM_EINVAL;
    }
    for (int b0 = 0; b0 < B; b0 += FP_MAXB) {
        const int nb = B - b0 < FP_MAXB ? B - b0 : FP_MAXB;
        FrameCrops crops{};
        if (mode == 0)
            for (int i = 0; i < nb; ++i) { crops.y0[i] = crop_y0[b0 + i]; crops.x0[i] = crop_x0[b0 + i]; }
        const dim3 grid((tw + 255) / 256, th, nb), block(256);
        if (mode == 0)
            hipLaunchKernelGGL((frame_prep_kernel<0, Src>), grid, block, 0, st, src, left, right, disp, image, crops, nrm, H, W, th,
                               tw, split, tail, b0);
        else
            hipLaunchKernelGGL((frame_prep_kernel<1, Src>), grid, block, 0, st, src, left, right, disp, image, crops, nrm, H, W, th,
                               tw, split, tail, b0);
    }
    return ECM_LAUNCH_RESULT();
}

}  // namespace

extern "C" int ecm_frame_prep(const float* frames, float* left, float* right, float* disp, float* image, int B, int H, int W,
                              const int* crop_y0, const int* crop_x0, int th, int tw, int split, int tail, const float* mean3,
                              const float* std3, void* stream) {
    ECM_CHECK_ARG(frames);
    return frame_prep_launch(SrcF32{frames}, left, right, disp, image, B, H, W, crop_y0, crop_x0, th, tw, split, tail, mean3,
                             std3, 0, ecm_stream(stream));
}

extern "C" int ecm_frame_prep_kitti_eval(const float* frames, float* left, float* right, float* disp, float* image, int B,
                                         int H, int W, int th, int tw, const float* mean3, const float* std3, void* stream) {
    ECM_CHECK_ARG(frames);
    return frame_prep_launch(SrcF32{frames}, left, right, disp, image, B, H, W, nullptr, nullptr, th, tw, 0, 0, mean3, std3, 1,
                             ecm_stream(stream));
}

extern "C" int ecm_frame_prep_packed(const unsigned char* rgb6, const void* disp_in, int disp_is_half, float* left,
                                     float* right, float* disp, float* image, int B, int H, int W, const int* crop_y0,
                                     const int* crop_x0, int th, int tw, int split, int tail, const float* mean3,
                                     const float* std3, int mode, void* stream) {
    ECM_CHECK_ARG(rgb6 && disp_in);
    hipStream_t st = ecm_stream(stream);
    if (disp_is_half)
        return frame_prep_launch(SrcPacked<true>{rgb6, disp_in}, left, right, disp, image, B, H, W, crop_y0, crop_x0, th, tw,
                                 split, tail, mean3, std3, mode, st);
    return frame_prep_launch(SrcPacked<false>{rgb6, disp_in}, left, right, disp, image, B, H, W, crop_y0, crop_x0, th, tw,
                             split, tail, mean3, std3, mode, st);
}
