// 2-D convolutions of the path as implicit GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32):
//   * the encoder's Conv2d layers (feature_extraction, cmfsm.py:126-236; convbn 36-46): 3x3 with stride 1|2 and dilation
//     1|2|4, the 3-channel stem, the 64/128/320-channel stages, and the 1x1 projections (downsample, SPP branches, lastconv);
//   * the class-indexed convolutions of the collapsed cost volume + dres0.0 (cmfsm.py:667-684; see costvol_conv.hip):
//     P = 3x3, 32 -> 15*32 on the reference features and Q = sheared 3x5, 32 -> 6*32 on the left-padded target features;
//   * every stride-1 data gradient (the same kernel on the flipped / transposed weights).
// Same GEMM view and staging as conv3d.hip (its KD = 1 instantiations remain the 32/64-channel fast path):
//   D[co][pixel] += sum_k A[co][k] B[k][pixel],  k = (tap, ci);  A = weights from LDS (global->LDS DMA, double buffered),
//   B = 32 consecutive x of one row of the staged halo tile (NCHW as it stands is the operand layout), register-pipelined
//   through buffer descriptors whose range check supplies the zero padding.
// New here: output channels beyond one workgroup's COT*32 go to blockIdx.y ("co groups", packed weights grouped to match);
// kernel shape KH x KW, stride, dilation and the (possibly asymmetric) padding are template / run-time parameters; input
// channel counts that are not a multiple of the chunk (the 3-channel stem) read zeros for the missing planes.
#include "conv2d_kernel.h"

extern "C" long long ecm_conv2d_packed_floats_ex(int Ci, int Co, int kh, int kw) {
    if (Ci <= 0 || Co <= 0 || kh <= 0 || kw <= 0) return 0;
    const C2Plan p = c2_plan(Ci, Co, kh, kw);
    return (long long)p.groups * kh * kw * p.cip * p.cot * 32;
}

extern "C" int ecm_conv2d_pack_weight_ex(const float* w, float* packed, int Co, int Ci, int kh, int kw, int flip_transpose,
                                         void* stream) {
    ECM_CHECK_ARG(w && packed && Co > 0 && Ci > 0 && kh > 0 && kw > 0);
    const int Kin = flip_transpose ? Co : Ci, Kout = flip_transpose ? Ci : Co;
    const C2Plan p = c2_plan(Kin, Kout, kh, kw);
    const long long n = (long long)p.groups * kh * kw * p.cip * p.cot * 32;
    hipLaunchKernelGGL(pack_conv2d_weight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ecm_stream(stream), w, packed, Co, Ci,
                       kh * kw, p.cot * 32, p.cip, flip_transpose, n);
    return ECM_LAUNCH_RESULT();
}

extern "C" int ecm_conv2d_fwd_ex(const float* x, const float* wpacked, float* y, int B, int Ci, int Co, int H, int W, int kh,
                                 int kw, int stride, int dil, int pad_top, int pad_left, int Ho, int Wo, void* stream) {
    ECM_CHECK_ARG(x && wpacked && y && B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0);
    hipStream_t st = ecm_stream(stream);
#define C2_CASE(KH, KW, S, D, FN) if (kh == KH && kw == KW && stride == S && dil == D) \
        return FN(x, wpacked, y, B, Ci, Co, H, W, Ho, Wo, pad_top, pad_left, st)
    C2_CASE(3, 3, 1, 1, ecm_c2_k33_s1_d1);
    C2_CASE(3, 3, 1, 2, ecm_c2_k33_s1_d2);
    C2_CASE(3, 3, 1, 4, ecm_c2_k33_s1_d4);
    C2_CASE(3, 3, 2, 1, ecm_c2_k33_s2_d1);
    C2_CASE(3, 5, 1, 1, ecm_c2_k35_s1_d1);
    C2_CASE(1, 1, 1, 1, ecm_c2_k11_s1);
    C2_CASE(1, 1, 2, 1, ecm_c2_k11_s2);
#undef C2_CASE
    return ECM_EUNSUP;
}
