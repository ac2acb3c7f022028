// conv2d_mfma instantiations for one (KH, KW, stride, dilation) case -- see conv2d_kernel.h / conv2d.hip
#include "conv2d_kernel.h"
int ecm_c2_k33_s1_d4(ECM_C2_ARGS) {
    return dispatch_c2<3, 3, 1, 4, 16>(x, wp, y, B, Ci, Co, H, W, Ho, Wo, pad_top, pad_left, st);
}
