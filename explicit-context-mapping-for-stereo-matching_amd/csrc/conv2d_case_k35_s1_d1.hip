// conv2d_mfma instantiations for one (KH, KW, stride, dilation) case -- see conv2d_kernel.h / conv2d.hip
#include "conv2d_kernel.h"
int ecm_c2_k35_s1_d1(ECM_C2_ARGS) {
    return dispatch_c2<3, 5, 1, 1, 10>(x, wp, y, B, Ci, Co, H, W, Ho, Wo, pad_top, pad_left, st);
}
