// conv2d_mfma instantiations for one (KH, KW, stride, dilation) case -- see conv2d_kernel.h / conv2d.hip
#include "conv2d_kernel.h"
int ecm_c2_k11_s2(ECM_C2_ARGS) {
    return dispatch_c2<1, 1, 2, 1, 22>(x, wp, y, B, Ci, Co, H, W, Ho, Wo, pad_top, pad_left, st);
}
