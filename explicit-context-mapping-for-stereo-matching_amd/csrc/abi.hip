#include "common.h"
extern "C" int ecm_abi_version(void) { return 4; }
extern "C" const char* ecm_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case ECM_EINVAL: return "ecm: invalid argument (null pointer or non-positive shape)";
        case ECM_EUNSUP: return "ecm: shape not supported by the gfx950 kernels";
        case ECM_ESCRATCH: return "ecm: scratch buffer too small";
        case ECM_EASYNC: return "ecm: an earlier GroupNorm cluster launch timed out on the device (shared / partitioned GPU?); "
                                "its outputs are NaN -- see ecm_async_status / ecm_gn3d_cluster_mode";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ecm: unknown error";
    }
}
