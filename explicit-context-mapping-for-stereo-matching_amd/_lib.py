"""ctypes binding of libecm_hip.so (the C ABI declared in include/ecm_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is
raised.  Prototypes below mirror include/ecm_hip.h one to one (checked by tests/test_abi.py).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ECM_HIP_LIB") or os.path.join(_HERE, "csrc", "libecm_hip.so")    # ECM_HIP_LIB: an experiment build

_P, _I, _LL, _F = C.c_void_p, C.c_int, C.c_longlong, C.c_float

# name -> (restype, argtypes); the single source of truth for the Python side of the ABI
PROTOTYPES = {
    "ecm_abi_version": (_I, []),
    "ecm_error_string": (C.c_char_p, [_I]),
    "ecm_costvol_concat_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_costvol_concat_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_softargmin_heads_fwd": (_I, [_P, _LL, _P, _I, _I, _I, _I, _P]),
    "ecm_softargmin_heads_bwd": (_I, [_P, _LL, _P, _P, _I, _I, _I, _I, _P]),
    "ecm_disparity_regression_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "ecm_aggregate9_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_aggregate9_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_weights9_scratch_bytes": (_LL, [_I, _I, _I]),
    "ecm_weights9_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _P]),
    "ecm_context_weights_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _P]),
    "ecm_context_weights_bwd_scratch_bytes": (_LL, [_I, _I, _I, _I, _I]),
    "ecm_context_weights_bwd": (_I, [_P] * 11 + [_P, _LL, _I, _I, _I, _I, _I, _P]),
    "ecm_volume_mapping_fwd": (_I, [_P, _LL, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_trilinear_softargmin_fwd": (_I, [_P, _LL, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_volume_mapping_bwd_scratch_bytes": (_LL, [_I] * 6),
    "ecm_volume_mapping_bwd": (_I, [_P, _LL, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_trilinear_softargmin_bwd_scratch_bytes": (_LL, [_I] * 7),
    "ecm_trilinear_softargmin_bwd": (_I, [_P, _LL, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_weights9_bwd_scratch_bytes": (_LL, [_I, _I, _I, _I]),
    "ecm_weights9_bwd": (_I, [_P] * 11 + [_P, _LL, _I, _I, _I, _I, _P]),
    "ecm_conv3d_packed_floats": (_LL, [_I, _I]),
    "ecm_conv3d_pack_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "ecm_conv3d_k3_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_conv_wino_packed_floats": (_LL, [_I, _I, _I]),
    "ecm_conv_wino_pack_weight": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ecm_conv_wino_pack_weight2": (_I, [_P, _P, _I, _I, _I, _P]),
    "ecm_conv_wino_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_conv_wino_fwd_add": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_sum_n": (_I, [_P, _P, _P, _P, _P, _LL, _P]),
    "ecm_zero_insert2d": (_I, [_P, _P, _LL, _I, _I, _I, _I, _P]),
    "ecm_costvol_class_weights_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "ecm_costvol_class_weights_bwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "ecm_conv_wino_wgrad_scratch_bytes": (_LL, [_I] * 7),
    "ecm_conv_wino_wgrad": (_I, [_P, _P, _P, _P, _LL] + [_I] * 7 + [_P]),
    "ecm_conv3d_c1_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_conv3d_c1_dgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_conv3d_c1_wgrad_scratch_bytes": (_LL, [_I, _I, _I, _I, _I]),
    "ecm_conv3d_c1_wgrad": (_I, [_P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _P]),
    "ecm_conv3d_c1_gn_fwd": (_I, [_P] * 6 + [_I, _I, _I, _I, _I, _P]),
    "ecm_conv3d_c1_gn_wgrad": (_I, [_P] * 7 + [_LL, _I, _I, _I, _I, _I, _P]),
    "ecm_deconv3d_pack_weight": (_I, [_P, _P, _I, _I, _P]),
    "ecm_deconv3d_k3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_conv3d_wgrad_scratch_bytes": (_LL, [_I, _I, _I, _I, _I, _I, _I]),
    "ecm_conv3d_k3_wgrad": (_I, [_P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ecm_conv2d_packed_floats_ex": (_LL, [_I] * 4),
    "ecm_conv2d_pack_weight_ex": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_conv2d_fwd_ex": (_I, [_P, _P, _P] + [_I] * 13 + [_P]),
    "ecm_conv2d_wgrad_ex_scratch_bytes": (_LL, [_I] * 8),
    "ecm_conv2d_wgrad_ex": (_I, [_P, _P, _P, _P, _LL] + [_I] * 13 + [_P]),
    "ecm_deconv2d_pack_weight": (_I, [_P, _P, _I, _I, _P]),
    "ecm_deconv2d_k3s2_fwd": (_I, [_P, _P, _P] + [_I] * 7 + [_P]),
    "ecm_stereo_loss_scratch_bytes": (_LL, [_LL]),
    "ecm_stereo_loss_fwd": (_I, [_P] * 6 + [_LL, _LL, _F, _F, _F, _F, _P]),
    "ecm_stereo_loss_bwd": (_I, [_P] * 9 + [_LL, _F, _F, _F, _F, _P]),
    "ecm_costvol_conv_assemble_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_costvol_conv_assemble_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ecm_frame_prep": (_I, [_P] * 5 + [_I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "ecm_frame_prep_kitti_eval": (_I, [_P] * 5 + [_I, _I, _I, _I, _I, _P, _P, _P]),
    "ecm_frame_prep_packed": (_I, [_P, _P, _I] + [_P] * 4 + [_I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "ecm_eval_epe_scratch_bytes": (_LL, [_LL]),
    "ecm_eval_epe": (_I, [_P, _P, _P, _P, _LL] + [_I] * 7 + [_F, _P]),
    "ecm_disp_to_u16": (_I, [_P, _P, _I, _I, _I, _P, _P, _I, _I, _F, _P]),
    "ecm_gn3d_cluster_mode": (_I, [_I]),
    "ecm_gn3d_poll_ms": (_I, [_I]),
    "ecm_async_status": (_I, [_I]),
    "ecm_gn3d_scratch_bytes": (_LL, [_I, _I, _LL]),
    "ecm_gn3d_stats": (_I, [_P, _P, _P, _LL, _I, _I, _LL, _F, _P]),
    "ecm_gn3d_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _LL, _I, _P]),
    "ecm_gn3d_fwd": (_I, [_P] * 7 + [_LL, _I, _I, _LL, _I, _F, _P]),
    "ecm_gn3d_bwd": (_I, [_P] * 11 + [_LL, _I, _I, _LL, _I, _P]),
    "ecm_gn3d_cluster_bytes": (_LL, [_I]),
    "ecm_gn3d_cluster_preset": (_I, [_P, _LL, _P]),
    "ecm_gn3d_fwd_p": (_I, [_P] * 7 + [_LL, _P, _LL, _I, _I, _LL, _I, _F, _P]),
    "ecm_gn3d_bwd_p": (_I, [_P] * 11 + [_LL, _P, _LL, _I, _I, _LL, _I, _P]),
}

_lib = None


def load():
    """Load libecm_hip.so once; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C explicit-context-mapping-for-stereo-matching_amd/csrc`). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                continue            # reported by missing_symbols()/tests; calling it raises below
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def missing_symbols():
    lib = load()
    return [n for n in PROTOTYPES if not hasattr(lib, n)]


# Optional per-launch timing (bench.py's roofline leg): name -> list of (start_event, end_event, int_args).
# Events are recorded on torch's current stream, which is the stream every kernel is launched on.
_timers = {}
_time_all = False


def enable_timer(name: str):
    _timers[name] = []


def enable_all_timers():
    """Time EVERY int-returning entry point from now on (bench.py's per-family step breakdown)."""
    global _time_all
    _time_all = True


def disable_timers():
    global _time_all
    out = dict(_timers)
    _timers.clear()
    _time_all = False
    return out


class _Args(tuple):
    """The integer arguments of a timed launch (what bench.py indexes); `.longs` holds the 64-bit ones (sizes) beside them,
    `.ptrs` one flag per pointer argument (False = NULL: an optional operand that was not passed)."""
    longs = ()
    ptrs = ()


def call(name: str, *args):
    """Call an int-returning entry point and raise RuntimeError on a non-zero code."""
    lib = load()
    fn = getattr(lib, name, None)
    if fn is None:
        raise RuntimeError(f"libecm_hip.so does not export {name}")
    rec = _timers.get(name)
    if rec is None and _time_all:
        rec = _timers[name] = []
    if rec is not None:
        import torch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        ints = _Args(a for a in args if isinstance(a, int))
        ints.longs = tuple(a.value for a in args if isinstance(a, C.c_longlong))
        ints.ptrs = tuple(a is not None and bool(a.value) for a in args if a is None or isinstance(a, C.c_void_p))   # which optional operands were passed
        rec.append((s, e, ints))
    else:
        rc = fn(*args)
    if rc != 0:
        msg = lib.ecm_error_string(rc)
        raise RuntimeError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")


def query(name: str, *args):
    """Call a value-returning entry point (sizes)."""
    return getattr(load(), name)(*args)
