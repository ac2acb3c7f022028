"""CPU-side checks of the C ABI: libecm_hip.so loads (no GPU needed to dlopen it), exports every symbol that
include/ecm_hip.h declares, and the ctypes prototypes cover exactly that set.  No compute calls."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ecm_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ecm_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_mod():
    import ecm_amd
    if not os.path.exists(ecm_amd._lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return ecm_amd._lib


def test_header_declares_functions():
    names = declared_functions()
    assert "ecm_costvol_concat_fwd" in names and "ecm_conv3d_k3_fwd" in names and len(names) >= 20


def test_library_exports_every_declared_symbol(lib_mod):
    lib = ctypes.CDLL(lib_mod.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in ecm_hip.h but not exported: {missing}"


def test_python_prototypes_match_header(lib_mod):
    assert sorted(lib_mod.PROTOTYPES) == declared_functions()
    assert lib_mod.missing_symbols() == []


def test_version_and_error_strings(lib_mod):
    assert lib_mod.query("ecm_abi_version") >= 1
    lib = lib_mod.load()
    assert b"invalid" in lib.ecm_error_string(-1)
    assert b"scratch" in lib.ecm_error_string(-3)


def test_size_queries_need_no_gpu(lib_mod):
    assert lib_mod.query("ecm_conv3d_packed_floats", 32, 32) == 27 * 32 * 32
    assert lib_mod.query("ecm_conv3d_packed_floats", 32, 1) == 27 * 32 * 32          # Co padded to 32
    assert lib_mod.query("ecm_weights9_scratch_bytes", 1, 144, 240) == 144 * 240 * 32 * 4
    assert lib_mod.query("ecm_gn3d_scratch_bytes", 1, 32, ctypes.c_longlong(48 * 144 * 240)) > 0
    assert lib_mod.query("ecm_weights9_bwd_scratch_bytes", 1, 8, 12, 4) > 0
    assert lib_mod.query("ecm_weights9_bwd_scratch_bytes", 1, 8, 12, 6) == 0          # unsupported scale (s % 4 != 0)
    assert lib_mod.query("ecm_context_weights_bwd_scratch_bytes", 1, 8, 12, 8, 1) > 0


def test_null_pointers_are_rejected_without_touching_the_gpu(lib_mod):
    lib = lib_mod.load()
    assert lib.ecm_costvol_concat_fwd(None, None, None, 1, 1, 1, 1, 1, None) == -1
    assert lib.ecm_conv3d_k3_fwd(None, None, None, 1, 32, 32, 4, 4, 4, 1, None) == -1


def test_product_path_refuses_cpu_tensors():
    import torch
    import ecm_amd
    with pytest.raises(RuntimeError):
        ecm_amd.ops.cost_volume(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4), 2)
    with pytest.raises(RuntimeError):
        ecm_amd.get_model("cmfsm").hot_path(torch.zeros(1, 32, 8, 12), torch.zeros(1, 32, 32, 48), torch.zeros(1, 32, 8, 12))


def test_state_dict_contract_on_cpu(cmfsm_shapes):
    import ecm_amd
    sd = ecm_amd.get_model("cmfsm").state_dict()
    assert list(sd.keys()) == list(cmfsm_shapes.keys())
    assert all(list(sd[k].shape) == cmfsm_shapes[k] for k in sd)
    assert sum(v.numel() for v in sd.values()) == 5255368


def test_wino_asm_loads_not_read_early():
    """csrc/conv_wino.hip issues its patch loads as inline asm with hand-counted s_waitcnt: prove on the built code object
    that no instruction touches a load's destination register while the load may be in flight (tools/check_wino_isa.py)."""
    import subprocess
    import sys
    so = os.path.join(ROOT, "explicit-context-mapping-for-stereo-matching_amd", "csrc", "libecm_hip.so")
    tools = ("/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-objcopy", "/opt/rocm/lib/llvm/bin/clang-offload-bundler")
    if not os.path.exists(so) or not all(os.path.exists(t) for t in tools):
        pytest.skip("needs the built library and the ROCm llvm binutils")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_wino_isa.py"), so], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "2 kernels checked, 0 problems" in r.stdout



def test_no_kernel_uses_private_scratch_memory():
    """Every kernel of libecm_hip.so must have `.private_segment_fixed_size` 0 and no VGPR spills (tools/check_private_segment.py
    reads the AMDGPU metadata of the gfx950 code objects): a spilling kernel needs scratch memory provisioned at dispatch --
    round 2's one such kernel was the one that aborted -- and pays for every spilled access."""
    import subprocess
    import sys
    so = os.path.join(ROOT, "explicit-context-mapping-for-stereo-matching_amd", "csrc", "libecm_hip.so")
    tools = ("/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/lib/llvm/bin/llvm-objcopy", "/opt/rocm/lib/llvm/bin/clang-offload-bundler")
    if not os.path.exists(so) or not all(os.path.exists(t) for t in tools):
        pytest.skip("needs the built library and the ROCm llvm binutils")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_private_segment.py"), so], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 with a private segment" in r.stdout
