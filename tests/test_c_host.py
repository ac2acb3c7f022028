"""The C ABI from a torch-free host: C++ code written against include/ecm_hip.h and linked to libecm_hip.so
(tests/c_host/costvol_host.cpp), the way a C/C++ maintainer would bind the library."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "tests", "c_host")
HOST_SO = os.path.join(HOST_DIR, "libcostvol_host.so")
CHAIN_SO = os.path.join(HOST_DIR, "libchain_host.so")
LIB = os.path.join(ROOT, "explicit-context-mapping-for-stereo-matching_amd", "csrc", "libecm_hip.so")


def test_c_host_builds_against_the_header_and_library():
    """Compiles and links here (no GPU needed): the header is plain C and every symbol the host uses is exported."""
    if not os.path.exists(LIB):
        pytest.skip("libecm_hip.so not built")
    subprocess.run(["make", "-C", HOST_DIR], check=True, capture_output=True, timeout=600)
    assert os.path.exists(HOST_SO) and os.path.exists(CHAIN_SO)


@pytest.mark.gpu
def test_c_host_runs_bit_exact():
    """Runs the prebuilt host in-process (no exec from a GPU-initialised process): cost volume bit-exact, backward exact."""
    if not os.path.exists(HOST_SO):
        pytest.skip("tests/c_host/libcostvol_host.so not built (run __graft_entry__.build())")
    ctypes.CDLL(LIB, mode=ctypes.RTLD_GLOBAL)
    host = ctypes.CDLL(HOST_SO)
    host.costvol_host_main.restype = ctypes.c_int
    assert host.costvol_host_main() == 0


@pytest.mark.gpu
def test_c_host_chain_matches_the_reference_modules():
    """Weight packing, scratch queries (and the refusal of an undersized scratch), five kernels chained on one stream, the
    asynchronous-error query -- all from C++, against outputs of the reference's own classif1 / disparityregression /
    eight_related_context_mapping (tests/golden/make_golden_chain.py); final disparity within 2e-2 px."""
    if not os.path.exists(CHAIN_SO):
        pytest.skip("tests/c_host/libchain_host.so not built (run __graft_entry__.build())")
    ctypes.CDLL(LIB, mode=ctypes.RTLD_GLOBAL)
    host = ctypes.CDLL(CHAIN_SO)
    host.chain_host_main.restype = ctypes.c_int
    host.chain_host_main.argtypes = [ctypes.c_char_p]
    assert host.chain_host_main(os.path.join(ROOT, "tests", "golden", "chain_classif_heads.bin").encode()) == 0
