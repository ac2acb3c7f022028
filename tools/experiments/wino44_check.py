"""Winograd F(4x4,3x3) for the 2-D layers (csrc/conv_wino44.hip): forward and data gradient of
nn.Conv2d(3x3, stride 1, pad 1, bias=False) (convbn, cmfsm.py:36-46) against CPU F.conv2d autograd at ragged sizes, its
fp32 error against an fp64 evaluation next to that of the F(2x2) and direct kernels at an encoder layer's real size, and the
addend epilogue."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


def G(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


_SHAPES = [(1, 32, 32, 8, 8), (2, 32, 32, 33, 70), (1, 64, 64, 18, 35), (1, 128, 128, 17, 33), (1, 320, 128, 12, 36), (1, 32, 64, 5, 130),
           (1, 32, 32, 4, 4), (1, 32, 32, 3, 5), (1, 32, 32, 1, 1), (1, 32, 32, 2, 127), (1, 6, 10, 9, 9), (3, 36, 40, 7, 6), (1, 32, 32, 40, 129)]


@pytest.mark.parametrize("B,Ci,Co,H,W", _SHAPES)
def test_wino44_forward_dgrad_vs_torch(ecm, B, Ci, Co, H, W):
    ops = ecm.ops
    x = torch.randn(B, Ci, H, W, device="cuda", generator=G(1))
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=G(2)) * (2.0 / (9 * Ci)) ** 0.5
    y = ops._wino44_run(x, ops._wino44_pack(w, False), Co)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
    torch.testing.assert_close(y.cpu().double(), ref, rtol=1e-4, atol=2e-5)
    gy = torch.randn(B, Co, H, W, device="cuda", generator=G(3))
    sk = torch.randn(B, Ci, H, W, device="cuda", generator=G(4))
    gx = ops._wino44_run(gy, ops._wino44_pack(w, True), Ci, addend=sk)
    gref = F.conv_transpose2d(gy.double().cpu(), w.double().cpu(), None, 1, 1) + sk.double().cpu()
    torch.testing.assert_close(gx.cpu().double(), gref, rtol=1e-4, atol=2e-5)


def test_wino44_phase_planes(ecm):
    ops = ecm.ops
    x = torch.randn(2, 32, 4, 9, 13, device="cuda", generator=G(5))
    w = torch.randn(32, 32, 3, 3, device="cuda", generator=G(6)) * 0.08
    y = ops._wino44_run(x, ops._wino44_pack(w, False), 32)
    for p in range(4):
        torch.testing.assert_close(y[:, :, p].cpu().double(), F.conv2d(x[:, :, p].double().cpu(), w.double().cpu(), None, 1, 1),
                                   rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("Ci,Co,H,W", [(32, 32, 576, 960), (128, 128, 144, 240)])
def test_wino44_fp32_error_at_layer_size(ecm, Ci, Co, H, W):
    """Error against fp64 at an encoder layer's real size, beside F(2x2) and the direct kernel: F(4x4)'s transforms
    (constants up to 8, weights scaled by 1/24 .. 1/4) cost accuracy; the stage tolerance (rtol 1e-4 / atol 2e-5, DESIGN
    section 4) must hold with room to spare, and the relative rms error is reported."""
    ops = ecm.ops
    x = torch.randn(1, Ci, H, W, device="cuda", generator=G(7))
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=G(8)) * (2.0 / (9 * Co)) ** 0.5
    t64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    res = {}
    res["F(4x4)"] = ops._wino44_run(x, ops._wino44_pack(w, False), Co)
    res["F(2x2)"] = ops._wino_run(x, ops._wino_pack(w, 1, False), Co, 1)
    res["direct"] = ops._conv2d_run(x, ops._pack2d(w, False), Co, 3, 3, 1, 1, 1, 1, H, W)
    rms = {k: float((v.double() - t64).pow(2).mean().sqrt() / t64.pow(2).mean().sqrt()) for k, v in res.items()}
    print("relative rms error vs fp64:", {k: f"{v:.2e}" for k, v in rms.items()})
    torch.testing.assert_close(res["F(4x4)"].double(), t64, rtol=1e-4, atol=2e-5)
    assert rms["F(4x4)"] <= 2e-6, rms
