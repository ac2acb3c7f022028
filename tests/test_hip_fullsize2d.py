"""The 2-D family at the sizes the training step runs (BASELINE cfg 2: 4 pairs = 8 images of 576x960): every layer shape of
cmfsm's encoder (cmfsm.py:126-236) and the two class convolutions of the collapsed cost volume (cmfsm.py:667-684) --
forward, data gradient, weight gradient -- against F.conv2d on the device, i.e. MIOpen reached through torch: an
independent implementation (the CPU oracle would take minutes per layer here).  Companion of test_hip_fullsize.py, which
does the same for the 3-D kernels.  Tolerances: the two sides sum up to 9*384 fp32 products in different orders; MIOpen's
own algorithm choice (Winograd / implicit GEMM / direct) is not ours to control, so 1e-3 relative with an absolute term
of 1e-4 x the tensor's scale."""
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    torch.backends.cudnn.benchmark = False
    return ecm_amd


def _close(a, b, what):
    scale = float(b.detach().abs().max())
    torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4 * max(scale, 1e-30), msg=lambda m: f"{what}: {m}")


def _run_pair(fn_hip, fn_ref, x, w, what):
    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    y, ref = fn_hip(xg, wg), fn_ref(xr, wr)
    assert y.shape == ref.shape, (what, y.shape, ref.shape)
    _close(y, ref, what + " forward")
    G = torch.randn(ref.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(99))
    y.backward(G)
    ref.backward(G)
    _close(xg.grad, xr.grad, what + " data gradient")
    _close(wg.grad, wr.grad, what + " weight gradient")


# (label, images, Ci, Co, H, W, k, stride, dil): cmfsm's encoder at 8 x 3 x 576 x 960
_LAYERS = [
    ("firstconv.0 stem", 8, 3, 32, 576, 960, 3, 1, 1),
    ("firstconv 32->32", 8, 32, 32, 576, 960, 3, 1, 1),
    ("secondconv stride 2", 8, 32, 32, 576, 960, 3, 2, 1),
    ("layer1 32->32", 8, 32, 32, 288, 480, 3, 1, 1),
    ("layer2.0 stride 2", 8, 32, 64, 288, 480, 3, 2, 1),
    ("layer2.0 downsample 1x1 s2", 8, 32, 64, 288, 480, 1, 2, 1),
    ("layer2 64->64", 8, 64, 64, 144, 240, 3, 1, 1),
    ("layer3.0 64->128", 8, 64, 128, 144, 240, 3, 1, 1),
    ("layer3.0 downsample 1x1", 8, 64, 128, 144, 240, 1, 1, 1),
    ("layer3 128->128", 8, 128, 128, 144, 240, 3, 1, 1),
    ("layer4 128->128 dilation 2 (direct kernel)", 8, 128, 128, 144, 240, 3, 1, 2),
    ("lastconv 320->128", 8, 320, 128, 144, 240, 3, 1, 1),
    ("lastconv 1x1 128->32", 8, 128, 32, 144, 240, 1, 1, 1),
    ("branch 1x1 128->32 on the 2x3 map", 8, 128, 32, 2, 3, 1, 1, 1),
]


@pytest.mark.parametrize("label,B,Ci,Co,H,W,k,stride,dil", _LAYERS, ids=[l[0] for l in _LAYERS])
def test_encoder_layer_at_bench_geometry_vs_miopen(ecm, label, B, Ci, Co, H, W, k, stride, dil):
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(label.encode()) % 1000)
    x = torch.randn(B, Ci, H, W, device="cuda", generator=g)
    w = torch.randn(Co, Ci, k, k, device="cuda", generator=g) * (2.0 / (k * k * Co)) ** 0.5      # the reference's init rule
    pad = dil * (k - 1) // 2
    assert ecm.ops.conv2d_supported(Ci, Co, k, k, stride, dil)
    _run_pair(lambda a, b: ecm.ops.conv2d(a, b, stride, dil), lambda a, b: F.conv2d(a, b, None, stride, pad, dil), x, w, label)


def test_dilated_stage_as_phase_planes_vs_miopen(ecm):
    """layer4 (cmfsm.py:150: 128 -> 128, dilation 2) as the model runs it: four phase planes through the Winograd kernels
    (ops.phase_split / conv2d_planes / phase_merge) vs one dilated F.conv2d."""
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.randn(8, 128, 144, 240, device="cuda", generator=g)
    w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * (2.0 / (9 * 128)) ** 0.5
    ops = ecm.ops
    _run_pair(lambda a, b: ops.phase_merge(ops.conv2d_planes(ops.phase_split(a, 2), b), 2),
              lambda a, b: F.conv2d(a, b, None, 1, 2, 2), x, w, "layer4 phase planes")


def test_class_convolutions_at_bench_geometry_vs_miopen(ecm):
    """P: 3x3, 32 -> 15*32 on the reference features; Q: sheared 3x5, 32 -> 6*32 on the target features with left padding
    4 / right padding 2 (output width w + 2) -- at [4,32,144,240] (ops.costvol_conv3d)."""
    g = torch.Generator(device="cuda").manual_seed(23)
    B, h, w = 4, 144, 240
    L, R = torch.randn(B, 32, h, w, device="cuda", generator=g), torch.randn(B, 32, h, w, device="cuda", generator=g)
    wP = torch.randn(480, 32, 3, 3, device="cuda", generator=g) * 0.08
    wQ = torch.randn(192, 32, 3, 5, device="cuda", generator=g) * 0.08
    _run_pair(lambda a, b: ecm.ops.conv2d(a, b, 1, 1, 1, 1, h, w), lambda a, b: F.conv2d(a, b, None, 1, 1), L, wP, "class conv P")
    _run_pair(lambda a, b: ecm.ops.conv2d(a, b, 1, 1, 1, 4, h, w + 2), lambda a, b: F.conv2d(F.pad(a, (4, 2, 1, 1)), b), R, wQ,
              "class conv Q (sheared 3x5)")


def test_collapsed_first_conv_vs_conv3d_of_the_explicit_volume_b4(ecm):
    """ops.costvol_conv3d at batch 4 against MIOpen's conv3d of the explicit [4,64,48,144,240] volume built by plain torch
    slicing (cmfsm.py:667-684): values, both feature gradients, the weight gradient."""
    g = torch.Generator(device="cuda").manual_seed(29)
    B, h, w, D = 4, 144, 240, 48
    L, R = torch.randn(B, 32, h, w, device="cuda", generator=g), torch.randn(B, 32, h, w, device="cuda", generator=g)
    wt = torch.randn(32, 64, 3, 3, 3, device="cuda", generator=g) * (2.0 / (27 * 32)) ** 0.5

    def explicit(l, r, wgt):
        cost = torch.zeros(B, 64, D, h, w, device="cuda")
        for d in range(D):
            cost[:, :32, d, :, d:] = l[..., d:]
            cost[:, 32:, d, :, d:] = r[..., :w - d]
        return F.conv3d(cost, wgt, None, 1, 1)
    a = [t.clone().requires_grad_() for t in (L, R, wt)]
    b = [t.clone().requires_grad_() for t in (L, R, wt)]
    y = ecm.ops.costvol_conv3d(a[0], a[1], a[2], D)
    ref = explicit(*b)
    _close(y, ref, "collapsed conv forward")
    G = torch.randn(ref.shape, device="cuda", generator=g)
    y.backward(G)
    ref.backward(G)
    for i, nm in enumerate(("gL", "gR", "gW")):
        _close(a[i].grad, b[i].grad, "collapsed conv " + nm)
