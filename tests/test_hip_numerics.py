"""Conditioning of the fp32 kernels, measured against fp64 (found with tests/diag_stage_error.py / diag_encoder_error.py in
round 3): every stage of the HIP path must be about as close to the fp64 truth as torch's own fp32 evaluation of the same
stage -- in particular where the problem is ill-conditioned."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


@pytest.mark.parametrize("shape", [(1, 32, 1, 2), (8, 32, 2, 3), (2, 32, 4, 8), (2, 64, 1, 1, 2), (1, 32, 8, 36, 60), (1, 128, 36, 64)])
@pytest.mark.parametrize("offset", [0.0, 3.0, 100.0])
def test_groupnorm_on_offset_data_is_as_accurate_as_torch(ecm, shape, offset):
    """GroupNorm(32) where |mean| >> std -- the SPP branches of the encoder normalise 1x2 .. 2x3 maps per channel
    (cmfsm.py:152-170 at 256x512 / 576x960) whose two to six values differ in the third digit.  Variance as
    E[x^2] - E[x]^2 loses those digits; the kernels accumulate around a pivot instead.  Yardstick: F.group_norm in fp64;
    the kernel may be at most 4x as far from it as F.group_norm in fp32 (plus one ulp of the output scale)."""
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + int(offset))
    x = torch.randn(shape, device="cuda", generator=g) * 0.01 + offset + torch.randn(shape[0], shape[1], *([1] * (len(shape) - 2)),
                                                                                     device="cuda", generator=g)
    C = shape[1]
    gm, bt = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
    y = ecm.ops.group_norm_act(x, gm, bt, None, False)
    t64 = F.group_norm(x.double(), 32, gm.double(), bt.double(), 1e-5)
    t32 = F.group_norm(x, 32, gm, bt, 1e-5)
    e_hip, e_t32 = float((y.double() - t64).abs().max()), float((t32.double() - t64).abs().max())
    assert e_hip <= 4.0 * e_t32 + 2e-7 * float(t64.abs().max()), (shape, offset, e_hip, e_t32)


@pytest.mark.parametrize("shape", [(2, 32, 4, 8), (1, 32, 8, 36, 60), (1, 128, 36, 64), (2, 32, 24, 72, 120)])
@pytest.mark.parametrize("spike", [1e2, 1e4])
def test_groupnorm_with_an_outlier_first_element(ecm, shape, spike):
    """a pivot taken from the span's first element alone would make every element of a group inherit the
    cancellation when THAT element is an outlier (an activation spike in the corner).  Data: offset 5, std 0.01, and the
    first element of every group span set to offset + spike.  Same yardstick as above (fp64, 4 x torch's fp32 error)."""
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(shape, device="cuda", generator=g) * 0.01 + 5.0
    C = shape[1]
    cpg = C // 32
    xv = x.view(shape[0], 32, -1)
    xv[:, :, 0] = 5.0 + spike
    gm, bt = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
    y = ecm.ops.group_norm_act(x, gm, bt, None, False)
    old = ecm.ops.gn_cluster_mode(0)                       # the two-stage kernels share the pivot rule
    try:
        y2 = ecm.ops.group_norm_act(x, gm, bt, None, False)
    finally:
        ecm.ops.gn_cluster_mode(old)
    t64 = F.group_norm(x.double(), 32, gm.double(), bt.double(), 1e-5)
    t32 = F.group_norm(x, 32, gm, bt, 1e-5)
    e_t32 = float((t32.double() - t64).abs().max())
    for out in (y, y2):
        e_hip = float((out.double() - t64).abs().max())
        assert e_hip <= 4.0 * e_t32 + 2e-7 * float(t64.abs().max()), (shape, spike, e_hip, e_t32)
    assert cpg >= 1
