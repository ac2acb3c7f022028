"""Operand-shape contracts of the ops (`ops._need`).  The C ABI receives pointers and sizes: a second operand that is smaller
than the sizes imply would be read past its end by the kernel -- a device fault, not an exception.  Every public op therefore
checks, before it launches anything, what the torch ops it replaces check (F.conv3d / F.group_norm / torch.cat ... raise
RuntimeError on mismatched shapes), and the model built from them fails the way the reference does on a frame the
architecture cannot take (cmfsm.py:287-299: the hourglass's residual add of maps of different sizes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import ecm_amd
    return ecm_amd.ops


def r(*shape):
    return torch.randn(*shape, device="cuda")


CASES = {
    "cost_volume: maps of different widths": lambda o: o.cost_volume(r(1, 32, 8, 16), r(1, 32, 8, 12), 8),
    "cost_volume: zero disparities": lambda o: o.cost_volume(r(1, 32, 8, 16), r(1, 32, 8, 16), 0),
    "costvol_conv3d: maps of different shapes": lambda o: o.costvol_conv3d(r(1, 32, 8, 16), r(1, 32, 9, 16), r(32, 64, 3, 3, 3), 8),
    "conv3d: channel mismatch": lambda o: o.conv3d_k3(r(1, 32, 4, 8, 16), r(32, 16, 3, 3, 3)),
    "conv3d: 5x5x5 kernel": lambda o: o.conv3d_k3(r(1, 32, 4, 8, 16), r(32, 32, 5, 5, 5)),
    "conv3d: stride 3": lambda o: o.conv3d_k3(r(1, 32, 4, 8, 16), r(32, 32, 3, 3, 3), 3),
    "conv3d: 4-D input": lambda o: o.conv3d_k3(r(32, 4, 8, 16), r(32, 32, 3, 3, 3)),
    "conv3d 32->1: channel mismatch": lambda o: o.conv3d_k3(r(1, 16, 4, 8, 16), r(1, 32, 3, 3, 3)),
    "deconv3d: channel mismatch": lambda o: o.deconv3d_k3s2(r(1, 64, 2, 4, 8), r(32, 64, 3, 3, 3)),
    "conv2d: channel mismatch": lambda o: o.conv2d(r(2, 32, 16, 32), r(32, 64, 3, 3)),
    "conv2d_planes: channel mismatch": lambda o: o.conv2d_planes(r(1, 64, 4, 8, 16), r(128, 128, 3, 3)),
    "group norm: skip of another shape": lambda o: o.group_norm_act(r(1, 32, 4, 8, 16), r(32), r(32), skip=r(1, 32, 4, 7, 16)),
    "group norm: 48 channels in 32 groups": lambda o: o.group_norm_act(r(1, 48, 4, 8, 16), r(48), r(48)),
    "group norm: 16 channels in 32 groups": lambda o: o.group_norm_act(r(1, 16, 4, 8, 16), r(16), r(16)),
    "group norm: gamma of another width": lambda o: o.group_norm_act(r(1, 64, 4, 8, 16), r(32), r(64)),
    "group norm: head beyond the batch": lambda o: o.group_norm_act(r(2, 32, 8, 16), r(32), r(32), head=3),
    "classifier tail: gamma of another width": lambda o: o.classifier_tail(r(1, 32, 4, 8, 16), r(16), r(32), r(1, 32, 3, 3, 3)),
    "classifier tail: 64 channels": lambda o: o.classifier_tail(r(1, 64, 4, 8, 16), r(64), r(64), r(1, 64, 3, 3, 3)),
    "aggregate9: weights of another size": lambda o: o.ecm_aggregate9(r(3, 1, 8, 16), r(1, 9, 32, 60), 4),
    "aggregate9: weights of another batch": lambda o: o.ecm_aggregate9(r(3, 2, 8, 16), r(1, 9, 32, 64), 4),
    "ecm weights: MLP kernels of the wrong size": lambda o: o.ecm_weights9(r(1, 32, 8, 16), r(1, 32, 32, 64), r(32, 64, 1, 1), r(16, 32, 1, 1),
                                                                           r(8, 16, 1, 1), r(1, 8, 1, 1)),
    "ecm weights: maps of different batch": lambda o: o.ecm_weights9(r(2, 32, 8, 16), r(1, 32, 32, 64), r(32, 66, 1, 1), r(16, 32, 1, 1),
                                                                     r(8, 16, 1, 1), r(1, 8, 1, 1)),
    "ecm weights: hr not a multiple of lr": lambda o: o.ecm_weights9(r(1, 32, 8, 16), r(1, 32, 33, 64), r(32, 66, 1, 1), r(16, 32, 1, 1),
                                                                     r(8, 16, 1, 1), r(1, 8, 1, 1)),
    "context weights: unknown variant": lambda o: o.context_weights(r(1, 32, 8, 16), r(1, 32, 32, 64), r(32, 66, 1, 1), r(16, 32, 1, 1),
                                                                    r(8, 16, 1, 1), r(1, 8, 1, 1), 3),
    "volume mapping: m5 of another size": lambda o: o.volume_mapping(r(3, 1, 12, 8, 16), r(1, 5, 64, 128), r(1, 3, 128, 256), 16),
    "volume mapping: mt3 with 5 planes": lambda o: o.volume_mapping(r(3, 1, 12, 8, 16), r(1, 5, 128, 256), r(1, 5, 128, 256), 16),
    "softargmin heads: 4-D input": lambda o: o.softargmin_heads(r(3, 48, 8, 16)),
    "stereo loss: ground truth of another size": lambda o: o.stereo_loss3((r(1, 1, 8, 16), r(1, 1, 8, 16), r(1, 1, 8, 16)), r(1, 8, 12)),
    "eval_epe: crop beyond the prediction": lambda o: o.eval_epe(r(1, 1, 16, 32), r(1, 16, 32), crop_h=20, crop_w=32),
    "frame_prep: window outside the frame": lambda o: o.frame_prep(r(1, 64, 96, 7), [40], [0], 32, 64),
    "CPU tensor": lambda o: o.conv3d_k3(torch.randn(1, 32, 4, 8, 16), torch.randn(32, 32, 3, 3, 3)),
    "fp64 tensor": lambda o: o.group_norm_act(r(1, 32, 4, 8, 16).double(), r(32).double(), r(32).double()),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_mismatched_operands_raise_before_any_launch(ops, name):
    torch.cuda.synchronize()
    with pytest.raises((RuntimeError, ValueError)):
        CASES[name](ops)
    torch.cuda.synchronize()                                   # nothing faulted behind the exception
    ops.check_async_errors()
    y = ops.conv3d_k3(r(1, 32, 4, 8, 16), r(32, 32, 3, 3, 3))   # and the device still takes work
    assert torch.isfinite(y).all()


def test_frame_the_architecture_cannot_take_fails_like_the_reference(ops):
    """264 rows: 66 at quarter resolution, 33 and 17 inside the hourglass, and its transposed convolution returns 34 -- the
    reference's `conv5 + presqu` (cmfsm.py:287-299) raises a size-mismatch RuntimeError there; so does the fused GroupNorm + add."""
    import ecm_amd
    torch.manual_seed(0)
    model = ecm_amd.get_model("cmfsm").cuda().eval()
    left, right = r(1, 3, 264, 256), r(1, 3, 264, 256)
    with torch.no_grad(), pytest.raises(RuntimeError):
        model(left, right)
    torch.cuda.synchronize()
    ops.check_async_errors()
    with torch.no_grad():                                      # the same model on a frame it can take
        out = model(r(1, 3, 256, 256), r(1, 3, 256, 256))
    assert all(torch.isfinite(o).all() for o in out)


def test_groupnorm_refuses_in_place_calls(ops):
    """The C ABI itself (a host that binds it directly): y == x or gx == gy would let one workgroup overwrite what another of the
    same (sample, group) span still reads (the span's pivot samples; since round 4 also the output parked in LDS) -> ECM_EINVAL."""
    import ctypes as C
    lib, p = ops._lib, ops._p
    B, Cc, S = 1, 32, 4 * 8 * 16
    x, gm, bt = r(B, Cc, 4, 8, 16), r(Cc), r(Cc)
    stats = torch.empty(B, 32, 2, device="cuda")
    nb = lib.query("ecm_gn3d_scratch_bytes", B, Cc, C.c_longlong(S))
    scratch = torch.empty(nb // 4 + 16, device="cuda")
    keep = x.clone()
    with pytest.raises(RuntimeError):
        lib.call("ecm_gn3d_fwd", p(x), p(gm), p(bt), None, p(x), p(stats), p(scratch), C.c_longlong(nb), B, Cc, C.c_longlong(S), 1,
                 C.c_float(1e-5), ops._stream())
    y = torch.empty_like(x)
    lib.call("ecm_gn3d_fwd", p(x), p(gm), p(bt), None, p(y), p(stats), p(scratch), C.c_longlong(nb), B, Cc, C.c_longlong(S), 1,
             C.c_float(1e-5), ops._stream())
    gy, gg, gb = r(B, Cc, 4, 8, 16), torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    with pytest.raises(RuntimeError):
        lib.call("ecm_gn3d_bwd", p(x), p(stats), p(gm), p(bt), None, p(gy), p(gy), None, p(gg), p(gb), p(scratch), C.c_longlong(nb),
                 B, Cc, C.c_longlong(S), 1, ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(x, keep)                                # nothing was launched on the refused calls
    ops.check_async_errors()
