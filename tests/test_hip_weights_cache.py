"""Packed-weight layouts must follow EVERY kind of weight update: optimizer steps, load_state_dict, and
in-place writes through `.data` that the version counter does not see -- eagerly, under a captured HIP graph, and inside
the explicit ops.frozen_weights() opt-in (where `.data` writes need ops.invalidate_packed())."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


def _inputs(seed=0, h=256, w=256):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(1, 3, h, w, device="cuda", generator=g), torch.randn(1, 3, h, w, device="cuda", generator=g)


def _perturbed_state(model, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return {k: v + 0.05 * v.abs().mean() * torch.randn(v.shape, device=v.device, generator=g) for k, v in model.state_dict().items()}


def test_data_write_is_seen_by_the_next_forward(ecm):
    torch.manual_seed(1)
    model = ecm.get_model("cmfsm").cuda().eval()
    left, right = _inputs()
    with torch.no_grad():
        before = model(left, right)[2].clone()
        new = _perturbed_state(model, 5)
        for k, p in model.named_parameters():
            p.data.copy_(new[k])                           # does NOT bump p._version
        got = model(left, right)[2].clone()
        fresh = ecm.get_model("cmfsm").cuda().eval()
        fresh.load_state_dict(new)
        want = fresh(left, right)[2]
    assert not torch.equal(before, got)
    assert torch.equal(got, want)


def test_frozen_weights_reuses_and_invalidates(ecm):
    ops = ecm.ops
    torch.manual_seed(2)
    conv = torch.nn.Conv3d(32, 32, 3, padding=1, bias=False).cuda()
    x = torch.randn(1, 32, 6, 10, 12, device="cuda")
    with torch.no_grad(), ops.frozen_weights():
        y0 = ops.conv3d_k3(x, conv.weight, 1)
        assert getattr(conv.weight, "_ecm_packed", None), "inside frozen_weights() the packed layout is cached"
        packed = conv.weight._ecm_packed["w3"][1]
        ops.conv3d_k3(x, conv.weight, 1)
        assert conv.weight._ecm_packed["w3"][1] is packed                      # reused
        conv.weight.mul_(2.0)                                                  # version counter moves: re-packed
        y1 = ops.conv3d_k3(x, conv.weight, 1)
        torch.testing.assert_close(y1, 2.0 * y0, rtol=1e-4, atol=1e-5)
        conv.weight.data.mul_(0.5)                                             # invisible to the version counter ...
        ops.invalidate_packed()                                                # ... so the contract asks for this
        y2 = ops.conv3d_k3(x, conv.weight, 1)
        torch.testing.assert_close(y2, y0, rtol=1e-4, atol=1e-5)
    with torch.no_grad():                                                       # outside: never cached
        conv.weight.data.mul_(3.0)
        y3 = ops.conv3d_k3(x, conv.weight, 1)
    torch.testing.assert_close(y3, 3.0 * y0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(y0, F.conv3d(x, conv.weight / 3.0, None, 1, 1), rtol=1e-3, atol=1e-4)


def test_graph_replay_follows_load_state_dict(ecm):
    from importlib import import_module
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    torch.manual_seed(3)
    model = ecm.get_model("cmfsm").cuda().eval()
    left, right = _inputs(7)
    graphed = D.GraphedForward(model, left, right)
    a = graphed(left, right)[2].clone()
    new = _perturbed_state(model, 9)
    model.load_state_dict(new)                             # test.py's checkpoint loop: same model object, new weights
    b = graphed(left, right)[2].clone()
    old = ecm.ops.gn_cluster_mode(0)                       # a captured GroupNorm runs the two-stage kernels: like against like
    try:
        with torch.no_grad():
            want = model(left, right)[2]
    finally:
        ecm.ops.gn_cluster_mode(old)
    assert not torch.equal(a, b)
    assert torch.equal(b, want)


def test_winograd_range_check_at_the_descriptor_bound(ecm):
    """ecm_conv_wino_fwd addresses 32 channel planes with 32-bit byte offsets: a plane set above 2^24 elements is refused
    (ECM_EUNSUP) instead of wrapping, and ops routes such a volume to the direct kernel."""
    lib = ecm._lib
    import ctypes as C
    x = torch.zeros(16, device="cuda")
    D, H, W = 65, 512, 512                                  # 17.0 M elements per channel > 2^24
    with pytest.raises(RuntimeError, match="not supported"):
        lib.call("ecm_conv_wino_fwd", C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 1, 32, 32,
                 D, H, W, 3, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert not ecm.ops._wino_ok(torch.empty(0, 32, D, H, W, device="meta"))
    assert ecm.ops._wino_ok(torch.empty(0, 32, 64, 512, 512, device="meta"))


@pytest.mark.parametrize("kd,Co,Ci", [(3, 32, 32), (3, 32, 64), (3, 24, 40), (1, 32, 32), (1, 128, 320), (1, 480, 32), (1, 20, 36)])
def test_wino_pack_weight2_equals_the_two_single_packs(ecm, kd, Co, Ci):
    """ecm_conv_wino_pack_weight2 (forward + data-gradient layout from one launch) against two ecm_conv_wino_pack_weight
    launches, bit for bit, ragged channel counts included."""
    import ctypes as C
    lib = ecm._lib
    g = torch.Generator(device="cuda").manual_seed(kd * 1000 + Co + Ci)
    w = torch.randn((Co, Ci) + ((3, 3, 3) if kd == 3 else (3, 3)), device="cuda", generator=g)
    nf, nb = lib.query("ecm_conv_wino_packed_floats", Ci, Co, kd), lib.query("ecm_conv_wino_packed_floats", Co, Ci, kd)
    both = torch.full((nf + nb,), float("nan"), device="cuda")
    pf, pb = torch.full((nf,), float("nan"), device="cuda"), torch.full((nb,), float("nan"), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    lib.call("ecm_conv_wino_pack_weight2", p(w), p(both), Co, Ci, kd, st)
    lib.call("ecm_conv_wino_pack_weight", p(w), p(pf), Co, Ci, kd, 0, st)
    lib.call("ecm_conv_wino_pack_weight", p(w), p(pb), Co, Ci, kd, 1, st)
    torch.cuda.synchronize()
    assert torch.isfinite(both).all()
    assert torch.equal(both[:nf], pf) and torch.equal(both[nf:], pb)


@pytest.mark.parametrize("dim", [2, 3])
def test_training_backward_uses_the_layout_packed_in_forward(ecm, dim):
    """A layer's training step packs its weight ONCE (forward + data-gradient layout together); the backward must find the
    second layout on the autograd context instead of launching another pack in front of its data-gradient kernel -- counted
    on the C-ABI launches themselves.  Under no_grad only the forward layout is packed."""
    ops, lib = ecm.ops, ecm._lib
    g = torch.Generator(device="cuda").manual_seed(3)
    if dim == 3:
        w = torch.randn(32, 32, 3, 3, 3, device="cuda", generator=g).requires_grad_()
        x = torch.randn(1, 32, 6, 10, 12, device="cuda", generator=g).requires_grad_()
        f = lambda: ops.conv3d_k3(x, w, 1)
    else:
        w = torch.randn(32, 32, 3, 3, device="cuda", generator=g).requires_grad_()
        x = torch.randn(1, 32, 20, 24, device="cuda", generator=g).requires_grad_()
        f = lambda: ops.conv2d(x, w, 1, 1)
    for name in ("ecm_conv_wino_pack_weight", "ecm_conv_wino_pack_weight2", "ecm_conv_wino_fwd"):
        lib.enable_timer(name)
    y = f()
    y.sum().backward()
    torch.cuda.synchronize()
    t = lib.disable_timers()
    assert len(t["ecm_conv_wino_pack_weight2"]) == 1 and len(t["ecm_conv_wino_pack_weight"]) == 0, {k: len(v) for k, v in t.items()}
    assert len(t["ecm_conv_wino_fwd"]) == 2                                   # forward + data gradient
    gx = x.grad.clone()
    # same data gradient as with the backward layout packed on its own
    gy = torch.ones_like(y)
    Ci = w.shape[1]
    ref = ops._wino_run(gy, ops._wino_pack(w.detach(), 3 if dim == 3 else 1, True), Ci, 3 if dim == 3 else 1)
    assert torch.equal(gx, ref)
    lib.enable_timer("ecm_conv_wino_pack_weight")
    lib.enable_timer("ecm_conv_wino_pack_weight2")
    with torch.no_grad():
        f()
    torch.cuda.synchronize()
    t = lib.disable_timers()
    assert len(t["ecm_conv_wino_pack_weight2"]) == 0 and len(t["ecm_conv_wino_pack_weight"]) == 1
