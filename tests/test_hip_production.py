"""Production-geometry parity on the MI355X: the kernels and host paths that bench.py actually runs, at the sizes it runs
them (BASELINE.json cfg 2: 576x960, D=192, batch 4; cfg 5: 1080x1920, D=256 cost volume), checked against independent
DEVICE-side arithmetic -- plain torch ops / autograd of a torch restatement of the same math (SURVEY.md 8a closed forms),
never against the kernels themselves.  The CPU oracle would need minutes to hours at these sizes; it pins the same kernels
at small sizes in test_hip_parity.py.  Tolerances are fp32 re-association bounds, stated per test."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


@pytest.fixture(autouse=True)
def _no_async_errors(ecm):
    yield
    ecm.ops.check_async_errors()          # a GroupNorm cluster time-out anywhere in a test is a failure


def gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


def rel_close(a, b, tol, what="", kinks=0, kink_tol=0.1):
    """max |a - b| <= tol * max |b| (a whole-tensor bound: gradients span many orders of magnitude).
    `kinks`: number of elements allowed to miss `tol` (but not kink_tol * max |b|).  A ReLU / LeakyReLU whose pre-activation
    is within an ulp or two of zero can take the other branch when the same sum is formed in a different order; among
    10^8..10^9 activations that happens a few dozen times, and each flip changes ONE term of the affected gradient sums by
    up to 99 % of that term -- a property of comparing two fp32 evaluation orders at this size, not of either one."""
    a, b = a.detach(), b.detach()
    d = (a - b).abs()
    ref = float(b.abs().max())
    err = float(d.max())
    if kinks == 0:
        assert err <= tol * ref + 1e-30, f"{what}: max err {err:.3e} vs max |ref| {ref:.3e} (tol {tol})"
        return
    n_out = int((d > tol * ref).sum())
    assert n_out <= kinks and err <= kink_tol * ref, \
        f"{what}: {n_out} elements beyond {tol} * max |ref| (allowed {kinks}), max err {err:.3e} vs max |ref| {ref:.3e}"


# ---------------------------------------------------------------------------------------------- (a) cfg 5 cost volume
def test_cost_volume_cfg5_1080p_d256(ecm):
    """cmfsm.py:667-682 at BASELINE cfg 5: L,R [1,32,270,480], D'=64 -> [1,64,64,270,480] (2.12 GB).  Bit-exact slices,
    zero wedge, and the adjoint identity <costvol(L,R), G> = <L, gL> + <R, gR> for the backward kernel."""
    h, w, D = 270, 480, 64
    L = torch.randn(1, 32, h, w, device="cuda", generator=gen(11))
    R = torch.randn(1, 32, h, w, device="cuda", generator=gen(12))
    cost = ecm.ops.cost_volume(L, R, D)
    assert cost.shape == (1, 64, D, h, w)
    for d in range(D):                                   # every plane, bit for bit
        assert torch.equal(cost[:, :32, d, :, d:], L[..., d:])
        assert torch.equal(cost[:, 32:, d, :, d:], R[..., :w - d])
        assert not cost[:, :, d, :, :d].any()
    G = torch.randn(cost.shape, device="cuda", generator=gen(13))
    Lg, Rg = L.clone().requires_grad_(), R.clone().requires_grad_()
    ecm.ops.cost_volume(Lg, Rg, D).backward(G)
    # the backward is a plain sum over d of shifted slices: compare with that sum formed by torch in fp64
    gl = torch.zeros(1, 32, h, w, device="cuda", dtype=torch.float64)
    gr = torch.zeros_like(gl)
    for d in range(D):
        gl[..., d:] += G[:, :32, d, :, d:]
        gr[..., :w - d] += G[:, 32:, d, :, d:]
    torch.testing.assert_close(Lg.grad.double(), gl, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(Rg.grad.double(), gr, rtol=1e-5, atol=2e-5)


# ------------------------------------------------------------------------- (b) collapsed cost volume + first conv
def test_costvol_conv3d_production_size(ecm):
    """The headline path (ops.costvol_conv3d: class-indexed 2-D convolutions + assembly, cmfsm.py:667-684) at
    [2,32,144,240], D'=48 against the reference's explicit op sequence on device -- cost_volume kernel (bit-exact, test
    above) + 64->32 Conv3d (checked against MIOpen at this size in test_hip_fullsize.py) -- values and all three gradients."""
    B, h, w, D = 2, 144, 240, 48
    L = torch.randn(B, 32, h, w, device="cuda", generator=gen(21))
    R = torch.randn(B, 32, h, w, device="cuda", generator=gen(22))
    W = torch.randn(32, 64, 3, 3, 3, device="cuda", generator=gen(23)) * (2.0 / (27 * 32)) ** 0.5
    G = torch.randn(B, 32, D, h, w, device="cuda", generator=gen(24))
    a = [t.clone().requires_grad_() for t in (L, R, W)]
    b = [t.clone().requires_grad_() for t in (L, R, W)]
    y = ecm.ops.costvol_conv3d(a[0], a[1], a[2], D)
    ref = ecm.ops.conv3d_k3(ecm.ops.cost_volume(b[0], b[1], D), b[2], 1)
    rel_close(y, ref, 2e-5, "y")
    y.backward(G)
    ref.backward(G)
    rel_close(a[0].grad, b[0].grad, 5e-5, "gL")
    rel_close(a[1].grad, b[1].grad, 5e-5, "gR")
    rel_close(a[2].grad, b[2].grad, 2e-4, "gW")         # 3.3 M-term sums in different orders


# ------------------------------------------------------------------------------------ (c) GroupNorm at cluster scale
@pytest.mark.parametrize("shape,relu,skip", [((4, 32, 48, 144, 240), True, False), ((4, 32, 48, 144, 240), True, True),
                                             ((4, 32, 48, 144, 240), False, True), ((4, 64, 24, 72, 120), True, False),
                                             ((4, 64, 24, 72, 120), True, True), ((8, 32, 1, 576, 960), True, False),
                                             ((8, 128, 1, 144, 240), False, True)])
def test_groupnorm_fwd_bwd_production_size(ecm, shape, relu, skip):
    """convbn_3d's GroupNorm(32) + the ReLU / residual that follows it (cmfsm.py:58, 287-299, 685-693), forward AND
    backward, at the geometry of the training step (clusters of 51 / 81 workgroups per span) vs F.group_norm autograd."""
    Bn, C = shape[:2]
    x = torch.randn(shape, device="cuda", generator=gen(31)) * 1.7 + 0.3
    gm = torch.rand(C, device="cuda", generator=gen(32)) + 0.5
    bt = torch.randn(C, device="cuda", generator=gen(33)) * 0.2
    sk = torch.randn(shape, device="cuda", generator=gen(34)) if skip else None
    G = torch.randn(shape, device="cuda", generator=gen(35))
    got_in = [t.clone().requires_grad_() if t is not None else None for t in (x, gm, bt, sk)]
    ref_in = [t.clone().requires_grad_() if t is not None else None for t in (x, gm, bt, sk)]
    y = ecm.ops.group_norm_act(*got_in, relu)
    ref = F.group_norm(ref_in[0], 32, ref_in[1], ref_in[2], 1e-5)
    if skip:
        ref = ref + ref_in[3]
    if relu:
        ref = F.relu(ref)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=2e-5)
    y.backward(G)
    ref.backward(G)
    gx, gx_ref = got_in[0].grad, ref_in[0].grad
    if relu:
        # where the pre-activation is within rounding of 0 the two evaluations may take different ReLU branches (1-3 of
        # 2e8 elements at this size): those positions are excluded from the comparison, everything else must agree
        tie = ref.detach() <= 2e-5
        tie &= y.detach() <= 2e-5
        pre_small = tie & ((y.detach() > 0) != (ref.detach() > 0))
        assert int(pre_small.sum()) <= 16
        gx, gx_ref = torch.where(pre_small, gx_ref, gx), gx_ref
        if skip:
            got_in[3].grad = torch.where(pre_small, ref_in[3].grad, got_in[3].grad)
    del y, ref
    torch.testing.assert_close(gx, gx_ref, rtol=1e-3, atol=1e-4)
    rel_close(got_in[1].grad, ref_in[1].grad, 1e-3, "ggamma")
    rel_close(got_in[2].grad, ref_in[2].grad, 1e-3, "gbeta")
    if skip:
        torch.testing.assert_close(got_in[3].grad, ref_in[3].grad, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------- (d) ECM weights / heads backward
def _leaky(x):
    return F.leaky_relu(x, 0.01)


def ecm_weights9_torch(lr, hr, W0, W1, W2, W3, s=4):
    """SURVEY.md 8a closed form of eight_related_context_mapping (cmfsm.py:431-593) in plain torch on the tensors' device:
    logit_n = MLP(W0[:, :32] lr[cell+n] + W0[:, 32:64] hr + W0[:, 64] offx_n + W0[:, 65] offy_n), -100 outside, softmax."""
    B, _, h, w = lr.shape
    H, Wd = h * s, w * s
    w0 = W0.view(32, 66)
    A = torch.einsum("oc,bchw->bohw", w0[:, :32], lr)
    Bv = torch.einsum("oc,bchw->bohw", w0[:, 32:64], hr)
    dev_ = lr.device
    half = torch.tensor([-2., -1., 1., 2.], device=dev_)
    dn, up = torch.tensor([4., 3., 2., 1.], device=dev_), torch.tensor([1., 2., 3., 4.], device=dev_)
    X, Y = torch.arange(Wd, device=dev_) % s, torch.arange(H, device=dev_) % s
    nb = ((0, 0, 0), (0, -1, 1), (0, 1, 2), (-1, 0, 3), (1, 0, 4), (-1, -1, 1), (-1, 1, 2), (1, -1, 3), (1, 1, 4))
    Ap = F.pad(A, (1, 1, 1, 1))
    ok = F.pad(torch.ones(1, 1, h, w, device=dev_), (1, 1, 1, 1))
    logits = []
    for dy, dx, t in nb:
        offx = (dn if t == 1 else up if t == 2 else half)[X].view(1, 1, 1, Wd)
        offy = (dn if t == 3 else up if t == 4 else half)[Y].view(1, 1, H, 1)
        a = Ap[:, :, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w].repeat_interleave(s, -1).repeat_interleave(s, -2)
        valid = ok[:, :, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w].repeat_interleave(s, -1).repeat_interleave(s, -2)
        z = _leaky(a + Bv + w0[:, 64].view(1, 32, 1, 1) * offx + w0[:, 65].view(1, 32, 1, 1) * offy)
        # 1x1 convolutions as einsums: plain GEMM-free torch arithmetic in any dtype (MIOpen has no fp64 convolution)
        z = _leaky(torch.einsum("oc,bchw->bohw", W1.view(16, 32), z))
        z = _leaky(torch.einsum("oc,bchw->bohw", W2.view(8, 16), z))
        z = torch.einsum("oc,bchw->bohw", W3.view(1, 8), z)
        logits.append(torch.where(valid > 0, z, torch.full_like(z, -100.0)))
    return F.softmax(torch.cat(logits, 1), 1)


def test_ecm_weights9_restatement_matches_kernel_small(ecm):
    """The torch closed form used below is itself checked against the kernel (which the CPU oracle pins) on a small map."""
    lr = torch.randn(2, 32, 5, 9, device="cuda", generator=gen(41))
    hr = torch.randn(2, 32, 20, 36, device="cuda", generator=gen(42))
    Ws = [torch.randn(*s, device="cuda", generator=gen(43 + i)) * 0.2
          for i, s in enumerate(((32, 66, 1, 1), (16, 32, 1, 1), (8, 16, 1, 1), (1, 8, 1, 1)))]
    torch.testing.assert_close(ecm.ops.ecm_weights9(lr, hr, *Ws), ecm_weights9_torch(lr, hr, *Ws), rtol=1e-4, atol=1e-6)


def test_ecm_weights9_backward_576x960(ecm):
    """ecm_weights9 forward + backward at 576x960 (batch 2) vs autograd of the closed form: glr, ghr and the four MLP
    weight gradients (sums over 1.1 M pixels x 9 neighbours).  The MLP has ~3e8 LeakyReLU units per pass; a few hundred sit
    within rounding of their kink, where two fp32 evaluation orders pick different slopes and ONE term of a gradient sum
    changes by 99 %.  So the yardstick is the fp64 evaluation of the same closed form: the kernel must be as close to it as
    torch's own fp32 evaluation is (same bulk tolerance, comparable number and size of kink outliers)."""
    B, h, w = 2, 144, 240
    lr = torch.randn(B, 32, h, w, device="cuda", generator=gen(51))
    hr = torch.randn(B, 32, 4 * h, 4 * w, device="cuda", generator=gen(52))
    Ws = [torch.randn(*s, device="cuda", generator=gen(53 + i)) * sc
          for i, (s, sc) in enumerate((((32, 66, 1, 1), 0.2), ((16, 32, 1, 1), 0.3), ((8, 16, 1, 1), 0.4), ((1, 8, 1, 1), 0.5)))]
    G = torch.randn(B, 9, 4 * h, 4 * w, device="cuda", generator=gen(59))
    a = [t.clone().requires_grad_() for t in (lr, hr, *Ws)]
    b = [t.clone().requires_grad_() for t in (lr, hr, *Ws)]
    w9 = ecm.ops.ecm_weights9(*a)
    ref = ecm_weights9_torch(*b)
    torch.testing.assert_close(w9, ref, rtol=1e-4, atol=1e-6)
    w9.backward(G)
    ref.backward(G)
    del w9, ref
    grads64 = []
    for bi in range(B):                                  # fp64 truth, one sample at a time (memory)
        c = [t[bi:bi + 1].double().requires_grad_() for t in (lr, hr)] + [t.double().requires_grad_() for t in Ws]
        ecm_weights9_torch(*c).backward(G[bi:bi + 1].double())
        grads64.append([t.grad for t in c])
    truth = [torch.cat([g[0] for g in grads64]), torch.cat([g[1] for g in grads64])] + \
            [sum(g[2 + i] for g in grads64) for i in range(4)]
    for i, nm in enumerate(("glr", "ghr", "gW0", "gW1", "gW2", "gW3")):
        t64 = truth[i]
        scale = float(t64.abs().max())
        e_hip = (a[i].grad.double() - t64).abs()
        e_t32 = (b[i].grad.double() - t64).abs()
        tol = (1e-4 if i < 2 else 5e-4) * scale
        n_hip, n_t32 = int((e_hip > tol).sum()), int((e_t32 > tol).sum())
        assert n_hip <= 3 * n_t32 + 64, f"{nm}: {n_hip} elements beyond tolerance vs {n_t32} for torch fp32"
        assert float(e_hip.max()) <= 3.0 * float(e_t32.max()) + tol, \
            f"{nm}: max error {float(e_hip.max()):.3e} vs torch fp32's {float(e_t32.max()):.3e} (scale {scale:.3e})"
        # the bulk: 99 % of the elements within the plain fp32 tolerance (a glr element sums ~8000 activation terms, so a
        # few tenths of a percent of them contain a kink flip)
        if e_hip.numel() > 100000:
            assert float(torch.quantile(e_hip.flatten()[:8000000].float(), 0.99)) <= tol, nm


def test_heads_backward_576x960(ecm):
    """softargmin_heads (cmfsm.py:703-706, 725-728, 748-753) and ecm_aggregate9 (709-723, x3) forward + backward at the
    production size vs autograd of the same math in plain torch."""
    B, D, h, w, s = 4, 48, 144, 240, 4
    c = torch.randn(3, B, D, h, w, device="cuda", generator=gen(61)) * 2.0
    w9 = torch.softmax(torch.randn(B, 9, h * s, w * s, device="cuda", generator=gen(62)), 1)
    G = torch.randn(3, B, h * s, w * s, device="cuda", generator=gen(63))
    a = [t.clone().requires_grad_() for t in (c, w9)]
    b = [t.clone().requires_grad_() for t in (c, w9)]
    out = ecm.ops.ecm_aggregate9(ecm.ops.softargmin_heads(a[0]), a[1], s)
    logits = torch.cumsum(b[0], 0)
    disp = (F.softmax(logits, 2) * torch.arange(D, device="cuda", dtype=torch.float32).view(1, 1, D, 1, 1)).sum(2)
    pred = F.pad(s * disp, (1, 1, 1, 1))
    nb = ((0, 0), (0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1))
    ref = 0
    for n, (dy, dx) in enumerate(nb):
        shifted = pred[..., 1 + dy:1 + dy + h, 1 + dx:1 + dx + w].repeat_interleave(s, -1).repeat_interleave(s, -2)
        ref = ref + shifted * b[1][:, n].unsqueeze(0)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-3)          # values up to 4*47 px
    out.backward(G)
    ref.backward(G)
    rel_close(a[0].grad, b[0].grad, 2e-4, "gc")
    rel_close(a[1].grad, b[1].grad, 2e-5, "gw9")


# ------------------------------------------------------------------------------ (e) one batch-4 training step, both forms
@pytest.mark.parametrize("B,H,W", [(4, 576, 960), (1, 384, 1248)])
def test_train_step_b4_collapsed_vs_explicit(ecm, B, H, W):
    """One cmfsm training step (train.py:162-181) at BASELINE cfg 2 -- batch 4, 576x960, D=192 -- and at the KITTI frame of
    cfg 4 (375x1242 padded to 384x1248, one pair per GPU: 312 / 156 / 78 columns at the three scales of the aggregation
    stack, none a multiple of a tile width) -- with the cost volume collapsed into 2-D convolutions (default) and with the
    reference's explicit op sequence: same loss, same gradient for EVERY parameter (whole-tensor bound)."""
    from importlib import import_module
    mdl = import_module("explicit-context-mapping-for-stereo-matching_amd.models")
    dist = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    torch.manual_seed(0)
    model = ecm.get_model("cmfsm").cuda().train()
    g = torch.Generator(device="cpu").manual_seed(1234)
    left, right = torch.randn(B, 3, H, W, generator=g).cuda(), torch.randn(B, 3, H, W, generator=g).cuda()
    gt = (torch.rand(B, H, W, generator=g) * 191.0).cuda()
    res = {}
    prev = mdl.EXPLICIT_COST_VOLUME
    try:
        for tag, flag in (("collapsed", False), ("explicit", True)):
            mdl.EXPLICIT_COST_VOLUME = flag
            model.zero_grad(set_to_none=True)
            loss = dist.masked_smooth_l1_x3(model(left, right), gt, 192)
            loss.backward()
            torch.cuda.synchronize()
            res[tag] = (float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()})
            model.zero_grad(set_to_none=True)
    finally:
        mdl.EXPLICIT_COST_VOLUME = prev
    (la, ga), (lb, gb) = res["collapsed"], res["explicit"]
    assert la == la and abs(la - lb) <= 1e-4 * abs(lb), (la, lb)
    worst = max(((float((ga[k] - gb[k]).abs().max()) / (float(gb[k].abs().max()) + 1e-30)), k) for k in ga)
    worst_l2 = max(((float((ga[k] - gb[k]).norm()) / (float(gb[k].norm()) + 1e-30)), k) for k in ga)
    # everything after dres0.0 is the same kernels on inputs that differ by fp32 rounding of the first conv (~1e-6
    # relative), which kink flips of ReLU / LeakyReLU / smooth-L1 amplify to the per-mille level in single gradient
    # elements (the fp64 yardstick of the same effect: tests/test_hip_fp64_yardstick.py).  The largest single element is
    # a chaotic quantity (it moved from 1.6e-2 to 2.2e-2 at 384x1248 when the GroupNorm pivot rule changed in round 4,
    # with both sides of this comparison on the new rule), so it is bounded loosely and the per-tensor L2 distance tightly.
    assert worst[0] <= 5e-2, f"largest relative gradient difference {worst}"
    assert worst_l2[0] <= 1e-2, f"largest relative L2 gradient difference {worst_l2}"
    for k in ("dres0.0.0.weight", "dres0.0.1.weight", "dres1.0.0.weight", "mapping_matrix.similarity1.conv0.weight"):
        rel_close(ga[k], gb[k], 5e-3, k)
