"""Full-size checks on the MI355X at BASELINE.json's shapes (SceneFlow 576x960 and KITTI 384x1248, D=192), where the CPU
oracle would take minutes: size-independent properties and cross-checks against independent device-side arithmetic
(plain torch indexing / MIOpen through torch), never against the oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = {"sceneflow": (144, 240), "kitti": (96, 312)}


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


@pytest.mark.parametrize("name", list(SHAPES))
def test_cost_volume_full_size_exact(ecm, name):
    h, w = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(1)
    L = torch.randn(2, 32, h, w, device="cuda", generator=g)
    R = torch.randn(2, 32, h, w, device="cuda", generator=g)
    cost = ecm.ops.cost_volume(L, R, 48)
    assert cost.shape == (2, 64, 48, h, w)
    for d in (0, 1, 7, 31, 47):                                   # slices equal plain shifted copies, bit for bit
        assert torch.equal(cost[:, :32, d, :, d:], L[..., d:])
        assert torch.equal(cost[:, 32:, d, :, d:], R[..., :w - d])
        assert not cost[:, :, d, :, :d].any()
    # linearity + adjoint identity:  <costvol(L,R), G> == <L, gL> + <R, gR>
    G = torch.randn_like(cost)
    Lg, Rg = L.clone().requires_grad_(), R.clone().requires_grad_()
    ecm.ops.cost_volume(Lg, Rg, 48).backward(G)
    lhs = (cost.double() * G.double()).sum()
    rhs = (L.double() * Lg.grad.double()).sum() + (R.double() * Rg.grad.double()).sum()
    # 4.7e8 fp32 terms on the left, gradients accumulated in fp32 on the right: rounding alone is ~ sqrt(N) * eps ~ 1e-3
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs) + 5e-3


@pytest.mark.parametrize("name,ci,co,stride", [("sceneflow", 32, 32, 1), ("sceneflow", 32, 64, 2), ("sceneflow", 32, 1, 1),
                                               ("kitti", 64, 32, 1)])
def test_conv3d_full_size_vs_miopen(ecm, name, ci, co, stride):
    """The MFMA conv (fwd, dgrad, wgrad) at the real volume sizes against MIOpen's conv3d reached through torch."""
    h, w = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(1, ci, 48, h, w, device="cuda", generator=g)
    wt = torch.randn(co, ci, 3, 3, 3, device="cuda", generator=g) * (2.0 / (27 * ci)) ** 0.5
    xg, wg = x.clone().requires_grad_(), wt.clone().requires_grad_()
    y = ecm.ops.conv3d_k3(xg, wg, stride)
    xr, wr = x.clone().requires_grad_(), wt.clone().requires_grad_()
    ref = F.conv3d(xr, wr, None, stride, 1)
    torch.testing.assert_close(y, ref, rtol=1e-3, atol=1e-4)
    G = torch.randn_like(ref)
    y.backward(G); ref.backward(G)
    torch.testing.assert_close(xg.grad, xr.grad, rtol=1e-3, atol=1e-4)
    scale = float(wr.grad.abs().max())
    torch.testing.assert_close(wg.grad, wr.grad, rtol=2e-3, atol=2e-4 * scale)


def test_deconv3d_full_size_vs_miopen(ecm):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(1, 64, 24, 72, 120, device="cuda", generator=g)
    wt = torch.randn(64, 32, 3, 3, 3, device="cuda", generator=g) * 0.03
    xg, wg = x.clone().requires_grad_(), wt.clone().requires_grad_()
    y = ecm.ops.deconv3d_k3s2(xg, wg)
    xr, wr = x.clone().requires_grad_(), wt.clone().requires_grad_()
    ref = F.conv_transpose3d(xr, wr, None, 2, 1, 1)
    torch.testing.assert_close(y, ref, rtol=1e-3, atol=1e-4)
    G = torch.randn_like(ref)
    y.backward(G); ref.backward(G)
    torch.testing.assert_close(xg.grad, xr.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(wg.grad, wr.grad, rtol=2e-3, atol=2e-4 * float(wr.grad.abs().max()))


def test_groupnorm_full_size_vs_torch(ecm):
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(2, 32, 48, 144, 240, device="cuda", generator=g) * 1.5 + 0.2
    gm, bt = torch.rand(32, device="cuda", generator=g) + 0.5, torch.randn(32, device="cuda", generator=g) * 0.1
    y = ecm.ops.group_norm_act(x, gm, bt, None, True)
    ref = F.relu(F.group_norm(x, 32, gm, bt, 1e-5))
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", list(SHAPES))
def test_ecm_weights_full_size_properties(ecm, name):
    """Softmax planes sum to 1; out-of-image neighbours get (numerically) zero weight; aggregation with these weights
    of a constant disparity field returns scale*constant everywhere (partition of unity)."""
    h, w = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(5)
    lr = torch.randn(1, 32, h, w, device="cuda", generator=g)
    hr = torch.randn(1, 32, 4 * h, 4 * w, device="cuda", generator=g)
    Ws = [torch.randn(*s, device="cuda", generator=g) * 0.2 for s in ((32, 66, 1, 1), (16, 32, 1, 1), (8, 16, 1, 1), (1, 8, 1, 1))]
    w9 = ecm.ops.ecm_weights9(lr, hr, *Ws)
    torch.testing.assert_close(w9.sum(1), torch.ones(1, 4 * h, 4 * w, device="cuda"), rtol=1e-5, atol=1e-5)
    assert float(w9[:, 1, :, :4].max()) < 1e-30 and float(w9[:, 3, :4, :].max()) < 1e-30      # left / top neighbours outside
    d = torch.full((3, 1, h, w), 7.0, device="cuda")
    out = ecm.ops.ecm_aggregate9(d, w9, 4)
    torch.testing.assert_close(out, torch.full_like(out, 28.0), rtol=1e-5, atol=1e-4)


def test_full_forward_kitti_shape_is_finite_and_deterministic(ecm):
    model = ecm.get_model("cmfsm").cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(6)
    left = torch.randn(1, 3, 384, 1248, device="cuda", generator=g)
    right = torch.randn(1, 3, 384, 1248, device="cuda", generator=g)
    with torch.no_grad():
        out = model(left, right)
        lr_l, _, hr_l = model.feature_extraction(left)
        lr_r, _, _ = model.feature_extraction(right)
        a = model.hot_path(lr_l, hr_l, lr_r)
        b = model.hot_path(lr_l, hr_l, lr_r)
    for p in out:
        assert p.shape == (1, 1, 384, 1248) and torch.isfinite(p).all()
        assert float(p.min()) >= -1e-3 and float(p.max()) <= 4 * 47 + 1e-3      # convex combinations of 4*[0,47]
    for p, q in zip(a, b):
        assert torch.equal(p, q)       # every forward HIP kernel is deterministic


def test_graphed_forward_matches_eager(ecm):
    """The eval forward captured into one HIP graph (dist.GraphedForward) replays to the eager result on new inputs.
    Every kernel of the forward is this library's and deterministic: bit-identical, for the hot path alone and (below)
    for the whole model."""
    import importlib
    dist = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    torch.manual_seed(3)
    model = ecm.get_model("cmfsm").cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(11)

    class Hot(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, lr_pair, hr_l):            # lr_pair = [lr_l ; lr_r] stacked on the batch axis
            return self.m.hot_path(lr_pair[:1], hr_l, lr_pair[1:])

    def feats():
        return (torch.randn(2, 32, 64, 128, device="cuda", generator=g), torch.randn(1, 32, 256, 512, device="cuda", generator=g))

    hot = Hot(model)
    gh = dist.GraphedForward(hot, *feats())
    a, b = feats()
    got = [t.clone() for t in gh(a, b)]
    # a captured GroupNorm runs the two-stage kernels (csrc/gn3d.hip: GnControl): compare like against like, bit for bit ...
    old = ecm.ops.gn_cluster_mode(0)
    try:
        with torch.no_grad():
            want = hot(a, b)
    finally:
        ecm.ops.gn_cluster_mode(old)
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    with torch.no_grad():            # ... and against the default (cluster) kernels at the end-to-end tolerance
        want = hot(a, b)
    for x, y in zip(got, want):
        assert float((x - y).abs().max()) <= 2e-2 and float((x - y).abs().mean()) <= 1e-3
    # a graph that will be replayed alone may keep the cluster kernels (ecm_gn3d_cluster_mode 4 during capture): then it
    # equals the eager default bit for bit, and the process-wide mode is back to what it was
    before = ecm.ops.gn_cluster_mode(-1)
    gc = dist.GraphedForward(hot, *feats(), cluster_groupnorm=True)
    assert ecm.ops.gn_cluster_mode(-1) == before
    got = [t.clone() for t in gc(a, b)]
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    ecm.ops.check_async_errors()

    l0, r0 = (torch.randn(1, 3, 256, 512, device="cuda", generator=g) for _ in range(2))
    gf = dist.GraphedForward(model, l0, r0)
    l1, r1 = (torch.randn(1, 3, 256, 512, device="cuda", generator=g) for _ in range(2))
    got = [t.clone() for t in gf(l1, r1)]
    with torch.no_grad():
        want = model(l1, r1)
    for x, y in zip(got, want):      # whole model: ATen's tiny SPP matmuls may take another rocBLAS path under capture
        assert torch.isfinite(x).all() and float((x - y).abs().max()) <= 2e-2 and float((x - y).abs().mean()) <= 1e-3


def test_training_loop_reduces_the_loss(ecm):
    """The harness end to end (train.py:148-178 as bench.py runs it): FlatBucketDDP zero_grad -> forward -> fused loss ->
    backward -> fused Adam on ONE fixed synthetic pair must drive the loss down (an integration check of every backward
    kernel together; their individual gradients are checked against the oracle in test_hip_parity.py)."""
    import importlib
    dist = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    torch.manual_seed(0)
    model = ecm.get_model("cmfsm").cuda().train()
    g = torch.Generator(device="cuda").manual_seed(21)
    left = torch.randn(1, 3, 256, 512, device="cuda", generator=g)
    right = torch.randn(1, 3, 256, 512, device="cuda", generator=g)
    gt = torch.rand(1, 256, 512, device="cuda", generator=g) * 60.0 + 20.0
    ddp = dist.FlatBucketDDP(model, 1)
    opt = torch.optim.Adam(ddp.params, lr=1e-3, betas=(0.9, 0.999), fused=True)
    losses = []
    for _ in range(30):
        ddp.zero_grad()
        loss = dist.masked_smooth_l1_x3(model(left, right), gt, 192)
        loss.backward()
        ddp.allreduce_gradients()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(l == l and l < 1e6 for l in losses), losses
    assert sum(losses[-5:]) / 5 < 0.8 * (sum(losses[:3]) / 3), losses

@pytest.mark.parametrize("shape,co", [((1, 32, 48, 96, 312), 64), ((1, 32, 48, 144, 240), 32), ((8, 128, 144, 240), 128)])
def test_winograd_conv_is_reproducible_at_full_size(ecm, shape, co):
    """The Winograd kernel hands LDS buffers between waves with raw barriers and hand-counted waits; a missing wait shows
    up as a rare wrong tile.  Same input, ten launches: bit-identical outputs (and equal to the first within the parity
    tolerance of the direct kernel, checked once)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(*shape, device="cuda", generator=g)
    ci = shape[1]
    three = len(shape) == 5
    w = torch.randn(co, ci, *((3, 3, 3) if three else (3, 3)), device="cuda", generator=g) * (2.0 / ((27 if three else 9) * ci)) ** 0.5
    run = (lambda: ecm.ops.conv3d_k3(x, w, 1)) if three else (lambda: ecm.ops.conv2d(x, w, 1, 1))
    with torch.no_grad():
        first = run()
        for _ in range(9):
            assert torch.equal(run(), first)
        prev = ecm.ops.WINOGRAD
        try:
            ecm.ops.WINOGRAD = False
            direct = run()
        finally:
            ecm.ops.WINOGRAD = prev
    torch.testing.assert_close(first, direct, rtol=1e-3, atol=1e-4)
