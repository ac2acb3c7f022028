"""The heads of the other registered architectures at PRODUCTION size (576x960, D=192):
  a10 volume mapping   (cmfsm_sub_16.py:767-848, cm_sub_8.py:765-800)           ops.volume_mapping
  a11 trilinear head   (bilinear_cmf.py:447-471; test.py:111's default arch)     ops.trilinear_softargmin
  a4  six-related context weights (cmfsm_sub_8.py:440-572)                       ops.context_weights variants 1 / 2
Forward against device-side torch restatements of the reference's op sequence (each first checked against the kernel at a
small size, where the CPU oracle pins the kernel: tests/test_hip_parity.py); backward against autograd of the same
restatement with its fp64 evaluation as the yardstick (the kernel must be as close to fp64 as torch's own fp32 evaluation
is); and bit-identical gradients from run to run -- the backward kernels hold no float atomics any more."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


def gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


# ------------------------------------------------------------------------------------------------ restatements
def soft_argmin_t(cost):
    D = cost.shape[1]
    return (F.softmax(cost, 1) * torch.arange(D, device=cost.device, dtype=cost.dtype).view(1, D, 1, 1)).sum(1)


def volume_mapping_t(cost_lr, m5, mt3, s):
    """cmfsm_sub_16.py:767-801 on [B,Dl,h,w] logits (already accumulated over heads): vectorised, any dtype / device.
    The spatial 5-neighbour fuse (:774-778) does not mix depths, so it is done before the depth upsample (:768-773)."""
    B, Dl, h, w = cost_lr.shape
    H, W, D = h * s, w * s, Dl * s
    up = cost_lr.repeat_interleave(s, -1).repeat_interleave(s, -2)                       # [B,Dl,H,W]
    fused = up * m5[:, 0:1]
    for n, (dy, dx) in enumerate(((0, 0), (0, 1), (0, -1), (-1, 0), (1, 0))):           # c, r, l, t, b
        if n == 0:
            continue
        sh = torch.roll(up, shifts=(-dy * s, -dx * s), dims=(-2, -1))                    # value of the neighbour cell
        ok = torch.ones(1, 1, H, W, device=up.device, dtype=up.dtype)
        if dy < 0: ok[:, :, :s] = 0
        if dy > 0: ok[:, :, H - s:] = 0
        if dx < 0: ok[..., :s] = 0
        if dx > 0: ok[..., W - s:] = 0
        fused = fused + sh * ok * m5[:, n:n + 1]
    fused = fused.repeat_interleave(s, 1)                                                 # [B,D,H,W]
    X = torch.arange(W, device=up.device).view(1, 1, 1, W)
    Dv = torch.arange(D, device=up.device).view(1, D, 1, 1)
    idx = (X - Dv).expand(B, D, H, W)
    inside = idx >= 0

    def target(pl):                                                                       # :782-794, ones where x < d
        return torch.where(inside, torch.gather(mt3[:, pl:pl + 1].expand(B, D, H, W), 3, idx.clamp(min=0)),
                           torch.ones((), device=up.device, dtype=up.dtype))
    out = fused * target(0)                                                               # :796
    out = out + F.pad(fused[:, s:] * target(2)[:, :-s], (0, 0, 0, 0, 0, s))               # :797
    out = out + F.pad(fused[:, :-s] * target(1)[:, s:], (0, 0, 0, 0, s, 0))               # :798
    return soft_argmin_t(out)


def trilinear_t(cost_lr, Do, H, W):
    up = F.interpolate(cost_lr.unsqueeze(1), [Do, H, W], mode="trilinear", align_corners=False).squeeze(1)
    return soft_argmin_t(up)


def _leaky(x):
    return F.leaky_relu(x, 0.01)


def six_planes_t(lr, hr, W0, W1, W2, W3, target):
    """six_related_context_mapping (cmfsm_sub_8.py:449-572): reference image -> planes [c, r, l, t, b] with tables
    0,1,2,3,4; target image -> [c, r, l]; zero padding that stays in the softmax; extra LeakyReLU; softmax * logit."""
    B, _, h, w = lr.shape
    H, Wd = hr.shape[-2:]
    s = Wd // w
    dev_, dt = lr.device, lr.dtype
    w0 = W0.view(32, 66)
    A = torch.einsum("oc,bchw->bohw", w0[:, :32], lr)
    Bv = torch.einsum("oc,bchw->bohw", w0[:, 32:64], hr)
    r = torch.arange(s, device=dev_)
    half = torch.where(r < s // 2, r - s // 2, r - s // 2 + 1).to(dt)
    dn, up = (s - r).to(dt), (r + 1).to(dt)
    X, Y = torch.arange(Wd, device=dev_) % s, torch.arange(H, device=dev_) % s
    nb = ((0, 0, 0), (0, 1, 1), (0, -1, 2)) if target else ((0, 0, 0), (0, 1, 1), (0, -1, 2), (-1, 0, 3), (1, 0, 4))
    Ap = F.pad(A, (1, 1, 1, 1))
    ok = F.pad(torch.ones(1, 1, h, w, device=dev_, dtype=dt), (1, 1, 1, 1))
    logits = []
    for dy, dx, t in nb:
        offx = (dn if t == 1 else up if t == 2 else half)[X].view(1, 1, 1, Wd)
        offy = (dn if t == 3 else up if t == 4 else half)[Y].view(1, 1, H, 1)
        a = Ap[:, :, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w].repeat_interleave(s, -1).repeat_interleave(s, -2)
        valid = ok[:, :, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w].repeat_interleave(s, -1).repeat_interleave(s, -2)
        z = _leaky(a + Bv + w0[:, 64].view(1, 32, 1, 1) * offx + w0[:, 65].view(1, 32, 1, 1) * offy)
        z = _leaky(torch.einsum("oc,bchw->bohw", W1.view(16, 32), z))
        z = _leaky(torch.einsum("oc,bchw->bohw", W2.view(8, 16), z))
        z = _leaky(torch.einsum("oc,bchw->bohw", W3.view(1, 8), z))
        logits.append(z * valid)
    allp = torch.cat(logits, 1)
    return F.softmax(allp, 1) * allp


# ------------------------------------------------------------------------------------------------ yardstick helper
def yardstick(name, hip, t32, t64, tol_rel):
    scale = float(t64.abs().max())
    e_hip, e_t32 = (hip.double() - t64).abs(), (t32.double() - t64).abs()
    tol = tol_rel * scale
    n_hip, n_t32 = int((e_hip > tol).sum()), int((e_t32 > tol).sum())
    assert n_hip <= 3 * n_t32 + 64, f"{name}: {n_hip} elements beyond {tol:.2e} vs {n_t32} for torch fp32"
    assert float(e_hip.max()) <= 3.0 * float(e_t32.max()) + tol, \
        f"{name}: max error {float(e_hip.max()):.3e} vs torch fp32's {float(e_t32.max()):.3e} (scale {scale:.3e})"


def _heads_inputs(NH, B, Dl, h, w, s, seed):
    g = gen(seed)
    c = torch.randn(NH, B, Dl, h, w, device="cuda", generator=g) * 1.5
    m5 = torch.randn(B, 5, h * s, w * s, device="cuda", generator=g) * 0.5
    mt3 = torch.randn(B, 3, h * s, w * s, device="cuda", generator=g) * 0.5
    G = torch.randn(NH, B, h * s, w * s, device="cuda", generator=g)
    return c, m5, mt3, G


def test_restatements_match_the_kernels_small(ecm):
    c, m5, mt3, _ = _heads_inputs(2, 2, 6, 5, 7, 4, 1)
    out = ecm.ops.volume_mapping(c, m5, mt3, 4)
    ref = torch.stack([volume_mapping_t(c[:k + 1].sum(0), m5, mt3, 4) for k in range(2)], 0)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-4)
    out = ecm.ops.trilinear_softargmin(c, 24, 20, 28)
    ref = torch.stack([trilinear_t(c[:k + 1].sum(0), 24, 20, 28) for k in range(2)], 0)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-4)
    g = gen(2)
    lr, hr = torch.randn(2, 32, 3, 5, device="cuda", generator=g), torch.randn(2, 32, 24, 40, device="cuda", generator=g)
    Ws = [torch.randn(*sh, device="cuda", generator=g) * 0.2 for sh in ((32, 66, 1, 1), (16, 32, 1, 1), (8, 16, 1, 1), (1, 8, 1, 1))]
    for variant in (1, 2):
        torch.testing.assert_close(ecm.ops.context_weights(lr, hr, *Ws, variant), six_planes_t(lr, hr, *Ws, variant == 2),
                                   rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("arch,NH,s", [("cmfsm_sub_16", 3, 16), ("cm_sub_8", 1, 8)])
def test_volume_mapping_576x960(ecm, arch, NH, s):
    B, H, W = 1, 576, 960
    h, w, Dl = H // s, W // s, 192 // s
    c, m5, mt3, G = _heads_inputs(NH, B, Dl, h, w, s, 11)
    a = [t.clone().requires_grad_() for t in (c, m5, mt3)]
    out = ecm.ops.volume_mapping(*a, s)
    b = [t.clone().requires_grad_() for t in (c, m5, mt3)]
    ref = torch.stack([volume_mapping_t(b[0][:k + 1].sum(0), b[1], b[2], s) for k in range(NH)], 0)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-4)                # disparities 0..191 px
    out.backward(G)
    ref.backward(G)
    del ref
    d = [t.double().requires_grad_() for t in (c, m5, mt3)]
    torch.stack([volume_mapping_t(d[0][:k + 1].sum(0), d[1], d[2], s) for k in range(NH)], 0).backward(G.double())
    for i, nm in enumerate(("g_logits", "g_m5", "g_mt3")):
        yardstick(f"{arch} {nm}", a[i].grad, b[i].grad, d[i].grad, 1e-4)
    # deterministic: a second backward over the same inputs gives the same bits
    a2 = [t.clone().requires_grad_() for t in (c, m5, mt3)]
    ecm.ops.volume_mapping(*a2, s).backward(G)
    for x, y in zip(a, a2):
        assert torch.equal(x.grad, y.grad)


@pytest.mark.parametrize("arch,NH,s", [("bilinear_cmf_sub_16", 3, 16), ("bilinear_cmf", 3, 4)])
def test_trilinear_head_576x960(ecm, arch, NH, s):
    B, H, W = 1, 576, 960
    h, w, Dl = H // s, W // s, 192 // s
    c, _, _, G = _heads_inputs(NH, B, Dl, h, w, s, 13)
    a = c.clone().requires_grad_()
    out = ecm.ops.trilinear_softargmin(a, 192, H, W)
    b = c.clone().requires_grad_()
    ref = torch.stack([trilinear_t(b[:k + 1].sum(0), 192, H, W) for k in range(NH)], 0)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-4)
    out.backward(G)
    ref.backward(G)
    del ref
    d = c.double().requires_grad_()
    torch.stack([trilinear_t(d[:k + 1].sum(0), 192, H, W) for k in range(NH)], 0).backward(G.double())
    yardstick(f"{arch} g_logits", a.grad, b.grad, d.grad, 1e-4)
    a2 = c.clone().requires_grad_()
    ecm.ops.trilinear_softargmin(a2, 192, H, W).backward(G)
    assert torch.equal(a.grad, a2.grad)


@pytest.mark.parametrize("s,variant", [(8, 1), (8, 2), (16, 1), (16, 2)])
def test_context_weights_576x960(ecm, s, variant):
    B, H, W = 1, 576, 960
    g = gen(17 + s + variant)
    lr = torch.randn(B, 32, H // s, W // s, device="cuda", generator=g)
    hr = torch.randn(B, 32, H, W, device="cuda", generator=g)
    Ws = [torch.randn(*sh, device="cuda", generator=g) * sc for sh, sc in
          (((32, 66, 1, 1), 0.1), ((16, 32, 1, 1), 0.3), ((8, 16, 1, 1), 0.4), ((1, 8, 1, 1), 0.5))]
    a = [t.clone().requires_grad_() for t in (lr, hr, *Ws)]
    b = [t.clone().requires_grad_() for t in (lr, hr, *Ws)]
    out = ecm.ops.context_weights(*a, variant)
    ref = six_planes_t(*b, variant == 2)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)
    G = torch.randn(ref.shape, device="cuda", generator=g)
    out.backward(G)
    ref.backward(G)
    del ref
    d = [t.double().requires_grad_() for t in (lr, hr, *Ws)]
    six_planes_t(*d, variant == 2).backward(G.double())
    for i, nm in enumerate(("glr", "ghr", "gW0", "gW1", "gW2", "gW3")):
        yardstick(f"variant {variant} s {s} {nm}", a[i].grad, b[i].grad, d[i].grad, 1e-4 if i < 2 else 5e-4)
    a2 = [t.clone().requires_grad_() for t in (lr, hr, *Ws)]
    ecm.ops.context_weights(*a2, variant).backward(G)
    for x, y in zip(a, a2):
        assert torch.equal(x.grad, y.grad)
