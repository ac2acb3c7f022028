"""GPU parity tests: every HIP kernel (through the C ABI, via ecm_amd.ops) against the CPU oracle and the
golden vectors generated from the reference.  Run on the MI355X box:  pytest -m gpu

Tolerances (fp32): cost volume bit-exact; per-stage rtol 1e-4 / atol 1e-5 (reductions re-ordered);
end-to-end disparity max-abs <= 2e-2 px, mean-abs <= 1e-3 px (the oracle's own 1-vs-8-thread
noise is 2e-3 px max, fp32-vs-fp64 1.5e-2 px max).
"""
import importlib

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import ecm_oracle as O
from oracle.weights import seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import ecm_amd
    return ecm_amd


def dev(t):
    return t.cuda()


def close(a, b, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), rtol=rtol, atol=atol)


def close_grad(a, b, rel=1e-4):
    """Gradients that are SUMS over many pixels (summation order differs between the kernel and autograd): the stated
    per-stage rtol 1e-4, with the absolute floor tied to the tensor's scale -- 1e-4 of its largest element (an element near
    a zero crossing of a sum of O(1) terms cannot be held to 1e-4 of ITSELF by any fp32 evaluation)."""
    a, b = a.detach().cpu(), b.detach().cpu()
    torch.testing.assert_close(a, b, rtol=rel, atol=rel * float(b.abs().max()) + 1e-7)


# ------------------------------------------------------------------ a1
@pytest.mark.parametrize("B,C,h,w,D", [(1, 4, 5, 12, 6), (2, 8, 4, 24, 20), (1, 32, 8, 12, 48), (2, 32, 16, 60, 48),
                                       (1, 3, 7, 13, 5), (1, 2, 3, 1100, 9), (1, 2, 40, 64, 64)])
def test_costvol_fwd_bwd(ecm, B, C, h, w, D):
    L, R = seeded("cv.L", B, C, h, w), seeded("cv.R", B, C, h, w)
    Lg, Rg = dev(L).requires_grad_(), dev(R).requires_grad_()
    cost = ecm.ops.cost_volume(Lg, Rg, D)
    ref = O.cost_volume(L, R, D)
    assert torch.equal(cost.cpu(), ref)                       # bit-exact: it is a copy
    G = seeded("cv.G", *ref.shape)
    cost.backward(dev(G))
    Lc, Rc = L.clone().requires_grad_(), R.clone().requires_grad_()
    O.cost_volume(Lc, Rc, D).backward(G)
    close(Lg.grad, Lc.grad, 1e-5, 1e-5)
    close(Rg.grad, Rc.grad, 1e-5, 1e-5)


def test_costvol_golden(ecm):
    for tag, (B, C, h, w, D) in {"a": (1, 4, 5, 12, 6), "b": (2, 8, 4, 24, 20)}.items():
        g = load_golden(f"g1{tag}_costvol")
        L, R = seeded(f"g1{tag}.L", B, C, h, w), seeded(f"g1{tag}.R", B, C, h, w)
        assert torch.equal(ecm.ops.cost_volume(dev(L), dev(R), D).cpu(), g["cost"])
    g = load_golden("g7a_cost")
    assert torch.equal(ecm.ops.cost_volume(dev(seeded("g7a.lr_l", 1, 32, 8, 12)), dev(seeded("g7a.lr_r", 1, 32, 8, 12)),
                                           48).cpu(), g["cost"])


def test_matchshifted_module(ecm):
    L, R = seeded("ms.L", 1, 4, 5, 12), seeded("ms.R", 1, 4, 5, 12)
    for shift in (0, 3, 7):
        out = ecm.matchshifted()(dev(L), dev(R), shift)
        assert torch.equal(out.cpu(), O.matchshifted(L, R, shift))


# ------------------------------------------------------------------ a8
def test_softargmin_golden(ecm):
    g = load_golden("g4_softargmin")
    cost = dev(seeded("g4.cost", 2, 48, 6, 10, scale=2.0)).requires_grad_()
    d = ecm.ops.softargmin_heads(cost.unsqueeze(0))[0]
    close(d, g["disp"], 1e-5, 1e-5)
    (d * dev(seeded("g4.G", 2, 6, 10))).sum().backward()
    close(cost.grad, g["g_cost"], 1e-4, 1e-5)
    p = F.softmax(seeded("g4.cost", 2, 48, 6, 10, scale=2.0), 1)
    close(ecm.disparityregression(48)(dev(p)), O.soft_argmin(seeded("g4.cost", 2, 48, 6, 10, scale=2.0)), 1e-5, 1e-5)


@pytest.mark.parametrize("NH", [1, 2, 3])
def test_softargmin_heads(ecm, NH):
    c = seeded("sa.c", NH, 2, 48, 5, 9, scale=1.5)
    cg = dev(c).requires_grad_()
    d = ecm.ops.softargmin_heads(cg)
    cc = c.clone().requires_grad_()
    ref = torch.stack([O.soft_argmin(cc[:k + 1].sum(0)) for k in range(NH)], 0)
    close(d, ref, 1e-5, 1e-5)
    G = seeded("sa.G", NH, 2, 5, 9)
    d.backward(dev(G)); ref.backward(G)
    close(cg.grad, cc.grad, 1e-4, 1e-5)


# ------------------------------------------------------------------ a9
@pytest.mark.parametrize("B,h,w,s,NH", [(1, 3, 4, 4, 1), (2, 5, 7, 4, 3), (1, 4, 6, 2, 2), (1, 2, 3, 8, 3)])
def test_aggregate9(ecm, B, h, w, s, NH):
    d = seeded("ag.d", NH, B, h, w, scale=10.0)
    w9 = torch.softmax(seeded("ag.w", B, 9, h * s, w * s), 1)
    dg, wg = dev(d).requires_grad_(), dev(w9).requires_grad_()
    out = ecm.ops.ecm_aggregate9(dg, wg, s)
    dc, wc = d.clone().requires_grad_(), w9.clone().requires_grad_()
    ref = torch.stack([O.ecm_aggregate_eight(dc[k], wc, s)[:, 0] for k in range(NH)], 0)
    close(out, ref, 1e-5, 1e-5)
    G = seeded("ag.G", *ref.shape)
    out.backward(dev(G)); ref.backward(G)
    close(dg.grad, dc.grad, 1e-4, 1e-4)
    close(wg.grad, wc.grad, 1e-5, 1e-5)


# ------------------------------------------------------------------ a3
def _mlp(sd):
    return [sd[f"mapping_matrix.similarity1.conv{i}.weight"] for i in range(4)]


def test_ecm_weights_golden(ecm, cmfsm_sd):
    g = load_golden("g2_ecm_weights")
    lr, hr = seeded("g2.lr", 1, 32, 3, 4), seeded("g2.hr", 1, 32, 12, 16)
    w9 = ecm.ops.ecm_weights9(dev(lr), dev(hr), *[dev(t) for t in _mlp(cmfsm_sd)])
    close(w9, g["w9"], 1e-4, 1e-6)


@pytest.mark.parametrize("B,h,w,s", [(1, 8, 12, 4), (2, 5, 33, 4), (1, 3, 5, 2), (1, 4, 20, 8)])
def test_ecm_weights_vs_oracle(ecm, cmfsm_sd, B, h, w, s):
    if s != 4:
        # the reference's tables are hard-coded for scale 4 (quirk Q3) and its forward raises a shape error otherwise:
        # the eight-related kernel refuses such a call instead of computing something nothing can be compared with
        lr, hr = seeded("ew.lr", B, 32, h, w), seeded("ew.hr", B, 32, h * s, w * s)
        with pytest.raises(RuntimeError, match="scale 4 only"):
            ecm.ops.ecm_weights9(dev(lr), dev(hr), *[dev(t) for t in _mlp(cmfsm_sd)])
        return
    lr, hr = seeded("ew.lr", B, 32, h, w), seeded("ew.hr", B, 32, h * s, w * s)
    w9 = ecm.ops.ecm_weights9(dev(lr), dev(hr), *[dev(t) for t in _mlp(cmfsm_sd)])
    close(w9, O.ecm_weights_eight(lr, hr, cmfsm_sd), 1e-4, 1e-6)
    close(w9.sum(1), torch.ones(B, h * s, w * s), 1e-5, 1e-5)


def test_ecm_weights_bwd_golden(ecm, cmfsm_sd):
    g = load_golden("g2_ecm_weights")
    lr, hr = dev(seeded("g2.lr", 1, 32, 3, 4)).requires_grad_(), dev(seeded("g2.hr", 1, 32, 12, 16)).requires_grad_()
    Ws = [dev(t).requires_grad_() for t in _mlp(cmfsm_sd)]
    w9 = ecm.ops.ecm_weights9(lr, hr, *Ws)
    (w9 * dev(seeded("g2.G", 1, 9, 12, 16))).sum().backward()
    close_grad(lr.grad, g["g_lr"])
    close_grad(hr.grad, g["g_hr"])
    for i in range(4):
        close_grad(Ws[i].grad, g[f"g_similarity1_conv{i}_weight"])


@pytest.mark.parametrize("B,h,w", [(1, 8, 12), (2, 5, 33), (1, 3, 16)])
def test_ecm_weights_bwd_vs_oracle(ecm, cmfsm_sd, B, h, w):
    lr, hr = seeded("ewb.lr", B, 32, h, w), seeded("ewb.hr", B, 32, 4 * h, 4 * w)
    G = seeded("ewb.G", B, 9, 4 * h, 4 * w)
    lg, hg = dev(lr).requires_grad_(), dev(hr).requires_grad_()
    Ws = [dev(t).requires_grad_() for t in _mlp(cmfsm_sd)]
    (ecm.ops.ecm_weights9(lg, hg, *Ws) * dev(G)).sum().backward()
    sd = {k: v.clone().requires_grad_() for k, v in cmfsm_sd.items() if k.startswith("mapping_matrix")}
    lc, hc = lr.clone().requires_grad_(), hr.clone().requires_grad_()
    (O.ecm_weights_eight(lc, hc, sd) * G).sum().backward()
    close_grad(lg.grad, lc.grad)
    close_grad(hg.grad, hc.grad)
    for i in range(4):
        close_grad(Ws[i].grad, sd[f"mapping_matrix.similarity1.conv{i}.weight"].grad)


def test_ecm_module_tuple(ecm, cmfsm_sd):
    mm = ecm.eight_related_context_mapping()
    mm.load_state_dict({k[len("mapping_matrix."):]: v for k, v in cmfsm_sd.items() if k.startswith("mapping_matrix.")})
    mm = mm.cuda()
    planes = mm(dev(seeded("g2.lr", 1, 32, 3, 4)), dev(seeded("g2.hr", 1, 32, 12, 16)), None, None)
    assert len(planes) == 9 and all(p.shape == (1, 1, 12, 16) for p in planes)
    close(torch.cat(planes, 1), load_golden("g2_ecm_weights")["w9"], 1e-4, 1e-6)
    with pytest.raises(ValueError):
        mm(dev(seeded("x", 1, 32, 4, 4)), dev(seeded("y", 1, 32, 12, 12)), None, None)      # scale 3


# ------------------------------------------------------------------ a1 + a5 fused: first conv on the concat volume
@pytest.mark.parametrize("B,h,w,D", [(1, 8, 16, 6), (2, 6, 36, 12), (1, 5, 9, 2), (1, 4, 12, 20), (1, 8, 64, 48), (1, 3, 5, 3)])
def test_costvol_conv3d_split(ecm, B, h, w, D):
    """conv3d(concat volume) computed as class-indexed 2-D convolutions (no 4-D volume) == the plain composition."""
    L, R = seeded("cvc.L", B, 32, h, w), seeded("cvc.R", B, 32, h, w)
    W = seeded("cvc.W", 32, 64, 3, 3, 3) * (2.0 / (27 * 64)) ** 0.5
    G = seeded("cvc.G", B, 32, D, h, w)
    Ls, Rs, Ws = (t.clone().requires_grad_() for t in (L, R, W))
    ref = F.conv3d(O.cost_volume(Ls, Rs, D), Ws, None, 1, 1)
    ref.backward(G)
    Lg, Rg, Wg = (dev(t).requires_grad_() for t in (L, R, W))
    y = ecm.ops.costvol_conv3d(Lg, Rg, Wg, D)
    y.backward(dev(G))
    close(y, ref, 1e-4, 2e-5)
    close(Lg.grad, Ls.grad, 1e-4, 2e-5)
    close(Rg.grad, Rs.grad, 1e-4, 2e-5)
    close(Wg.grad, Ws.grad, 1e-4, 1e-4 * float(Ws.grad.abs().max()))


# ------------------------------------------------------------------ GroupNorm
@pytest.mark.parametrize("B,C,dims,relu,skip", [(1, 32, (4, 6, 10), True, False), (2, 64, (3, 5, 7), True, True),
                                                (1, 32, (8, 8, 8), False, True), (2, 32, (2, 3, 5), False, False),
                                                (1, 64, (16, 24, 40), True, True),
                                                # several workgroups per channel in the register-resident cluster kernels
                                                (2, 32, (8, 48, 96), True, False), (1, 64, (8, 48, 96), True, True),
                                                (1, 128, (1, 96, 160), False, False), (2, 32, (12, 40, 100), False, True)])
def test_groupnorm(ecm, B, C, dims, relu, skip):
    x = seeded("gn.x", B, C, *dims) * 1.7 + 0.3
    gm, bt = 1 + 0.2 * seeded("gn.g", C), 0.2 * seeded("gn.b", C)
    sk = seeded("gn.s", B, C, *dims) if skip else None
    xs = [t.clone().requires_grad_() if t is not None else None for t in (x, gm, bt, sk)]
    xg = [dev(t).requires_grad_() if t is not None else None for t in (x, gm, bt, sk)]
    y = ecm.ops.group_norm_act(xg[0], xg[1], xg[2], xg[3], relu)
    ref = F.group_norm(xs[0], 32, xs[1], xs[2], 1e-5)
    if skip:
        ref = ref + xs[3]
    if relu:
        ref = F.relu(ref)
    close(y, ref, 1e-4, 1e-5)
    G = seeded("gn.G", B, C, *dims)
    y.backward(dev(G)); ref.backward(G)
    close(xg[0].grad, xs[0].grad, 1e-3, 1e-4)
    close(xg[1].grad, xs[1].grad, 1e-3, 1e-3)
    close(xg[2].grad, xs[2].grad, 1e-3, 1e-3)
    if skip:
        close(xg[3].grad, xs[3].grad, 1e-5, 1e-6)


# ------------------------------------------------------------------ conv / deconv
@pytest.mark.parametrize("B,Ci,Co,dims,stride", [
    (1, 32, 32, (4, 8, 32), 1), (1, 64, 32, (8, 8, 12), 1), (2, 32, 32, (5, 9, 37), 1), (1, 32, 1, (8, 8, 12), 1),
    (1, 32, 64, (8, 16, 24), 2), (1, 64, 64, (4, 8, 12), 1), (1, 64, 64, (8, 8, 12), 2), (1, 32, 64, (6, 10, 70), 2),
    (1, 32, 32, (12, 20, 60), 1), (2, 32, 1, (20, 14, 70), 1), (1, 16, 1, (5, 7, 33), 1), (1, 32, 1, (33, 6, 30), 1),
    (1, 8, 1, (2, 3, 4), 1)])
def test_conv3d_fwd(ecm, B, Ci, Co, dims, stride):
    x = seeded("cv3.x", B, Ci, *dims)
    w = seeded("cv3.w", Co, Ci, 3, 3, 3) * (2.0 / (27 * Ci)) ** 0.5
    y = ecm.ops.conv3d_k3(dev(x), dev(w), stride)
    close(y, F.conv3d(x, w, None, stride, 1), 1e-4, 1e-5)


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 32, 32, 20, 40), (1, 64, 64, 33, 50), (1, 32, 64, 16, 16), (2, 64, 32, 48, 70),
                                         (1, 32, 32, 5, 7)])
def test_conv2d_k3_wgrad(ecm, B, Ci, Co, H, W):
    """Encoder 3x3 / stride 1 Conv2d on the native 2-D family (forward, data and weight gradient) == autograd of F.conv2d."""
    x = seeded("cv2.x", B, Ci, H, W)
    w = seeded("cv2.w", Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5
    G = seeded("cv2.G", B, Co, H, W)
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    F.conv2d(xs, ws, None, 1, 1).backward(G)
    xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
    y = ecm.ops.conv2d_k3(xg, wg)
    y.backward(dev(G))
    close(y, F.conv2d(x, w, None, 1, 1), 1e-4, 1e-5)
    close(xg.grad, xs.grad, 1e-4, 1e-5)
    close(wg.grad, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))


@pytest.mark.parametrize("B,Ci,Co,dims", [(1, 64, 64, (2, 4, 6)), (1, 64, 32, (4, 8, 12)), (2, 64, 32, (3, 5, 35)),
                                          (1, 64, 64, (3, 9, 15))])
def test_deconv3d_fwd(ecm, B, Ci, Co, dims):
    x = seeded("dc3.x", B, Ci, *dims)
    w = seeded("dc3.w", Ci, Co, 3, 3, 3) * (2.0 / (27 * Ci)) ** 0.5
    y = ecm.ops.deconv3d_k3s2(dev(x), dev(w))
    close(y, F.conv_transpose3d(x, w, None, 2, 1, 1), 1e-4, 1e-5)


@pytest.mark.parametrize("B,Ci,Co,dims,stride", [
    (1, 32, 32, (4, 8, 32), 1), (2, 64, 32, (4, 6, 20), 1), (1, 32, 1, (8, 8, 12), 1), (1, 32, 64, (8, 16, 24), 2),
    (1, 64, 64, (4, 8, 12), 1), (2, 64, 64, (8, 8, 12), 2), (1, 32, 32, (5, 7, 19), 1),
    # the classifier's 32 -> 1 layer: several footprints, ragged edges, several disparity chunks
    (2, 32, 1, (20, 14, 70), 1), (1, 16, 1, (5, 7, 33), 1), (1, 32, 1, (33, 6, 30), 1)])
def test_conv3d_bwd(ecm, B, Ci, Co, dims, stride):
    x = seeded("cb3.x", B, Ci, *dims)
    w = seeded("cb3.w", Co, Ci, 3, 3, 3) * (2.0 / (27 * Ci)) ** 0.5
    xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
    y = ecm.ops.conv3d_k3(xg, wg, stride)
    xc, wc = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv3d(xc, wc, None, stride, 1)
    G = seeded("cb3.G", *ref.shape)
    y.backward(dev(G)); ref.backward(G)
    close(xg.grad, xc.grad, 1e-4, 1e-5)
    # a weight gradient is a sum over every voxel: the absolute tolerance scales with the magnitude of the sums
    close(wg.grad, wc.grad, 1e-4, 1e-5 * max(10.0, float(wc.grad.abs().max())))


@pytest.mark.parametrize("B,Ci,Co,dims", [(1, 64, 64, (2, 4, 6)), (2, 64, 32, (3, 5, 19))])
def test_deconv3d_bwd(ecm, B, Ci, Co, dims):
    x = seeded("db3.x", B, Ci, *dims)
    w = seeded("db3.w", Ci, Co, 3, 3, 3) * (2.0 / (27 * Ci)) ** 0.5
    xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
    y = ecm.ops.deconv3d_k3s2(xg, wg)
    xc, wc = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv_transpose3d(xc, wc, None, 2, 1, 1)
    G = seeded("db3.G", *ref.shape)
    y.backward(dev(G)); ref.backward(G)
    close(xg.grad, xc.grad, 1e-4, 1e-5)
    close(wg.grad, wc.grad, 1e-4, 1e-4)


# ------------------------------------------------------------------ modules and the whole path
def test_hourglass_golden(ecm, cmfsm_sd):
    g = load_golden("g6_hourglass")
    hg = ecm.hourglass(32)
    hg.load_state_dict({k[len("dres3."):]: v for k, v in cmfsm_sd.items() if k.startswith("dres3.")})
    hg = hg.cuda()
    x = dev(seeded("g6.x", 1, 32, 8, 8, 8))
    pre_in, post_in = dev(seeded("g6.pre", 1, 64, 4, 4, 4)), dev(seeded("g6.post", 1, 64, 4, 4, 4))
    with torch.no_grad():
        for tag, (pi, qi) in {"none": (None, None), "both": (pre_in, post_in)}.items():
            out, pre, post = hg(x, pi, qi)
            close(out, g[f"{tag}_out"], 1e-3, 1e-4); close(pre, g[f"{tag}_pre"], 1e-3, 1e-4)
            close(post, g[f"{tag}_post"], 1e-3, 1e-4)


def _load_hot(ecm, sd):
    model = ecm.get_model("cmfsm")
    model.load_state_dict(sd)
    return model.cuda()


def test_state_dict_contract(ecm, cmfsm_shapes):
    sd = ecm.get_model("cmfsm").state_dict()
    assert list(sd.keys()) == list(cmfsm_shapes.keys())
    assert all(list(sd[k].shape) == cmfsm_shapes[k] for k in sd)
    assert ecm.get_model("no_such_arch") is None


def test_hot_path_tiny_golden(ecm, cmfsm_sd):
    g = load_golden("g7a_hotpath")
    model = _load_hot(ecm, cmfsm_sd)
    lr_l, hr_l, lr_r = (dev(seeded(f"g7a.{n}", 1, 32, *s)) for n, s in
                        (("lr_l", (8, 12)), ("hr_l", (32, 48)), ("lr_r", (8, 12))))
    with torch.no_grad():
        preds = model.hot_path(lr_l, hr_l, lr_r)
    for i, p in enumerate(preds, 1):
        d = (p.cpu() - g[f"pred{i}"]).abs()
        assert d.max() <= 2e-2 and d.mean() <= 1e-3, (i, d.max(), d.mean())


def test_hot_path_tiny_grads_golden(ecm, cmfsm_sd):
    """Backward of the whole hot path against gradients produced by the reference's autograd (fixture g7a)."""
    g = load_golden("g7a_hotpath")
    model = _load_hot(ecm, cmfsm_sd)
    lr_l, hr_l, lr_r = (dev(seeded(f"g7a.{n}", 1, 32, *s)).requires_grad_() for n, s in
                        (("lr_l", (8, 12)), ("hr_l", (32, 48)), ("lr_r", (8, 12))))
    preds = model.hot_path(lr_l, hr_l, lr_r)
    loss = sum((p * dev(seeded(f"g7a.G{i}", 1, 1, 32, 48))).sum() for i, p in enumerate(preds, 1))
    loss.backward()
    close(lr_l.grad, g["g_lr_l"], 5e-3, 5e-4)
    close(hr_l.grad, g["g_hr_l"], 5e-3, 5e-4)
    close(lr_r.grad, g["g_lr_r"], 5e-3, 5e-4)
    for k, v in model.named_parameters():
        if k.startswith("feature_extraction"):
            continue
        kk = k.replace(".", "_")
        ref_norm = float(g["gn_" + kk])
        got_norm = float(v.grad.norm()) if v.grad is not None else 0.0
        assert abs(got_norm - ref_norm) <= 5e-3 * ref_norm + 1e-5, (k, got_norm, ref_norm)
        if "g_" + kk in g:
            close(v.grad, g["g_" + kk], 1e-2, 1e-3 * ref_norm / max(1.0, v.numel() ** 0.5) + 1e-5)


def test_full_model_train_step_golden(ecm, cmfsm_sd):
    """train.py:162-181 on the 256x512 fixture: loss and a few gradient norms vs the reference run."""
    g = load_golden("g8_full_cmfsm_256x512")
    model = _load_hot(ecm, cmfsm_sd).train()
    left, right = dev(seeded("g8.left", 1, 3, 256, 512)), dev(seeded("g8.right", 1, 3, 256, 512))
    gt = dev(torch.rand(1, 256, 512, generator=torch.Generator().manual_seed(8)) * 191.0)
    o = model(left, right)
    loss = O.train_loss(o, gt)
    loss.backward()
    close(loss, g["loss"], 1e-4, 1e-4)
    params = dict(model.named_parameters())
    for k in ("mapping_matrix.similarity1.conv0.weight", "dres0.0.0.weight", "dres2.conv6.0.weight", "classif3.2.weight",
              "feature_extraction.firstconv.0.0.weight", "feature_extraction.lastconv.2.weight", "dres4.conv5.1.bias"):
        ref = float(g["gn_" + k.replace(".", "_")])
        got = float(params[k].grad.norm())
        assert abs(got - ref) <= 2e-2 * ref + 1e-7, (k, got, ref)
        ref_t = g["g_" + k.replace(".", "_")]                    # full tensor, whole-tensor bound
        err = float((params[k].grad.cpu() - ref_t).abs().max())
        # a coarse whole-tensor bound against the reference's fp32 run (two fp32 evaluations of this network differ by up
        # to ~1.5 % of the largest element, see the fixture's own fp32-vs-fp64 distances); the sharp statement -- distance
        # from the fp64 truth bounded by the reference-fp32's own distance, for EVERY parameter -- is
        # tests/test_hip_fp64_yardstick.py
        assert err <= 3e-2 * float(ref_t.abs().max()) + 1e-9, (k, err, float(ref_t.abs().max()))


def test_hot_path_explicit_cost_volume_agrees(ecm, cmfsm_sd):
    """The reference's explicit op sequence (cost-volume kernel + 64->32 Conv3d) and the collapsed 2-D form of the same
    convolution give the same disparities."""
    models = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.models")
    model = _load_hot(ecm, cmfsm_sd)
    lr_l, hr_l, lr_r = (dev(seeded(f"g7x.{n}", 1, 32, *s)) for n, s in
                        (("lr_l", (8, 16)), ("hr_l", (32, 64)), ("lr_r", (8, 16))))
    with torch.no_grad():
        a = model.hot_path(lr_l, hr_l, lr_r)
        models.EXPLICIT_COST_VOLUME = True
        try:
            b = model.hot_path(lr_l, hr_l, lr_r)
        finally:
            models.EXPLICIT_COST_VOLUME = False
    for p, q in zip(a, b):
        close(p, q, 1e-4, 2e-3)


def test_hot_path_batch2_q1(ecm, cmfsm_sd):
    g = load_golden("g7q1_hotpath")
    model = _load_hot(ecm, cmfsm_sd)
    lr_l, hr_l, lr_r = (dev(seeded(f"g7q1.{n}", 2, 32, *s)) for n, s in
                        (("lr_l", (4, 8)), ("hr_l", (16, 32)), ("lr_r", (4, 8))))
    with torch.no_grad():
        preds = model.hot_path(lr_l, hr_l, lr_r)
    for i, p in enumerate(preds, 1):
        assert p.shape == (2, 1, 16, 32)
        d = (p.cpu() - g[f"pred{i}"]).abs()
        assert d.max() <= 2e-2 and d.mean() <= 1e-3, (i, d.max(), d.mean())


def test_full_model_golden(ecm, cmfsm_sd):
    g = load_golden("g8_full_cmfsm_256x512")
    model = _load_hot(ecm, cmfsm_sd)
    left, right = dev(seeded("g8.left", 1, 3, 256, 512)), dev(seeded("g8.right", 1, 3, 256, 512))
    with torch.no_grad():
        o = model(left, right)
    for i, name in enumerate(("o1", "o2", "o3")):
        assert o[i].shape == (1, 1, 256, 512)
        d = (o[i].cpu()[..., ::4, ::4] - g[name]).abs()
        # stated tolerance for the END-TO-END model (disparities span 0..191 px): max 0.02 px, mean 0.001 px
        assert d.max() <= 2e-2 and d.mean() <= 1e-3, (name, d.max(), d.mean())


@pytest.mark.parametrize("B,H,W", [(1, 16, 32), (2, 37, 53), (4, 256, 512)])
def test_stereo_loss3(ecm, B, H, W):
    """Fused loss + metrics kernel == the harness restatement (train.py:162,172-174; train_kitti.py:213-216)."""
    gt = (seeded("loss.gt", B, H, W).abs() * 120.0)                 # some pixels > 192, and
    gt[:, ::7, ::5] = 0.0                                            # some invalid (0) pixels
    ps = [gt + seeded(f"loss.p{i}", B, 1, H, W).squeeze(1) * s for i, s in ((1, 4.0), (2, 2.0), (3, 0.7))]
    cpu = [p.clone().unsqueeze(1).requires_grad_() for p in ps]
    ref = O.train_loss(cpu, gt)
    ref.backward()
    epe, err3 = O.kitti_metrics(cpu[2].detach(), gt)
    gpu = [dev(p).unsqueeze(1).requires_grad_() for p in ps]
    loss, met = ecm.ops.stereo_loss3(gpu, dev(gt))
    (loss * 1.5).backward()
    mask = (gt < 192) & (gt > 0)
    close(loss, ref.detach(), 1e-5, 1e-6)
    assert float(met[1]) == float(mask.sum())
    close(met[2], epe, 1e-5, 1e-6)
    close(met[3], err3, 1e-5, 1e-4)
    for g, c in zip(gpu, cpu):
        close(g.grad, 1.5 * c.grad, 1e-5, 1e-9)
    # empty mask: NaN, like the reference's mean over an empty selection
    l0, _ = ecm.ops.stereo_loss3([dev(p).unsqueeze(1) for p in ps], dev(torch.zeros_like(gt)))
    assert torch.isnan(l0)
    with pytest.raises(RuntimeError):
        ecm.ops.stereo_loss3([gpu[0], gpu[1], gpu[2][:, :, :-1]], dev(gt))


def _frames(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randint(0, 256, (B, H, W, 6), generator=g).float()          # uint8 images promoted to float32
    disp = torch.rand(B, H, W, 1, generator=g) * 300.0
    return torch.cat([rgb, disp], -1).contiguous()


def test_frame_prep_train_crop_bit_exact(ecm):
    """GPU frame preparation == Flying3d.__getitem__ (train split) for given crop origins, bit for bit."""
    fr = _frames(3, 300, 620, 5)
    y0, x0 = [0, 44, 17], [108, 0, 61]
    l, r, d, im = ecm.ops.frame_prep(dev(fr), y0, x0, 256, 512, want_image=True)
    for b in range(3):
        L, R, Dp, I = O.flying3d_sample(fr[b].numpy(), "train", (y0[b], x0[b]))
        assert torch.equal(l[b].cpu(), L) and torch.equal(r[b].cpu(), R)
        assert torch.equal(d[b].cpu(), Dp) and torch.equal(im[b].cpu(), I)


def test_frame_prep_large_batch(ecm):
    """More samples than one launch carries crop origins for (32): the launcher splits the batch."""
    fr = _frames(35, 260, 516, 9)
    y0 = [i % 5 for i in range(35)]
    x0 = [(3 * i) % 5 for i in range(35)]
    l, r, d = ecm.ops.frame_prep(dev(fr), y0, x0, 256, 512)
    for b in (0, 31, 32, 34):
        L, R, Dp, _ = O.flying3d_sample(fr[b].numpy(), "train", (y0[b], x0[b]))
        assert torch.equal(l[b].cpu(), L) and torch.equal(r[b].cpu(), R) and torch.equal(d[b].cpu(), Dp)


def test_frame_prep_eval_pad_bit_exact(ecm):
    """Eval split: rows [0,540) then the frame's last 36 rows again (576 rows), 960 columns."""
    fr = _frames(2, 540, 960, 6)
    l, r, d = ecm.ops.frame_prep(dev(fr), [0, 0], [0, 0], 576, 960, split=540, tail=36)
    for b in range(2):
        L, R, Dp, _ = O.flying3d_sample(fr[b].numpy(), "test")
        assert L.shape == (3, 576, 960)
        assert torch.equal(l[b].cpu(), L) and torch.equal(r[b].cpu(), R) and torch.equal(d[b].cpu(), Dp)
    with pytest.raises(RuntimeError):
        ecm.ops.frame_prep(dev(fr), [300, 0], [0, 0], 256, 512)             # window outside the frame
    with pytest.raises(RuntimeError):
        ecm.ops.frame_prep(dev(fr[..., :6].contiguous()), [0, 0], [0, 0], 256, 512)


def test_cpu_tensor_is_refused(ecm):
    with pytest.raises(RuntimeError):
        ecm.ops.cost_volume(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4), 2)


# ------------------------------------------------------------------ the other registered architectures (a4, a10, a11)
from test_oracle_golden import ARCHS, arch_inputs, arch_sd  # noqa: E402


@pytest.mark.parametrize("B,h,w,s", [(1, 4, 8, 8), (2, 3, 5, 4), (1, 2, 4, 16), (1, 5, 9, 8)])
def test_six_related_weights_vs_oracle(ecm, cmfsm_sd, B, h, w, s):
    lr, hr = seeded("sx.lr", B, 32, h, w), seeded("sx.hr", B, 32, h * s, w * s)
    lr_r, hr_r = seeded("sx.lr_r", B, 32, h, w), seeded("sx.hr_r", B, 32, h * s, w * s)
    Ws = [dev(t) for t in _mlp(cmfsm_sd)]
    m5 = ecm.ops.context_weights(dev(lr), dev(hr), *Ws, 1)
    mt3 = ecm.ops.context_weights(dev(lr_r), dev(hr_r), *Ws, 2)
    rm5, rmt3 = O.ecm_weights_six(lr, hr, lr_r, hr_r, cmfsm_sd)
    close(m5, rm5, 1e-4, 1e-5)
    close(mt3, rmt3, 1e-4, 1e-5)


@pytest.mark.parametrize("B,h,w,s", [(1, 4, 8, 8), (2, 3, 17, 4), (1, 2, 4, 16)])
def test_six_related_weights_bwd_vs_oracle(ecm, cmfsm_sd, B, h, w, s):
    lr, hr = seeded("sxb.lr", B, 32, h, w), seeded("sxb.hr", B, 32, h * s, w * s)
    lr_r, hr_r = seeded("sxb.lr_r", B, 32, h, w), seeded("sxb.hr_r", B, 32, h * s, w * s)
    G5, G3 = seeded("sxb.G5", B, 5, h * s, w * s), seeded("sxb.G3", B, 3, h * s, w * s)
    tg = [dev(t).requires_grad_() for t in (lr, hr, lr_r, hr_r)]
    Ws = [dev(t).requires_grad_() for t in _mlp(cmfsm_sd)]
    m5 = ecm.ops.context_weights(tg[0], tg[1], *Ws, 1)
    mt3 = ecm.ops.context_weights(tg[2], tg[3], *Ws, 2)
    ((m5 * dev(G5)).sum() + (mt3 * dev(G3)).sum()).backward()
    sd = {k: v.clone().requires_grad_() for k, v in cmfsm_sd.items() if k.startswith("mapping_matrix")}
    tc = [t.clone().requires_grad_() for t in (lr, hr, lr_r, hr_r)]
    rm5, rmt3 = O.ecm_weights_six(*tc, sd)
    ((rm5 * G5).sum() + (rmt3 * G3).sum()).backward()
    for a, b_ in zip(tg, tc):
        close(a.grad, b_.grad, 2e-3, 2e-5)
    for i in range(4):
        close(Ws[i].grad, sd[f"mapping_matrix.similarity1.conv{i}.weight"].grad, 2e-3, 2e-4)


@pytest.mark.parametrize("NH,B,Dl,h,w,s", [(1, 1, 12, 4, 4, 16), (3, 1, 12, 3, 5, 16), (1, 2, 24, 4, 8, 8), (2, 1, 48, 4, 8, 4)])
def test_volume_mapping_vs_oracle(ecm, NH, B, Dl, h, w, s):
    c = seeded("vm.c", NH, B, Dl, h, w, scale=1.5)
    m5 = seeded("vm.m5", B, 5, h * s, w * s, scale=0.5)
    mt3 = seeded("vm.mt3", B, 3, h * s, w * s, scale=0.5)
    tg = [dev(t).requires_grad_() for t in (c, m5, mt3)]
    out = ecm.ops.volume_mapping(*tg, s)
    tc = [t.clone().requires_grad_() for t in (c, m5, mt3)]
    ref = torch.stack([O.volume_mapping(tc[0][:k + 1].sum(0), tc[1], tc[2], s, Dl * s) for k in range(NH)], 0)
    close(out, ref, 1e-4, 2e-4)
    G = seeded("vm.G", *ref.shape)
    out.backward(dev(G)); ref.backward(G)
    for a, b_ in zip(tg, tc):
        close(a.grad, b_.grad, 2e-3, 2e-3 * float(b_.grad.abs().max()))


@pytest.mark.parametrize("NH,B,Dl,h,w,Do,H,W", [(1, 1, 48, 4, 8, 192, 16, 32), (3, 2, 12, 3, 5, 192, 48, 80),
                                               (2, 1, 24, 4, 6, 192, 30, 50)])
def test_trilinear_head_vs_oracle(ecm, NH, B, Dl, h, w, Do, H, W):
    c = seeded("tl.c", NH, B, Dl, h, w, scale=1.5)
    cg = dev(c).requires_grad_()
    out = ecm.ops.trilinear_softargmin(cg, Do, H, W)
    cc = c.clone().requires_grad_()
    ref = torch.stack([O.trilinear_head(cc[:k + 1].sum(0), Do, H, W) for k in range(NH)], 0)
    close(out, ref, 1e-4, 2e-4)
    G = seeded("tl.G", *ref.shape)
    out.backward(dev(G)); ref.backward(G)
    close(cg.grad, cc.grad, 2e-3, 2e-3 * float(cc.grad.abs().max()))


@pytest.mark.parametrize("arch", list(ARCHS))
def test_arch_hot_path_golden(ecm, arch):
    """Every registered architecture's post-encoder path on the HIP kernels vs the reference's own forward (fixture)."""
    g = load_golden(f"arch_{arch}")
    sd = arch_sd(arch)
    model = ecm.get_model(arch)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("feature_extraction") for k in missing)
    model = model.cuda()
    lr_l, hr_l, lr_r, hr_r = (dev(t) for t in arch_inputs(arch))
    for t_ in (lr_l, hr_l, lr_r, hr_r):
        t_.requires_grad_()
    preds = model.hot_path(lr_l, hr_l, lr_r, hr_r)
    for i, p in enumerate(preds, 1):
        assert p.shape == g[f"pred{i}"].shape, (arch, p.shape, g[f"pred{i}"].shape)
        d = (p.detach().cpu() - g[f"pred{i}"]).abs()
        assert d.max() <= 2e-2 and d.mean() <= 1e-3, (arch, i, d.max(), d.mean())
    # backward of the whole path vs the gradients the reference's autograd produced
    loss = sum((p * dev(seeded(f"{arch}.G{i}", *p.shape))).sum() for i, p in enumerate(preds))
    loss.backward()
    for nm, t_ in (("g_lr_l", lr_l), ("g_hr_l", hr_l), ("g_lr_r", lr_r), ("g_hr_r", hr_r)):
        if nm in g:
            ref = g[nm]
            tol = 2e-2 * float(ref.abs().max()) + 1e-6
            assert (t_.grad.cpu() - ref).abs().max() <= tol, (arch, nm, float((t_.grad.cpu() - ref).abs().max()), tol)
    checked_full = 0
    for k, v in model.named_parameters():
        if k.startswith("feature_extraction"):
            continue
        kk = k.replace(".", "_")
        ref_norm = float(g["gn_" + kk])
        got = float(v.grad.norm()) if v.grad is not None else 0.0
        assert abs(got - ref_norm) <= 2e-2 * ref_norm + 1e-5, (arch, k, got, ref_norm)
        if "g_" + kk in g:          # full tensors for a few parameters per architecture: a norm cannot see a permutation
            ref_t = g["g_" + kk]
            err = float((v.grad.cpu() - ref_t).abs().max())
            assert err <= 2e-2 * float(ref_t.abs().max()) + 1e-6, (arch, k, err, float(ref_t.abs().max()))
            checked_full += 1
    assert checked_full >= 4, (arch, checked_full)


def test_arch_state_dict_contracts(ecm):
    import json, os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "arch_state_shapes.json")) as f:
        shapes = json.load(f)
    for arch, sh in shapes.items():
        sd = ecm.get_model(arch).state_dict()
        assert list(sd.keys()) == list(sh.keys()), arch
        assert all(list(sd[k].shape) == sh[k] for k in sh), arch


# ------------------------------------------------------------------ general 2-D conv family (encoder, class convolutions)
# (B, Ci, Co, H, W, k, stride, dil): every conv layer shape of the registered encoders (models._ENCODERS: stem, stride-2,
# dilated 2 / 4, 64 / 128 / 320 / 384-channel stages, 1x1 projections incl. stride 2 and the tiny SPP maps) at small sizes
_ENC_LAYERS = [
    (2, 3, 32, 20, 40, 3, 1, 1), (1, 32, 32, 33, 70, 3, 1, 1), (2, 32, 32, 34, 66, 3, 2, 1), (1, 32, 32, 21, 37, 3, 2, 1),
    (1, 32, 64, 24, 40, 3, 2, 1), (1, 64, 64, 18, 35, 3, 1, 1), (1, 64, 128, 16, 40, 3, 1, 1), (1, 128, 128, 17, 33, 3, 1, 1),
    (1, 64, 128, 16, 36, 3, 1, 2), (1, 128, 128, 20, 36, 3, 1, 2), (1, 128, 128, 24, 40, 3, 1, 4), (1, 64, 128, 22, 34, 3, 2, 1),
    (1, 320, 128, 12, 36, 3, 1, 1), (1, 384, 128, 9, 33, 3, 1, 1),
    (2, 32, 64, 16, 34, 1, 2, 1), (1, 32, 32, 15, 33, 1, 2, 1), (1, 64, 128, 16, 34, 1, 1, 1), (1, 64, 128, 14, 30, 1, 2, 1),
    (2, 128, 32, 9, 15, 1, 1, 1), (8, 128, 32, 2, 3, 1, 1, 1),
]


@pytest.mark.parametrize("B,Ci,Co,H,W,k,stride,dil", _ENC_LAYERS)
def test_conv2d_family_vs_torch(ecm, B, Ci, Co, H, W, k, stride, dil):
    """nn.Conv2d(bias=False) of the encoder (cmfsm.py:36-46, 126-236) on the MFMA 2-D family: forward, data gradient and
    weight gradient vs CPU F.conv2d autograd."""
    assert ecm.ops.conv2d_supported(Ci, Co, k, k, stride, dil)
    x = seeded("c2.x", B, Ci, H, W)
    w = seeded("c2.w", Co, Ci, k, k) * (2.0 / (k * k * Ci)) ** 0.5
    pad = dil * (k - 1) // 2
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv2d(xs, ws, None, stride, pad, dil)
    G = seeded("c2.G", *ref.shape)
    ref.backward(G)
    xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
    y = ecm.ops.conv2d(xg, wg, stride, dil)
    y.backward(dev(G))
    close(y, ref, 1e-4, 1e-5)
    close(xg.grad, xs.grad, 1e-4, 2e-5)
    close(wg.grad, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))


@pytest.mark.parametrize("B,C,h,w", [(1, 32, 12, 40), (2, 32, 9, 33)])
def test_class_convolutions_vs_torch(ecm, B, C, h, w):
    """The two class-indexed convolutions of the collapsed cost volume (ops.costvol_conv3d; cmfsm.py:667-684): P = 3x3,
    32 -> 15*32, and Q = sheared 3x5, 32 -> 6*32 on the target features with two zero columns on the left (asymmetric
    padding, output width w+2) -- forward and both gradients vs CPU F.conv2d autograd."""
    L, R = seeded("cc.L", B, C, h, w), seeded("cc.R", B, C, h, w)
    wP, wQ = seeded("cc.wP", 15 * 32, C, 3, 3) * 0.1, seeded("cc.wQ", 6 * 32, C, 3, 5) * 0.1
    for (inp, wt, fn_ref, fn_hip) in (
            (L, wP, lambda a, b: F.conv2d(a, b, None, 1, 1), lambda a, b: ecm.ops.conv2d(a, b, 1, 1, 1, 1, h, w)),
            (R, wQ, lambda a, b: F.conv2d(F.pad(a, (2, 0)), b, None, 1, (1, 2)), lambda a, b: ecm.ops.conv2d(a, b, 1, 1, 1, 4, h, w + 2))):
        xs, ws = inp.clone().requires_grad_(), wt.clone().requires_grad_()
        ref = fn_ref(xs, ws)
        G = seeded("cc.G", *ref.shape)
        ref.backward(G)
        xg, wg = dev(inp).requires_grad_(), dev(wt).requires_grad_()
        y = fn_hip(xg, wg)
        assert y.shape == ref.shape
        y.backward(dev(G))
        close(y, ref, 1e-4, 1e-5)
        close(xg.grad, xs.grad, 1e-4, 2e-5)
        close(wg.grad, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))


def test_class_weight_kernels_vs_mask_einsum(ecm):
    """ops.ClassWeights (the class-indexed kernels of the collapsed first convolution, cmfsm.py:667-684) against the 0/1 tap
    masks written out in plain torch, forward and backward."""
    mP, mQ = torch.zeros(15, 3, 3), torch.zeros(6, 3, 3, 5)
    for e in range(3):
        for kd in range(3):
            if (e == 0 and kd == 0) or (e == 2 and kd == 2):
                continue                                     # depth padding at the first / last disparity plane
            for kw in range(3):
                for dc in range(5):
                    if kw - kd >= dc - 2:                    # the wedge `x >= d` at the tap's position, class dc = clamp(d-x,-2,2)+2
                        mP[dc * 3 + e, kd, kw] = 1.0
                mQ[e * 2 + 0, kd, kw, kw - kd + 2] = 1.0
                if kw != 2:
                    mQ[e * 2 + 1, kd, kw, kw - kd + 2] = 1.0   # right border column: the tap to the right is outside
    Co, C = 8, 12
    w = seeded("cw.w", Co, 2 * C, 3, 3, 3)
    ws = w.clone().requires_grad_()
    rP = torch.einsum("xdk,oidhk->xoihk", mP, ws[:, :C]).reshape(15 * Co, C, 3, 3)
    rQ = torch.einsum("xdkq,oidhk->xoihq", mQ, ws[:, C:]).reshape(6 * Co, C, 3, 5)
    GP, GQ = seeded("cw.GP", *rP.shape), seeded("cw.GQ", *rQ.shape)
    ((rP * GP).sum() + (rQ * GQ).sum()).backward()
    wg = dev(w).requires_grad_()
    wP, wQ = ecm.ops.ClassWeights.apply(wg)
    ((wP * dev(GP)).sum() + (wQ * dev(GQ)).sum()).backward()
    close(wP, rP.detach(), 1e-6, 1e-7)
    close(wQ, rQ.detach(), 1e-6, 1e-7)
    close(wg.grad, ws.grad, 1e-5, 1e-6)


def test_no_miopen_convolution_in_the_model(ecm):
    """SURVEY 8f n2 / VERDICT r1: no convolution of the registered architectures is left on PyTorch-ROCm (MIOpen)."""
    mdl = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.models")
    for arch in ("cmfsm", "cmfsm_sub_8", "cmfsm_sub_16", "cm_sub_4", "cm_sub_8", "cm_sub_16", "bilinear_cmf"):
        model = ecm.get_model(arch)
        enc = [m for m in model.feature_extraction.modules() if isinstance(m, torch.nn.Conv2d)]
        assert enc and all(isinstance(m, mdl.EncConv2d) and m._native() for m in enc), arch


# ------------------------------------------------------------------ Winograd F(2x2,3x3) kernels (csrc/conv_wino.hip)
@pytest.mark.parametrize("B,Ci,Co,dims", [(1, 32, 32, (4, 6, 64)), (2, 32, 32, (5, 7, 70)), (1, 64, 64, (3, 9, 33)), (1, 32, 64, (2, 4, 130)),
                                          (1, 8, 12, (3, 5, 9)), (1, 32, 32, (1, 1, 1)), (1, 64, 32, (6, 13, 65)),
                                          # row ends: the patch is read as column pairs, shifted at either border
                                          (1, 4, 4, (2, 3, 2)), (1, 8, 8, (2, 5, 3)), (1, 16, 16, (2, 3, 65)), (1, 32, 32, (2, 4, 66)),
                                          (1, 16, 16, (1, 3, 127)), (1, 16, 16, (2, 2, 128)), (1, 8, 8, (1, 4, 129))])
def test_winograd_conv3d_vs_torch_and_direct(ecm, B, Ci, Co, dims):
    """nn.Conv3d(k 3, stride 1, pad 1, bias=False) (convbn_3d, cmfsm.py:49-58) on the Winograd kernel: forward and data
    gradient vs CPU F.conv3d autograd, and vs the direct implicit-GEMM kernel (same fp32 class; rounding differs)."""
    x = seeded("wn.x", B, Ci, *dims)
    w = seeded("wn.w", Co, Ci, 3, 3, 3) * (2.0 / (27 * Ci)) ** 0.5
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv3d(xs, ws, None, 1, 1)
    G = seeded("wn.G", *ref.shape)
    ref.backward(G)
    res = {}
    prev = ecm.ops.WINOGRAD
    try:
        for flag in (True, False):
            ecm.ops.WINOGRAD = flag
            xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
            y = ecm.ops.conv3d_k3(xg, wg, 1)
            y.backward(dev(G))
            res[flag] = (y.detach().cpu(), xg.grad.cpu(), wg.grad.cpu())
    finally:
        ecm.ops.WINOGRAD = prev
    for flag in (True, False):
        close(res[flag][0], ref, 1e-4, 2e-5)
        close(res[flag][1], xs.grad, 1e-4, 2e-5)
        close(res[flag][2], ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))
    close(res[True][0], res[False][0], 1e-4, 1e-5)


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 32, 32, 20, 64), (1, 64, 64, 33, 50), (1, 3, 32, 17, 37), (1, 128, 128, 9, 70), (1, 320, 128, 6, 34),
                                         (2, 32, 480, 12, 40), (1, 8, 8, 5, 2), (1, 8, 8, 4, 3), (1, 64, 64, 7, 65), (1, 16, 16, 6, 128)])
def test_winograd_conv2d_vs_torch(ecm, B, Ci, Co, H, W):
    """The encoder's 3x3 / stride 1 / pad 1 Conv2d layers (cmfsm.py:36-46) and the P class convolution on the 2-D
    instantiation of the Winograd kernel: forward + data gradient vs CPU F.conv2d autograd."""
    assert ecm.ops.WINOGRAD
    prev_min, ecm.ops.WINO2D_MIN_CI = ecm.ops.WINO2D_MIN_CI, 1          # every shape through the Winograd kernel here
    try:
        _winograd_conv2d_case(ecm, B, Ci, Co, H, W)
    finally:
        ecm.ops.WINO2D_MIN_CI = prev_min


def _winograd_conv2d_case(ecm, B, Ci, Co, H, W):
    x = seeded("wn2.x", B, Ci, H, W)
    w = seeded("wn2.w", Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv2d(xs, ws, None, 1, 1)
    G = seeded("wn2.G", *ref.shape)
    ref.backward(G)
    xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
    y = ecm.ops.conv2d(xg, wg, 1, 1)
    y.backward(dev(G))
    close(y, ref, 1e-4, 2e-5)
    close(xg.grad, xs.grad, 1e-4, 2e-5)
    close(wg.grad, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))


@pytest.mark.parametrize("B,Ci,Co,dims", [(1, 32, 32, (4, 8, 16)), (2, 32, 32, (5, 7, 70)), (1, 64, 64, (3, 9, 33)), (1, 32, 64, (2, 4, 130)),
                                          (1, 8, 12, (3, 5, 9)), (1, 64, 32, (6, 13, 65))])
def test_winograd_wgrad3d_vs_torch(ecm, B, Ci, Co, dims):
    """Weight gradient of nn.Conv3d(k 3, stride 1, pad 1) in Winograd form vs CPU autograd and vs the direct kernel."""
    x, gy = seeded("ww.x", B, Ci, *dims), seeded("ww.g", B, Co, *dims)
    ws = (seeded("ww.w", Co, Ci, 3, 3, 3) * 0.1).requires_grad_()
    F.conv3d(x, ws, None, 1, 1).backward(gy)
    got = ecm.ops._wino_wgrad(dev(x), dev(gy), Co, Ci, 3)
    close(got, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 32, 32, 20, 64), (1, 64, 64, 33, 50), (1, 3, 32, 17, 37), (1, 128, 128, 9, 70), (2, 32, 480, 12, 40)])
def test_winograd_wgrad2d_vs_torch(ecm, B, Ci, Co, H, W):
    """Weight gradient of the encoder's 3x3 / stride 1 Conv2d in Winograd form vs CPU autograd."""
    x, gy = seeded("ww2.x", B, Ci, H, W), seeded("ww2.g", B, Co, H, W)
    ws = (seeded("ww2.w", Co, Ci, 3, 3) * 0.1).requires_grad_()
    F.conv2d(x, ws, None, 1, 1).backward(gy)
    got = ecm.ops._wino_wgrad(dev(x), dev(gy), Co, Ci, 1)
    close(got, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))


@pytest.mark.parametrize("n,numel", [(2, 1000), (3, 4099), (4, 2 * 32 * 6 * 9 * 17), (6, 515)])
def test_fork_sums_consumer_gradients_in_one_pass(ecm, n, numel):
    """ops.fork: a tensor with n consumers (cost0: first hourglass + three residual adds, cmfsm.py:686-693); the consumers'
    gradients are added by ecm_sum_n in the fixed order ((g0 + g1) + g2) + g3 -- bit-identical to that torch expression."""
    x = dev(seeded("fs.x", numel)).requires_grad_()
    gs = [dev(seeded(f"fs.g{k}", numel)) for k in range(n)]
    parts = ecm.ops.fork(x, n)
    assert all(p.data_ptr() == x.data_ptr() for p in parts)
    sum((p * g).sum() for p, g in zip(parts, gs)).backward()
    want = gs[0].clone()
    if n <= 4:
        for g in gs[1:]:
            want = want + g
        assert torch.equal(x.grad, want)
    else:
        close(x.grad, sum(g.double() for g in gs).float(), 1e-6, 1e-6)


@pytest.mark.parametrize("shape,Co", [((2, 32, 5, 7, 70), 32), ((1, 64, 3, 9, 33), 64), ((1, 32, 20, 64), 32), ((2, 64, 33, 50), 64),
                                      ((1, 128, 4, 12, 20), 128)])
def test_fork_folds_skip_gradient_into_data_gradient(ecm, shape, Co):
    """A tensor consumed by a convolution and by a skip connection (BasicBlock cmfsm.py:76-85, dres1 612-613, the hourglass
    outputs 686-695): with fork=True the convolution returns its input for the skip branch and its backward adds the skip
    gradient in the epilogue of the Winograd data-gradient kernel (ecm_conv_wino_fwd_add).  Against CPU autograd of the
    same graph, and against the unforked graph on the device (where autograd does the addition)."""
    three = len(shape) == 5 and shape[1] != 128
    planes = len(shape) == 5 and not three                   # [B,C,P,h,w]: phase planes of a dilated layer
    x = seeded("fk.x", *shape)
    Ci = shape[1]
    w = seeded("fk.w", Co, Ci, *((3, 3, 3) if three else (3, 3))) * (2.0 / ((27 if three else 9) * Ci)) ** 0.5
    G1 = seeded("fk.g1", shape[0], Co, *shape[2:])
    G2 = seeded("fk.g2", *shape)
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    if three:
        ref = F.conv3d(xs, ws, None, 1, 1)
    elif planes:
        B, C, P, h, wd = shape
        ref = F.conv2d(xs.permute(0, 2, 1, 3, 4).reshape(B * P, C, h, wd), ws, None, 1, 1).view(B, P, Co, h, wd).permute(0, 2, 1, 3, 4)
    else:
        ref = F.conv2d(xs, ws, None, 1, 1)
    ((ref * G1).sum() + (xs * xs * G2).sum()).backward()
    res = []
    for fork in (True, False):
        xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
        op = ecm.ops.conv3d_k3 if three else ecm.ops.conv2d_planes if planes else ecm.ops.conv2d
        args = (xg, wg, 1) if three else (xg, wg) if planes else (xg, wg, 1, 1)
        if fork:
            y, xa = op(*args, fork=True)
            assert xa.data_ptr() == xg.data_ptr()
        else:
            y, xa = op(*args), xg
        ((y * dev(G1)).sum() + (xa * xa * dev(G2)).sum()).backward()
        close(y, ref, 1e-4, 2e-5)
        close(xg.grad, xs.grad, 1e-4, 5e-5)
        close(wg.grad, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))
        res.append(xg.grad.cpu())
    close(res[0], res[1], 1e-5, 1e-5)


@pytest.mark.parametrize("B,C,H,W,d", [(2, 128, 24, 40, 2), (1, 64, 16, 36, 2), (1, 128, 16, 32, 4)])
def test_dilated_layer_as_phase_planes(ecm, B, C, H, W, d):
    """A dilation-d 3x3 convolution (feature_extraction layer4, cmfsm.py:150) run as d*d phase planes on the Winograd
    kernels (ops.phase_split / conv2d_planes / phase_merge): forward and both gradients vs CPU F.conv2d with dilation."""
    x = seeded("ph.x", B, C, H, W)
    w = seeded("ph.w", C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv2d(xs, ws, None, 1, d, d)
    G = seeded("ph.G", *ref.shape)
    ref.backward(G)
    xg, wg = dev(x).requires_grad_(), dev(w).requires_grad_()
    y = ecm.ops.phase_merge(ecm.ops.conv2d_planes(ecm.ops.phase_split(xg, d), wg), d)
    y.backward(dev(G))
    close(y, ref, 1e-4, 2e-5)
    close(xg.grad, xs.grad, 1e-4, 2e-5)
    close(wg.grad, ws.grad, 1e-4, 1e-4 * float(ws.grad.abs().max()))


# ------------------------------------------------------------------ fused classifier tail (round 4): GN + ReLU + Conv3d(32 -> 1)
@pytest.mark.parametrize("B,dims", [(1, (8, 8, 12)), (2, (5, 7, 33)), (1, (3, 17, 40)), (2, (9, 20, 65))])
def test_classifier_tail_vs_torch_and_unfused(ecm, B, dims):
    """ops.classifier_tail (GroupNorm(32) + ReLU applied while the 32 -> 1 kernels stage x; cmfsm.py:621-634) against the
    CPU composition F.conv3d(F.relu(F.group_norm(x))) with autograd, and against the three separate HIP stages: forward and
    the gradients w.r.t. x, gamma, beta and the 32 -> 1 weight."""
    ops = ecm.ops
    x = seeded("ct.x", B, 32, *dims) * 1.5 + 0.3
    gm, bt = 1.0 + 0.2 * seeded("ct.g", 32), 0.2 * seeded("ct.b", 32)
    w = seeded("ct.w", 1, 32, 3, 3, 3) * (2.0 / (27 * 32)) ** 0.5
    G = seeded("ct.G", B, 1, *dims)
    ref_in = [t.clone().requires_grad_() for t in (x, gm, bt, w)]
    ref = F.conv3d(F.relu(F.group_norm(ref_in[0], 32, ref_in[1], ref_in[2], 1e-5)), ref_in[3], None, 1, 1)
    ref.backward(G)
    a = [dev(t).requires_grad_() for t in (x, gm, bt, w)]
    y = ops.classifier_tail(*a)
    y.backward(dev(G))
    b = [dev(t).requires_grad_() for t in (x, gm, bt, w)]
    y2 = ops.conv3d_k3(ops.group_norm_act(b[0], b[1], b[2], None, True), b[3], 1)
    y2.backward(dev(G))
    close(y, ref, 1e-4, 1e-5)
    close(y, y2, 1e-5, 1e-6)                                   # same arithmetic; only the statistics' summation order differs
    for t, u, r, name in zip(a, b, ref_in, ("x", "gamma", "beta", "w")):
        close_grad(t.grad, r.grad, 2e-4)
        close_grad(t.grad, u.grad, 2e-5)
    ecm.ops.check_async_errors()


def test_classifier_tail_in_the_model_equals_the_separate_stages(ecm, cmfsm_sd):
    """models.C1_GN_FUSE on / off: same predictions and parameter gradients of the classifiers on a small hot path."""
    mdl = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.models")
    model = ecm.get_model("cmfsm")
    model.load_state_dict(cmfsm_sd)
    model = model.cuda().train()
    lr_l, hr_l, lr_r = (dev(seeded(n, 1, 32, *s)) for n, s in (("ctm.lr_l", (8, 16)), ("ctm.hr_l", (32, 64)), ("ctm.lr_r", (8, 16))))
    res = {}
    prev = mdl.C1_GN_FUSE
    try:
        for flag in (True, False):
            mdl.C1_GN_FUSE = flag
            model.zero_grad(set_to_none=True)
            preds = model.hot_path(lr_l, hr_l, lr_r)
            sum((p * p).mean() for p in preds).backward()
            res[flag] = ([p.detach().clone() for p in preds],
                         {k: p.grad.clone() for k, p in model.named_parameters() if k.startswith("classif") and p.grad is not None})
    finally:
        mdl.C1_GN_FUSE = prev
    for pa, pb in zip(res[True][0], res[False][0]):
        close(pa, pb, 0, 2e-3)                                  # px; the heads' statistics differ in summation order only
    assert res[True][1].keys() == res[False][1].keys() and len(res[True][1]) == 12
    for k in res[True][1]:
        close_grad(res[True][1][k], res[False][1][k], 2e-3)


@pytest.mark.parametrize("B,Co,D,h,w", [(1, 32, 6, 5, 16), (2, 8, 12, 3, 36), (1, 32, 48, 4, 240), (1, 4, 2, 3, 8), (1, 4, 20, 2, 12)])
def test_costvol_assembly_row_staged_equals_elementwise(ecm, B, Co, D, h, w):
    """ecm_costvol_conv_assemble_{fwd,bwd}: the row-staged kernels of round 4 (one workgroup per (b, co, y), operands in LDS) and
    the element-wise ones (taken for unaligned / non-multiple-of-4 / oversized rows) must agree BIT FOR BIT -- same per-element
    arithmetic, same order of additions.  The element-wise path is forced by handing the library a 4-byte-offset copy."""
    import ctypes as C
    lib = ecm._lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    P = dev(seeded("asm.P", B, 15 * Co, h, w))
    Q = dev(seeded("asm.Q", B, 6 * Co, h, w + 2))
    G = dev(seeded("asm.G", B, Co, D, h, w))

    def off(t):                                             # same values, base pointer 4 bytes past a 16-byte boundary
        buf = torch.empty(t.numel() + 4, device="cuda")
        v = buf[1:1 + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v
    y_rows = torch.full((B, Co, D, h, w), float("nan"), device="cuda")
    y_elem = torch.full((B, Co, D, h, w), float("nan"), device="cuda")
    lib.call("ecm_costvol_conv_assemble_fwd", p(P), p(Q), p(y_rows), B, Co, D, h, w, st)
    Po = off(P)
    lib.call("ecm_costvol_conv_assemble_fwd", p(Po), p(Q), p(y_elem), B, Co, D, h, w, st)
    assert torch.isfinite(y_rows).all() and torch.equal(y_rows, y_elem)
    outs = []
    for g in (G, off(G)):
        gP = torch.full((B, 15 * Co, h, w), float("nan"), device="cuda")
        gQ = torch.full((B, 6 * Co, h, w + 2), float("nan"), device="cuda")
        lib.call("ecm_costvol_conv_assemble_bwd", p(g), p(gP), p(gQ), B, Co, D, h, w, st)
        outs.append((gP, gQ))
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
