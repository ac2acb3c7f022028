#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code on CPU.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (inputs are regenerated from seeds, outputs are stored) and are
what travels to the GPU box.  Usage:  python -B tests/golden/make_golden.py

Import recipe: the reference imports torchvision and a
protobuf-3-era caffe_pb2 at import time and calls .cuda() unconditionally;
stub the two dead imports and make .cuda() a no-op.
"""
from __future__ import annotations

import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F

for name in ("torchvision", "torchvision.models", "cmf.caffe_pb2"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, "/root/reference")

from cmf.models import get_model  # noqa: E402  (the reference)

ref = sys.modules["cmf.models.cmfsm"]
from oracle.weights import make_state_dict, seeded, tensor_for  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def load_seeded(module, prefix=""):
    """Load per-key seeded weights; keys are named as in the full cmfsm model."""
    sd = module.state_dict()
    new = {k: tensor_for(prefix + k, v.shape) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def sub(t, step):
    """Strided subsample of the two last dims."""
    return t[..., ::step, ::step].contiguous()


# ---------------------------------------------------------------- G0: parameter contract
model = get_model("cmfsm")
shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
with open(os.path.join(OUT, "cmfsm_state_shapes.json"), "w") as f:
    json.dump(shapes, f, indent=0)
print("cmfsm params:", sum(int(np.prod(s)) for s in shapes.values()), "tensors:", len(shapes))
load_seeded(model)

# ---------------------------------------------------------------- G1: cost volume via matchshifted
for tag, (B, C, h, w, D) in {"a": (1, 4, 5, 12, 6), "b": (2, 8, 4, 24, 20)}.items():
    L = seeded(f"g1{tag}.L", B, C, h, w)
    R = seeded(f"g1{tag}.R", B, C, h, w)
    ms = ref.matchshifted()
    cost = torch.cat([ms(L, R, d) for d in range(D)], 2)
    save(f"g1{tag}_costvol", cost=cost, shape=np.array([B, C, h, w, D]))

# ---------------------------------------------------------------- G2: eight-related ECM weights (+grads)
mm = load_seeded(ref.eight_related_context_mapping(), "mapping_matrix.")
lr = seeded("g2.lr", 1, 32, 3, 4).requires_grad_()
hr = seeded("g2.hr", 1, 32, 12, 16).requires_grad_()
planes = mm(lr, hr, None, None)
w9 = torch.cat(planes, 1)
G = seeded("g2.G", 1, 9, 12, 16)
(w9 * G).sum().backward()
save("g2_ecm_weights", w9=w9, g_lr=lr.grad, g_hr=hr.grad,
     **{"g_" + k.replace(".", "_"): p.grad for k, p in mm.named_parameters()})
tabs = ref.matrix_generation()
save("g2_tables", **{f"t{i}": t for i, t in enumerate(tabs)})

# ---------------------------------------------------------------- G4: soft-argmin (+grad)
cost = seeded("g4.cost", 2, 48, 6, 10, scale=2.0).requires_grad_()
disp = ref.disparityregression(48)(F.softmax(cost, dim=1))
Gd = seeded("g4.G", 2, 6, 10)
(disp * Gd).sum().backward()
save("g4_softargmin", disp=disp, g_cost=cost.grad)

# ---------------------------------------------------------------- G6: hourglass / dres / classif modules (+grads)
hg = load_seeded(ref.hourglass(32), "dres3.")
x = seeded("g6.x", 1, 32, 8, 8, 8).requires_grad_()
pre_in = seeded("g6.pre", 1, 64, 4, 4, 4).requires_grad_()
post_in = seeded("g6.post", 1, 64, 4, 4, 4).requires_grad_()
Go, Gp, Gq = seeded("g6.Go", 1, 32, 8, 8, 8), seeded("g6.Gp", 1, 64, 4, 4, 4), seeded("g6.Gq", 1, 64, 4, 4, 4)
res = {}
for tag, (pi, qi) in {"none": (None, None), "both": (pre_in, post_in)}.items():
    for p in hg.parameters():
        p.grad = None
    for t in (x, pre_in, post_in):
        t.grad = None
    out, pre, post = hg(x, pi, qi)
    ((out * Go).sum() + (pre * Gp).sum() + (post * Gq).sum()).backward()
    res.update({f"{tag}_out": out, f"{tag}_pre": pre, f"{tag}_post": post, f"{tag}_gx": x.grad.clone()})
    if pi is not None:
        res.update({f"{tag}_gpre": pre_in.grad.clone(), f"{tag}_gpost": post_in.grad.clone()})
    for k, p in hg.named_parameters():
        if k in ("conv1.0.0.weight", "conv1.0.1.weight", "conv1.0.1.bias", "conv5.0.weight",
                 "conv6.0.weight", "conv6.1.bias", "conv4.0.0.weight"):
            res[f"{tag}_g_{k.replace('.', '_')}"] = p.grad.clone()
save("g6_hourglass", **res)

# dres0 / dres1 / classif on the real cmfsm instance (weights already seeded)
x64 = seeded("g6.x64", 1, 64, 8, 8, 12).requires_grad_()
y0 = model.dres0(x64)
y1 = model.dres1(y0) + y0
yc = model.classif2(y1)
Gc = seeded("g6.Gc", 1, 1, 8, 8, 12)
(yc * Gc).sum().backward()
save("g6_dres_classif", dres0=y0, dres1=y1, classif2=yc, g_x64=x64.grad,
     g_dres0_0_0_weight=model.dres0[0][0].weight.grad, g_dres1_2_1_bias=model.dres1[2][1].bias.grad,
     g_classif2_2_weight=model.classif2[2].weight.grad, g_dres0_2_1_weight=model.dres0[2][1].weight.grad)
model.zero_grad()

# ---------------------------------------------------------------- G7: whole hot path on tiny feature maps
# The reference's forward() is run unmodified; only its 2-D encoder is replaced by a stub that
# returns chosen feature maps (the real encoder needs H,W >= 256, quirk Q4).


class _StubEncoder(torch.nn.Module):
    def __init__(self, feats):
        super().__init__()
        self.feats, self.i = feats, 0

    def forward(self, x):
        lr_, hr_ = self.feats[self.i % 2]
        self.i += 1
        return lr_, None, hr_


def run_tiny(B, h, w, tag, grads=True):
    lr_l = seeded(f"g7{tag}.lr_l", B, 32, h, w).requires_grad_()
    hr_l = seeded(f"g7{tag}.hr_l", B, 32, 4 * h, 4 * w).requires_grad_()
    lr_r = seeded(f"g7{tag}.lr_r", B, 32, h, w).requires_grad_()
    hr_r = seeded(f"g7{tag}.hr_r", B, 32, 4 * h, 4 * w)
    real = model.feature_extraction
    model.feature_extraction = _StubEncoder([(lr_l, hr_l), (lr_r, hr_r)])
    cap = {}
    hooks = [
        model.dres0.register_forward_pre_hook(lambda m, i: cap.__setitem__("cost", i[0].detach().clone())),
        model.dres0.register_forward_hook(lambda m, i, o: cap.__setitem__("dres0", o.detach().clone())),
        model.dres2.register_forward_hook(lambda m, i, o: cap.__setitem__("hg1", [t.detach().clone() for t in o])),
        model.dres4.register_forward_hook(lambda m, i, o: cap.__setitem__("hg3", [t.detach().clone() for t in o])),
        model.classif1.register_forward_hook(lambda m, i, o: cap.__setitem__("classif1", o.detach().clone())),
        model.classif3.register_forward_hook(lambda m, i, o: cap.__setitem__("classif3", o.detach().clone())),
        model.mapping_matrix.register_forward_hook(
            lambda m, i, o: cap.__setitem__("w9", torch.cat([t.detach() for t in o], 1))),
    ]
    model.zero_grad()
    dummy = torch.zeros(B, 3, 4 * h, 4 * w)
    p1, p2, p3 = model(dummy, dummy)
    for hk in hooks:
        hk.remove()
    model.feature_extraction = real
    out = dict(cap_w9=cap["w9"], cap_dres0=cap["dres0"], cap_classif1=cap["classif1"], cap_classif3=cap["classif3"],
               cap_hg1_out=cap["hg1"][0], cap_hg3_out=cap["hg3"][0], cap_hg3_post=cap["hg3"][2])
    # the loop-built volume equals cat_d(matchshifted) -- asserted on the reference's own code
    ms = ref.matchshifted()
    alt = torch.cat([ms(lr_l.detach(), lr_r.detach(), d) if d < w else torch.zeros(B, 64, 1, h, w)
                     for d in range(48)], 2)
    assert torch.equal(alt, cap["cost"]), "loop volume != matchshifted volume"
    if B == 1:
        out.update(pred1=p1, pred2=p2, pred3=p3)
    else:  # quirk Q1: [B,B,H,W]; keep the meaningful diagonal and the raw shape
        idx = torch.arange(B)
        out.update(pred1=p1[idx, idx].unsqueeze(1), pred2=p2[idx, idx].unsqueeze(1),
                   pred3=p3[idx, idx].unsqueeze(1), raw_shape=np.array(p1.shape))
    if grads:
        G1, G2, G3 = (seeded(f"g7{tag}.G{i}", B, 1, 4 * h, 4 * w) for i in (1, 2, 3))
        if B > 1:
            raise NotImplementedError
        ((p1 * G1).sum() + (p2 * G2).sum() + (p3 * G3).sum()).backward()
        out.update(g_lr_l=lr_l.grad, g_hr_l=hr_l.grad, g_lr_r=lr_r.grad)
        for k, p in model.named_parameters():
            if k.startswith("feature_extraction"):
                continue
            g = p.grad
            out["gn_" + k.replace(".", "_")] = g.norm() if g is not None else torch.zeros(())
            if g is not None and g.numel() <= 4096:
                out["g_" + k.replace(".", "_")] = g.clone()
        for k in ("dres0.0.0.weight", "dres2.conv1.0.0.weight", "dres3.conv5.0.weight",
                  "dres4.conv6.0.weight", "classif2.0.0.weight"):
            out["g_" + k.replace(".", "_")] = dict(model.named_parameters())[k].grad.clone()
    save(f"g7{tag}_hotpath", **out)
    save(f"g7{tag}_cost", cost=cap["cost"])


run_tiny(1, 8, 12, "a")
run_tiny(2, 4, 8, "q1", grads=False)

# ---------------------------------------------------------------- G8: full cmfsm at the minimum legal size (256x512)
left = seeded("g8.left", 1, 3, 256, 512)
right = seeded("g8.right", 1, 3, 256, 512)
gt = torch.rand(1, 256, 512, generator=torch.Generator().manual_seed(8)) * 191.0
cap = {}
hk = model.feature_extraction.register_forward_hook(
    lambda m, i, o: cap.__setitem__("fe", cap.get("fe") or [t.detach().clone() for t in o]))   # must return None
model.zero_grad()
model.train()   # 256x512: F.group_norm refuses branch1's [1,32,1,1] map at 256x256 (torch >= 1.x check)
o1, o2, o3 = model(left, right)
hk.remove()
mask = (gt < 192) & (gt > 0)
s1, s2, s3 = o1.squeeze(1), o2.squeeze(1), o3.squeeze(1)
loss = (0.5 * F.smooth_l1_loss(s1[mask], gt[mask], reduction="mean")
        + 0.7 * F.smooth_l1_loss(s2[mask], gt[mask], reduction="mean")
        + F.smooth_l1_loss(s3[mask], gt[mask], reduction="mean"))
loss.backward()
g8 = dict(o1=sub(o1, 4), o2=sub(o2, 4), o3=sub(o3, 4), loss=loss.detach(),
          o_mean=torch.stack([o1.mean(), o2.mean(), o3.mean()]).detach(),
          o_absmax=torch.stack([o1.abs().max(), o2.abs().max(), o3.abs().max()]).detach(),
          fe_lr=sub(cap["fe"][0], 4), fe_hr=sub(cap["fe"][2], 8), gt=sub(gt, 4))
for k, p in model.named_parameters():
    if k in ("mapping_matrix.similarity1.conv0.weight", "dres0.0.0.weight", "dres2.conv6.0.weight",
             "classif3.2.weight", "feature_extraction.firstconv.0.0.weight",
             "feature_extraction.lastconv.2.weight", "dres4.conv5.1.bias"):
        if p.grad is None:
            print("no grad for", k)
            continue
        g8["gn_" + k.replace(".", "_")] = p.grad.norm()
        g8["g_" + k.replace(".", "_")] = p.grad.clone()          # the full tensor: a norm cannot see a permuted gradient
save("g8_full_cmfsm_256x512", **g8)

print("done")
