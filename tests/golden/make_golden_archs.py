#!/usr/bin/env python3
"""Golden fixtures for the OTHER registered architectures (cmfsm_sub_8/16, cm_sub_4/8/16, bilinear_cmf[_sub_8/16]):
parameter contracts (state_dict key -> shape) and the whole post-encoder path on tiny feature maps, produced by the
REFERENCE's own forward() behind a stub encoder.  Build container only.  Usage: python -B tests/golden/make_golden_archs.py"""
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

for name in ("torchvision", "torchvision.models", "cmf.caffe_pb2"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, "/root/reference")
from cmf.models import get_model  # noqa: E402
from oracle.weights import seeded, tensor_for  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)

ARCHS = {  # name: (scale, h, w)
    "cmfsm_sub_8": (8, 4, 8), "cmfsm_sub_16": (16, 4, 4), "cm_sub_4": (4, 4, 8), "cm_sub_8": (8, 4, 8),
    "cm_sub_16": (16, 4, 4), "bilinear_cmf": (4, 4, 8), "bilinear_cmf_sub_8": (8, 4, 8), "bilinear_cmf_sub_16": (16, 4, 4),
}


FULL_GRADS = ("dres0.0.0.weight", "dres0.2.1.weight", "dres2.conv6.0.weight", "dres2.conv5.1.bias", "classif1.2.weight",
              "mapping_matrix.similarity1.conv0.weight", "mapping_matrix.similarity1.conv1.weight",
              "mapping_matrix.similarity1.conv3.weight")


class _Stub(torch.nn.Module):
    def __init__(self, feats):
        super().__init__()
        self.feats, self.i = feats, 0

    def forward(self, x):
        lr_, hr_ = self.feats[self.i % 2]
        self.i += 1
        return lr_, None, hr_


shapes_all = {}
for arch, (s, h, w) in ARCHS.items():
    model = get_model(arch)
    sd = model.state_dict()
    shapes_all[arch] = {k: list(v.shape) for k, v in sd.items()}
    model.load_state_dict({k: tensor_for(k, v.shape) for k, v in sd.items()})
    lr_l = seeded(f"{arch}.lr_l", 1, 32, h, w).requires_grad_()
    hr_l = seeded(f"{arch}.hr_l", 1, 32, s * h, s * w).requires_grad_()
    lr_r = seeded(f"{arch}.lr_r", 1, 32, h, w).requires_grad_()
    hr_r = seeded(f"{arch}.hr_r", 1, 32, s * h, s * w).requires_grad_()
    model.feature_extraction = _Stub([(lr_l, hr_l), (lr_r, hr_r)])
    cap = {}
    hooks = []
    if hasattr(model, "mapping_matrix"):
        hooks.append(model.mapping_matrix.register_forward_hook(
            lambda m, i, o: cap.__setitem__("maps", [torch.cat([t.detach() for t in grp], 1) for grp in o])))
    hooks.append(model.classif1.register_forward_hook(lambda m, i, o: cap.__setitem__("classif1", o.detach().clone())))
    dummy = torch.zeros(1, 3, s * h, s * w)
    preds = model(dummy, dummy)
    for hk in hooks:
        hk.remove()
    out = {f"pred{i + 1}": p.detach() for i, p in enumerate(preds)}
    out["classif1"] = cap["classif1"]
    if "maps" in cap:
        out["m5"], out["mt3"] = cap["maps"]
    Gs = [seeded(f"{arch}.G{i}", *p.shape) for i, p in enumerate(preds)]
    loss = sum((p * g).sum() for p, g in zip(preds, Gs))
    loss.backward()
    for nm, t in (("g_lr_l", lr_l), ("g_hr_l", hr_l), ("g_lr_r", lr_r), ("g_hr_r", hr_r)):
        if t.grad is not None:
            out[nm] = t.grad
    for k, p in model.named_parameters():
        out["gn_" + k.replace(".", "_")] = p.grad.norm() if p.grad is not None else torch.zeros(())
        # full-tensor gradients of a few parameters per architecture (a norm cannot see a permuted gradient)
        if k in FULL_GRADS and p.grad is not None:
            out["g_" + k.replace(".", "_")] = p.grad.clone()
    np.savez_compressed(os.path.join(OUT, f"arch_{arch}.npz"), **{k: v.detach().numpy() for k, v in out.items()})
    print(arch, [tuple(p.shape) for p in preds], "params", sum(v.numel() for v in sd.values()),
          f"{os.path.getsize(os.path.join(OUT, f'arch_{arch}.npz')) / 1024:.0f} KiB", flush=True)

with open(os.path.join(OUT, "arch_state_shapes.json"), "w") as f:
    json.dump(shapes_all, f)
