#!/usr/bin/env python3
"""Full-frame fixtures at the BASELINE.json sizes: the REFERENCE's own cmfsm (cmfsm.py:655-774) run in
the build container on ONE SceneFlow-shaped frame padded as cmf/loader/Flying3d.py:66-72 does (540x960 -> 576x960) and ONE
KITTI-shaped frame padded as cmf/loader/KITTI.py:98-108 does (375x1242 -> 384x1248), eval forward under no_grad, once as the
reference is (fp32) and once in double precision (`model.double()` + `torch.FloatTensor = torch.DoubleTensor`).

  g11_fullframe_sceneflow_576x960.npz     g11_fullframe_kitti_384x1248.npz

Stored (fp32 run: suffix _32, fp64 run: _64): every 4th pixel of the three outputs; the low-resolution features of both
images (every 4th position), the nine ECM weight planes (every 8th pixel), the raw outputs of classif1..3 (every 2nd
disparity, every 4th position) and the distance of the reference-fp32 tensors from the fp64 ones (max / mean) -- the
yardstick the -m gpu tests bound the HIP path with.  Inputs are NOT stored: oracle/weights.py:fullframe_frame(kind) rebuilds
the raw frame from a seed and the loader restatement (oracle: flying3d_sample / kitti_eval_sample) pads and normalises it;
weights are the per-key seeded ones (oracle/weights.py:tensor_for).

Build container only (needs /root/reference).  Usage: python -B tests/golden/make_golden_fullframe.py [sceneflow] [kitti]"""
from __future__ import annotations

import os
import sys
import time
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
from cmf.models import get_model  # noqa: E402  (the reference)
from oracle import ecm_oracle as O  # noqa: E402  (loader restatements: padding + normalisation of the raw frame)
from oracle.weights import fullframe_frame, tensor_for  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)
_F32 = torch.FloatTensor


def inputs(kind):
    frame = fullframe_frame(kind)
    if kind == "sceneflow":
        left, right, disp, _ = O.flying3d_sample(frame, "test")
    else:
        left, right, disp = O.kitti_eval_sample(frame)
    return left.unsqueeze(0), right.unsqueeze(0), disp.unsqueeze(0)


def run(kind, double):
    torch.FloatTensor = torch.DoubleTensor if double else _F32       # the reference allocates the cost volume with it
    model = get_model("cmfsm")
    model.load_state_dict({k: tensor_for(k, v.shape) for k, v in model.state_dict().items()})
    left, right, _ = inputs(kind)
    if double:
        model, left, right = model.double(), left.double(), right.double()
    model.eval()
    cap = {"fe": [], "w9": None, "c": {}}
    hooks = [model.feature_extraction.register_forward_hook(lambda m, i, o: cap["fe"].append(o[0].detach())),
             model.mapping_matrix.register_forward_hook(lambda m, i, o: cap.__setitem__("w9", torch.cat([t.detach() for t in o], 1)))]
    for k in (1, 2, 3):
        hooks.append(getattr(model, f"classif{k}").register_forward_hook(
            lambda m, i, o, k=k: cap["c"].__setitem__(k, o.detach())))
    t0 = time.time()
    with torch.no_grad():
        preds = model(left, right)
    for h in hooks:
        h.remove()
    torch.FloatTensor = _F32
    print(kind, "fp64" if double else "fp32", f"{time.time() - t0:.1f} s", flush=True)
    out = {f"o{i}": p.detach()[..., ::4, ::4].contiguous() for i, p in enumerate(preds, 1)}
    out["lr_l"], out["lr_r"] = (t[..., ::4, ::4].contiguous() for t in cap["fe"])
    out["w9"] = cap["w9"][..., ::8, ::8].contiguous()
    for k in (1, 2, 3):
        out[f"c{k}"] = cap["c"][k][:, 0, ::2, ::4, ::4].contiguous()
    return {k: v.numpy() for k, v in out.items()}


def make(kind):
    r32, r64 = run(kind, False), run(kind, True)
    out = {}
    for k in r32:
        out[k + "_32"], out[k + "_64"] = r32[k], r64[k]
        e = np.abs(r32[k].astype(np.float64) - r64[k])
        out["e32max_" + k], out["e32mean_" + k] = e.max(), e.mean()
        print(f"  {k}: reference fp32 vs fp64 max {e.max():.3e} mean {e.mean():.3e}", flush=True)
    H, W = (576, 960) if kind == "sceneflow" else (384, 1248)
    path = os.path.join(OUT, f"g11_fullframe_{kind}_{H}x{W}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    for kind in (sys.argv[1:] or ["sceneflow", "kitti"]):
        make(kind)
