#!/usr/bin/env python3
"""Launch a few conv kernels only (for rocprofv3 --pmc passes). usage: conv_only.py [conv|wino|wino2d|wino2d64|wino2d128|c1gn|wgrad|wino_wgrad|gn|ecmw_bwd|costvol] [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
ops = ecm_amd.ops
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
dev = "cuda"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if which == "conv":
    x = torch.randn(B, 32, 48, 144, 240, device=dev); w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
    pk = ops._pack_conv(w)
    for _ in range(5): y = ops._conv_fwd(x, pk, 32, 1)
elif which == "wino":
    x = torch.randn(B, 32, 48, 144, 240, device=dev); w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
    pk = ops._wino_pack(w, 3, False)
    for _ in range(5): y = ops._wino_run(x, pk, 32, 3)
elif which in ("wino2d", "wino2d64", "wino2d128"):
    Cc, H, W = {"wino2d": (32, 576, 960), "wino2d64": (64, 144, 240), "wino2d128": (128, 144, 240)}[which]
    x = torch.randn(B, Cc, H, W, device=dev); w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
    pk = ops._wino_pack(w, 1, False)
    for _ in range(5): y = ops._wino_run(x, pk, Cc, 1)
elif which == "wino_wgrad":
    x = torch.randn(B, 32, 48, 144, 240, device=dev); gy = torch.randn(B, 32, 48, 144, 240, device=dev)
    for _ in range(5): g = ops._wino_wgrad(x, gy, 32, 32, 3)
elif which == "wgrad":
    x = torch.randn(B, 32, 48, 144, 240, device=dev); gy = torch.randn(B, 32, 48, 144, 240, device=dev)
    ops.WINOGRAD_WGRAD = False
    for _ in range(5): g = ops._wgrad(x, gy, 32, 32, 1)
elif which == "gn":
    x = torch.randn(B, 32, 48, 144, 240, device=dev, requires_grad=True)
    gm, bt = torch.ones(32, device=dev, requires_grad=True), torch.zeros(32, device=dev, requires_grad=True)
    G = torch.randn(B, 32, 48, 144, 240, device=dev)
    for _ in range(3):
        y = ops.group_norm_act(x, gm, bt, None, True)
        torch.autograd.grad(y, x, G)
elif which == "ecmw_bwd":
    h, w = 144, 240
    lr, hr = torch.randn(B, 32, h, w, device=dev), torch.randn(B, 32, 4 * h, 4 * w, device=dev)
    Ws = [torch.randn(*s, device=dev) * 0.2 for s in ((32, 66, 1, 1), (16, 32, 1, 1), (8, 16, 1, 1), (1, 8, 1, 1))]
    a = [t.clone().requires_grad_() for t in (lr, hr, *Ws)]
    w9 = ops.ecm_weights9(*a)
    g9 = torch.randn_like(w9)
    for _ in range(5): torch.autograd.grad(w9, a, g9, retain_graph=True)
elif which == "c1gn":                     # the fused classifier tail: GroupNorm statistics + normalise-on-load 32 -> 1 convolution
    x = torch.randn(B, 32, 48, 144, 240, device=dev); w = torch.randn(1, 32, 3, 3, 3, device=dev) * 0.05
    gm, bt = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    with torch.no_grad():
        for _ in range(5): y = ops.classifier_tail(x, gm, bt, w)
elif which == "costvol":
    L, R = torch.randn(B, 32, 144, 240, device=dev), torch.randn(B, 32, 144, 240, device=dev)
    for _ in range(5): c = ops.cost_volume(L, R, 48)
torch.cuda.synchronize()
