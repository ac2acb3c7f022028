#!/usr/bin/env python3
"""Classifier 32 -> 1 layer (conv3d_c1.hip) at the training batch: forward, data gradient, weight gradient; HIP events."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
ops = ecm_amd.ops


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for B in (1, 4):
    x = torch.randn(B, 32, 48, 144, 240, device="cuda")
    w = torch.randn(1, 32, 3, 3, 3, device="cuda") * 0.05
    y = ops.conv3d_k3(x, w, 1)
    ref = F.conv3d(x[:1, :, :8], w, None, 1, 1)
    err = float((y[:1, :, 1:7] - ref[:, :, 1:7]).abs().max())
    gb = x.numel() * 4 / 1e6
    f = t(lambda: ops.conv3d_k3(x, w, 1))
    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    yy = ops.conv3d_k3(xg, wg, 1)
    g = torch.randn_like(yy)
    d = t(lambda: torch.autograd.grad(yy, xg, g, retain_graph=True))
    wgt = t(lambda: torch.autograd.grad(yy, wg, g, retain_graph=True))
    print(f"B={B}: fwd {f:.3f} ms ({gb / f:.0f} GB/s)  dgrad {d:.3f} ms ({gb / d:.0f} GB/s)  wgrad {wgt:.3f} ms ({gb / wgt:.0f} GB/s)  max|err| {err:.2e}", flush=True)

# round 4: the fused tail (GroupNorm + ReLU on load) against GroupNorm kernel + 32 -> 1 convolution
import torch as _t
x = _t.randn(4, 32, 48, 144, 240, device="cuda")
gm, bt = _t.ones(32, device="cuda"), _t.zeros(32, device="cuda")
w1 = _t.randn(1, 32, 3, 3, 3, device="cuda") * 0.05
def _tm(fn, n=10):
    for _ in range(3): fn()
    _t.cuda.synchronize()
    s, e = _t.cuda.Event(enable_timing=True), _t.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); _t.cuda.synchronize()
    return s.elapsed_time(e) / n
with _t.no_grad():
    print(f"B=4 classifier tail: GroupNorm kernel + c1 conv {_tm(lambda: ops.conv3d_k3(ops.group_norm_act(x, gm, bt, None, True), w1, 1)):.3f} ms   "
          f"fused (stats + normalise on load) {_tm(lambda: ops.classifier_tail(x, gm, bt, w1)):.3f} ms")
