#!/usr/bin/env python3
"""Time the native 2-D conv family on the encoder's layer shapes (8 images = one batch-4 stereo step).
usage: conv2d_time.py  ->  ms and TFLOP/s (algorithmic, direct-conv FLOPs) for forward / data gradient / weight gradient"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
ops = ecm_amd.ops
LAYERS = [(8, 3, 32, 576, 960, 3, 1, 1), (8, 32, 32, 576, 960, 3, 1, 1), (8, 32, 32, 576, 960, 3, 2, 1), (8, 32, 32, 288, 480, 3, 1, 1),
          (8, 32, 64, 288, 480, 3, 2, 1), (8, 64, 64, 144, 240, 3, 1, 1), (8, 64, 128, 144, 240, 3, 1, 1), (8, 128, 128, 144, 240, 3, 1, 1),
          (8, 128, 128, 144, 240, 3, 1, 2), (8, 320, 128, 144, 240, 3, 1, 1), (8, 128, 32, 144, 240, 1, 1, 1), (4, 32, 480, 144, 240, 3, 1, 1)]


def t(fn, n=10):
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


for (B, Ci, Co, H, W, k, st, dl) in LAYERS:
    x = torch.randn(B, Ci, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, device="cuda") * 0.05).requires_grad_()
    y = ops.conv2d(x, w, st, dl)
    G = torch.randn_like(y)
    fl = 2.0 * k * k * Ci * Co * y.shape[-1] * y.shape[-2] * B
    tf = t(lambda: ops.conv2d(x.detach(), w.detach(), st, dl))
    tx = t(lambda: torch.autograd.grad(ops.conv2d(x, w.detach(), st, dl), x, G)) - tf
    tw = t(lambda: torch.autograd.grad(ops.conv2d(x.detach(), w, st, dl), w, G)) - tf
    print(f"{Ci:3d}->{Co:3d} k{k} s{st} d{dl} @{H}x{W} B{B}: fwd {tf:6.3f} ms {fl / tf / 1e9:6.1f} TF | dgrad {tx:6.3f} ms {fl / tx / 1e9:6.1f} TF | "
          f"wgrad {tw:6.3f} ms {fl / tw / 1e9:6.1f} TF", flush=True)
