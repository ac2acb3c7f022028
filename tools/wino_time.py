#!/usr/bin/env python3
"""Direct implicit-GEMM vs Winograd F(2x2,3x3) on the stride-1 3x3(x3) layers of the training step (batch 4 / 8 images).
Prints ms and algorithmic (direct-conv) TFLOP/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
ops = ecm_amd.ops


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


for name, shape, Co in (("3D 32->32 48x144x240 B4", (4, 32, 48, 144, 240), 32), ("3D 64->64 24x72x120 B4", (4, 64, 24, 72, 120), 64),
                        ("3D 64->64 12x36x60 B4", (4, 64, 12, 36, 60), 64),
                        ("2D 32->32 576x960 B8", (8, 32, 576, 960), 32), ("2D 32->32 288x480 B8", (8, 32, 288, 480), 32),
                        ("2D 64->64 144x240 B8", (8, 64, 144, 240), 64), ("2D 128->128 144x240 B8", (8, 128, 144, 240), 128),
                        ("2D 320->128 144x240 B8", (8, 320, 144, 240), 128)):
    x = torch.randn(*shape, device="cuda")
    Ci = shape[1]
    three = len(shape) == 5
    w = torch.randn(Co, Ci, *((3, 3, 3) if three else (3, 3)), device="cuda") * 0.05
    vox = 1
    for d in shape[2:]:
        vox *= d
    fl = 2.0 * (27 if three else 9) * Ci * Co * vox * shape[0]
    out = []
    for flag in (False, True):
        ops.WINOGRAD = flag
        fn = (lambda: ops.conv3d_k3(x, w, 1)) if three else (lambda: ops.conv2d(x, w, 1, 1))
        with torch.no_grad():
            ms = t(fn)
        out.append(f"{'wino  ' if flag else 'direct'} {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF")
    print(f"{name:28s} " + " | ".join(out), flush=True)

print("weight gradients:")
for name, shape, Co in (("3D 32->32 48x144x240 B4", (4, 32, 48, 144, 240), 32), ("3D 64->64 24x72x120 B4", (4, 64, 24, 72, 120), 64),
                        ("2D 32->32 576x960 B8", (8, 32, 576, 960), 32), ("2D 32->32 288x480 B8", (8, 32, 288, 480), 32),
                        ("2D 64->64 144x240 B8", (8, 64, 144, 240), 64), ("2D 128->128 144x240 B8", (8, 128, 144, 240), 128),
                        ("2D 320->128 144x240 B8", (8, 320, 144, 240), 128)):
    x = torch.randn(*shape, device="cuda")
    Ci = shape[1]
    three = len(shape) == 5
    gy = torch.randn(shape[0], Co, *shape[2:], device="cuda")
    vox = 1
    for d in shape[2:]:
        vox *= d
    fl = 2.0 * (27 if three else 9) * Ci * Co * vox * shape[0]
    out = []
    for flag in (False, True):
        if three:
            ops.WINOGRAD_WGRAD = flag
            fn = lambda: ops._wgrad(x, gy, Co, Ci, 1)
        elif flag:
            fn = lambda: ops._wino_wgrad(x, gy, Co, Ci, 1)
        else:
            w = torch.randn(Co, Ci, 3, 3, device="cuda", requires_grad=True)
            ops.WINOGRAD_WGRAD = False
            fn = lambda: torch.autograd.grad(ops.conv2d(x, w, 1, 1), w, gy)
        ms = t(fn)
        out.append(f"{'wino  ' if flag else 'direct'} {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF")
    print(f"{name:28s} " + " | ".join(out), flush=True)
