#!/usr/bin/env python3
"""Forward time of the Winograd layers of the training step (batch 4 / 8 images); A/B runs through the environment
(ECM_WINO_PERSIST=0: one block per workgroup; ECM_WINO_LATE_PCT: start offset of a CU's second workgroup)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ecm_amd
ops = ecm_amd.ops


def t(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


out = []
for name, shape, Co in (("3D 32->32", (4, 32, 48, 144, 240), 32), ("3D 64->64", (4, 64, 24, 72, 120), 64),
                        ("3D 64->64 s", (4, 64, 12, 36, 60), 64), ("2D 32 576", (8, 32, 576, 960), 32),
                        ("2D 32 288", (8, 32, 288, 480), 32), ("2D 64", (8, 64, 144, 240), 64),
                        ("2D 128", (8, 128, 144, 240), 128), ("2D 320->128", (8, 320, 144, 240), 128),
                        ("2D 32->480", (4, 32, 144, 240), 480)):
    x = torch.randn(*shape, device="cuda")
    three = len(shape) == 5
    w = torch.randn(Co, shape[1], *((3, 3, 3) if three else (3, 3)), device="cuda") * 0.05
    fn = (lambda: ops.conv3d_k3(x, w, 1)) if three else (lambda: ops.conv2d(x, w, 1, 1))
    with torch.no_grad():
        out.append(f"{name} {t(fn):.3f}")
print(" | ".join(out), flush=True)
