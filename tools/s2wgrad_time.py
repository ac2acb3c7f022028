#!/usr/bin/env python3
"""Weight gradients of the stride-2 3-D layers at the step's sizes (batch 4): hourglass conv1 / conv3 and the transposed convolutions."""
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


for name, xs, Co in (("hg.conv1 32->64 s2 48x144x240", (4, 32, 48, 144, 240), 64), ("hg.conv3 64->64 s2 24x72x120", (4, 64, 24, 72, 120), 64),
                     ("hg.conv6 wgrad (x:=gy 32ch full res, Co':=64)", (4, 32, 48, 144, 240), 64),
                     ("hg.conv5 wgrad (x:=gy 64ch half res, Co':=64)", (4, 64, 24, 72, 120), 64)):
    x = torch.randn(*xs, device="cuda")
    od = [(d - 1) // 2 + 1 for d in xs[2:]]
    gy = torch.randn(xs[0], Co, *od, device="cuda")
    fl = 2.0 * 27 * xs[1] * Co * od[0] * od[1] * od[2] * xs[0]
    ms = t(lambda: ops._wgrad(x, gy, Co, xs[1], 2))
    print(f"{name:48s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF ({fl / ms / 1e9 / 157.3 * 100:.0f} % of 157.3)", flush=True)
