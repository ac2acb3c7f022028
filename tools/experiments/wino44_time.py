#!/usr/bin/env python3
"""2-D 3x3 layers of the training step: Winograd F(4x4) (conv_wino44.hip) vs F(2x2) (conv_wino.hip) vs direct, 8 images."""
import os, sys
import torch
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


for name, shape, Co in (("32->32 576x960", (8, 32, 576, 960), 32), ("32->32 288x480", (8, 32, 288, 480), 32), ("64->64 144x240", (8, 64, 144, 240), 64),
                        ("128->128 144x240", (8, 128, 144, 240), 128), ("320->128 144x240", (8, 320, 144, 240), 128)):
    x = torch.randn(*shape, device="cuda")
    Ci = shape[1]
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    fl = 2.0 * 9 * Ci * Co * shape[2] * shape[3] * shape[0]
    p44, p22, pd = ops._wino44_pack(w, False), ops._wino_pack(w, 1, False), ops._pack2d(w, False)
    a = t(lambda: ops._wino44_run(x, p44, Co))
    b = t(lambda: ops._wino_run(x, p22, Co, 1))
    c = t(lambda: ops._conv2d_run(x, pd, Co, 3, 3, 1, 1, 1, 1, shape[2], shape[3]))
    print(f"{name:20s} F(4x4) {a:7.3f} ms ({fl / a / 1e9:6.1f} TF direct-count) | F(2x2) {b:7.3f} ms ({fl / b / 1e9:6.1f}) | direct {c:7.3f} ms ({fl / c / 1e9:6.1f})", flush=True)
