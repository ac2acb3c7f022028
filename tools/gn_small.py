#!/usr/bin/env python3
"""GroupNorm forward + backward on ONE mid-size encoder map (default 8 x 64 x 144 x 240), looped: for rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
ops = ecm_amd.ops
B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 64, 144, 240)
x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
gm, bt = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
g = torch.randn(B, C, H, W, device="cuda")
for _ in range(50):
    y = ops.group_norm_act(x, gm, bt, None, True)
    y.backward(g)
torch.cuda.synchronize()
