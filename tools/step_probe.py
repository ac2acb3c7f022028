#!/usr/bin/env python3
"""Diagnostic: time the phases of one training step (B pairs, 576x960) with progress prints."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
from importlib import import_module
D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")

def log(*a):
    print(*a, file=sys.stderr, flush=True)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (576, 960)
dev = "cuda"
model = ecm_amd.get_model("cmfsm").to(dev).train()
left, right = torch.randn(B, 3, H, W, device=dev), torch.randn(B, 3, H, W, device=dev)
gt = torch.rand(B, H, W, device=dev) * 191

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

for it in range(3):
    t0 = sync()
    lr_l, _, hr_l = model.feature_extraction(left)
    lr_r, _, _ = model.feature_extraction(right)
    t1 = sync(); log(f"it{it} encoder fwd {1e3*(t1-t0):.1f} ms")
    preds = model.hot_path(lr_l, hr_l, lr_r)
    t2 = sync(); log(f"it{it} hot path fwd {1e3*(t2-t1):.1f} ms")
    loss = D.masked_smooth_l1_x3(preds, gt)
    # hot-path backward only: grads w.r.t. the encoder outputs
    g = torch.autograd.grad(loss, [lr_l, hr_l, lr_r] + [p for n, p in model.named_parameters() if not n.startswith("feature_extraction")], retain_graph=True)
    t3 = sync(); log(f"it{it} hot path bwd {1e3*(t3-t2):.1f} ms")
    torch.autograd.backward([lr_l, hr_l, lr_r], g[:3])
    t4 = sync(); log(f"it{it} encoder bwd {1e3*(t4-t3):.1f} ms; mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    model.zero_grad()
