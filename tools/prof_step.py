#!/usr/bin/env python3
"""torch.profiler view of one training step (which aten ops/shapes own the GPU time outside the HIP kernels)."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
from importlib import import_module
D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda"
model = ecm_amd.get_model("cmfsm").to(dev).train()
left, right = torch.randn(B, 3, 576, 960, device=dev), torch.randn(B, 3, 576, 960, device=dev)
gt = torch.rand(B, 576, 960, device=dev) * 191
def step():
    model.zero_grad()
    D.masked_smooth_l1_x3(model(left, right), gt).backward()
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_type.name != "CPU"] if hasattr(prof.key_averages()[0], "device_type") else prof.key_averages()
rows = sorted(rows, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("total device ms", tot / 1e3)
for e in rows[:70]:
    print(f"{e.self_device_time_total / 1e3:9.2f} ms  {100 * e.self_device_time_total / tot:5.1f}%  x{e.count:4d}  {e.key[:110]}")
