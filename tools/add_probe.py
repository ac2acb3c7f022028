"""Which tensors do the ~45 elementwise adds of one training step act on?  (autograd gradient accumulation for tensors with
several consumers + the model's own adds).  Prints aten::add* calls grouped by input shape with their device time."""
import sys
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import ecm_amd  # noqa: E402
from importlib import import_module

dist = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
torch.manual_seed(0)
model = ecm_amd.get_model("cmfsm").cuda().train()
B, H, W = 4, 576, 960
left, right = torch.randn(B, 3, H, W, device="cuda"), torch.randn(B, 3, H, W, device="cuda")
gt = torch.rand(B, H, W, device="cuda") * 191
ddp = dist.FlatBucketDDP(model, 1)


def step():
    ddp.zero_grad()
    dist.masked_smooth_l1_x3(model(left, right), gt, 192).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::add")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.device_time_total / 1e3:8.3f} ms  x{e.count:3d}  {e.key:14s} {e.input_shapes}")
