"""A/B of a CU-masked weight-gradient stream, in ONE process (round 4, second take).  The first sweep (one bench process per mask,
back to back) was confounded: an un-paced training loop went to the driver for memory inside the timed region, and a process
that starts right after another has exited gets that memory slowly ("host run-ahead").  Here the side
stream of ops._on_side is swapped for one created with hipExtStreamCreateWithCUMask between measurements of the same model.
Usage (GPU box): python tools/experiments/cumask_ab.py [batch]"""
import ctypes as C
import os
import sys
import time
from importlib import import_module

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ecm_amd                                                    # noqa: E402

ops = ecm_amd.ops
D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = ecm_amd.get_model("cmfsm").to(dev).train()
ddp = D.FlatBucketDDP(model, 1)
opt = torch.optim.Adam(ddp.params, lr=1e-3, fused=True)
left, right = torch.randn(B, 3, 576, 960, device=dev), torch.randn(B, 3, 576, 960, device=dev)
gt = torch.rand(B, 576, 960, device=dev) * 191
hip = C.CDLL("libamdhip64.so")


def masked_stream(n):
    words = 8                                                     # 256 CUs
    mask = (C.c_uint32 * words)(*[(0xFFFFFFFF if n >= 32 * (i + 1) else ((1 << max(0, n - 32 * i)) - 1)) for i in range(words)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def step():
    ddp.zero_grad()
    loss, count = D.masked_smooth_l1_x3_with_count(model(left, right), gt, 192)
    ddp.global_mean_loss(loss, count).backward()
    ddp.allreduce_gradients()
    opt.step()


def measure(tag, n=12):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    print(f"{tag:28s} {ms:8.2f} ms/step   ({torch.cuda.memory_stats()['num_device_alloc'] - a0} device allocations while timed)", flush=True)


measure("ordinary side stream")
for n in (224, 192, 128, 64):
    torch.cuda.synchronize()
    ops._SIDE[0] = [masked_stream(n), False, set(), None]
    measure(f"side stream on {n} CUs")
torch.cuda.synchronize()
ops._SIDE[0] = [torch.cuda.Stream(device=dev), False, set(), None]
measure("ordinary side stream again")
prev = ops.enable_wgrad_overlap(False)
measure("no side stream")
