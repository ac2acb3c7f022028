#!/usr/bin/env python3
"""One training step (B=4, 576x960) with the weight gradients on the side stream (ops.WGRAD_OVERLAP) vs without: device status
after every phase, step time, and the gradients compared (the two must agree bit for bit: same kernels, same order per tensor)."""
import os, sys, time
os.environ.setdefault("ECM_GN_POLL_MS", "300")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ecm_amd
from importlib import import_module
ops = ecm_amd.ops
D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
model = ecm_amd.get_model("cmfsm").cuda().train()
left, right = torch.randn(B, 3, 576, 960, device="cuda"), torch.randn(B, 3, 576, 960, device="cuda")
gt = torch.rand(B, 576, 960, device="cuda") * 191


def status(tag):
    torch.cuda.synchronize()
    st = ops._lib.query("ecm_async_status", 1)
    if st:
        ops._GN_CLUSTER.clear()
        print(f"   {tag}: async status {st}", flush=True)
    return st


def step():
    for p in model.parameters():
        p.grad = None
    preds = model(left, right)
    status("forward")
    loss = D.masked_smooth_l1_x3(preds, gt)
    t0 = time.perf_counter()
    loss.backward()
    st = status("backward")
    return (time.perf_counter() - t0) * 1e3, st, float(loss)


res = {}
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?", "ECM_WGRAD_PRIO", os.environ.get("ECM_WGRAD_PRIO"))
for flag in (False, True, False, True):
    ops.WGRAD_OVERLAP = flag
    step()
    ts = [step() for _ in range(3)]
    g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).clone()
    print(f"overlap={flag}: backward ms {[round(t[0], 1) for t in ts]} status {[t[1] for t in ts]} loss {ts[-1][2]:.6f} |g| {float(g.norm()):.6e} nan {int(torch.isnan(g).sum())}", flush=True)
    if flag in res:
        pass
    res.setdefault(flag, g)
d = (res[True] - res[False]).abs().max()
print("max |g_overlap - g_serial| =", float(d))
