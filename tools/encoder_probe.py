#!/usr/bin/env python3
"""Which encoder conv shapes fall onto slow MIOpen kernels?  Times every distinct Conv2d config fwd and bwd."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda"
fe = ecm_amd.models.feature_extraction().to(dev)
seen = {}
def hook(m, inp, out):
    key = (m.in_channels, m.out_channels, m.kernel_size, m.stride, m.dilation, tuple(inp[0].shape))
    seen.setdefault(key, 0); seen[key] += 1
for m in fe.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.register_forward_hook(hook)
x = torch.randn(B, 3, 576, 960, device=dev)
with torch.no_grad():
    fe(x)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tot_f = tot_b = 0
for key, cnt in sorted(seen.items(), key=lambda kv: -kv[1]):
    ci, co, k, st, dil, shp = key
    conv = torch.nn.Conv2d(ci, co, k, st, padding=(dil[0] if k[0] == 3 else 0), dilation=dil, bias=False).to(dev)
    inp = torch.randn(*shp, device=dev, requires_grad=True)
    out = conv(inp); g = torch.randn_like(out)
    f = t(lambda: conv(inp))
    def fb():
        o = conv(inp); o.backward(g)
    b = t(fb) - f
    tot_f += f * cnt; tot_b += b * cnt
    print(f"x{cnt:2d} {ci:3d}->{co:3d} k{k[0]} s{st[0]} d{dil[0]} in{tuple(shp)}  fwd {f:8.2f} ms  bwd {b:8.2f} ms", flush=True)
print("sum fwd", tot_f, "sum bwd", tot_b)
