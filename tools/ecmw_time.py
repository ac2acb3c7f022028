#!/usr/bin/env python3
"""ecm_weights9 forward / backward timing at 576x960 (B = 1 and 4), HIP events on the launch stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd  # noqa: E402

ops = ecm_amd.ops


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for B in (1, 4):
    h, w = 144, 240
    lr, hr = torch.randn(B, 32, h, w, device="cuda"), torch.randn(B, 32, 4 * h, 4 * w, device="cuda")
    Ws = [torch.randn(*s, device="cuda") * 0.2 for s in ((32, 66, 1, 1), (16, 32, 1, 1), (8, 16, 1, 1), (1, 8, 1, 1))]
    a = [t.clone().requires_grad_() for t in (lr, hr, *Ws)]
    w9 = ops.ecm_weights9(*a)
    g9 = torch.randn_like(w9)
    f = timeit(lambda: ops.ecm_weights9(lr, hr, *Ws))
    b = timeit(lambda: torch.autograd.grad(w9, a, g9, retain_graph=True))
    print(f"B={B}: ecm_weights9 fwd {f:.3f} ms  bwd {b:.3f} ms", flush=True)
