#!/usr/bin/env python3
"""Does a GroupNorm cluster kernel on the main stream survive a weight-gradient kernel running on a second stream?
(ECM_GN_POLL_MS=200 keeps a failure short.)  Prints per pairing: status, time alone / together."""
import os, sys, time
os.environ.setdefault("ECM_GN_POLL_MS", "200")
os.environ["ECM_WGRAD_OVERLAP"] = "0"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ecm_amd
ops = ecm_amd.ops
side = torch.cuda.Stream()


def ms(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def status():
    torch.cuda.synchronize()
    return ops._lib.query("ecm_async_status", 1)


cases = {
    "wino3d 32->32": (lambda: ops._wino_wgrad_now(x3, g3, 32, 32, 3)),
    "s2 wgrad 32->64": (lambda: ops._wgrad_direct(x3, g3s, 64, 32, 2)),
    "wino2d 128": (lambda: ops._wino_wgrad_now(x2, g2, 128, 128, 1)),
}
x3 = torch.randn(4, 32, 48, 144, 240, device="cuda"); g3 = torch.randn_like(x3)
g3s = torch.randn(4, 64, 24, 72, 120, device="cuda")
x2 = torch.randn(8, 128, 144, 240, device="cuda"); g2 = torch.randn_like(x2)
gam = torch.ones(32, device="cuda"); bet = torch.zeros(32, device="cuda")
xg = torch.randn(4, 32, 48, 144, 240, device="cuda", requires_grad=True)
gam2 = torch.ones(128, device="cuda"); xg2 = torch.randn(8, 128, 144, 240, device="cuda", requires_grad=True)


def gn_fb(x, g, b):
    y = ops.group_norm_act(x, g, b, None, True)
    y.backward(torch.ones_like(y))
    x.grad = None


for gname, gfn in (("gn 3-D full", lambda: gn_fb(xg, gam, bet)), ("gn 2-D 128ch", lambda: gn_fb(xg2, gam2, torch.zeros(128, device="cuda")))):
    t_gn = ms(gfn)
    print(f"{gname}: alone {t_gn:.3f} ms  status {status()}", flush=True)
    for wname, wfn in cases.items():
        t_w = ms(wfn)

        def both():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    wfn()
            gfn()
            torch.cuda.current_stream().wait_stream(side)
        t_b = ms(both, 3)
        st = status()
        if st:
            ops._GN_CLUSTER.clear()
        print(f"   with 3 x {wname}: wgrad alone {t_w:.3f} ms each, together {t_b:.3f} ms (serial would be {3 * t_w + t_gn:.3f})  status {st}", flush=True)
