"""Per-shape MIOpen fp32 conv2d timings (fwd / dgrad / wgrad) for the encoder's layers at the training batch
(8 images of 576x960): the baseline a native 2-D MFMA conv has to beat."""
import importlib
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
ops = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd").ops

torch.backends.cudnn.benchmark = False
dev = torch.device("cuda:0")
SHAPES = [  # Ci, Co, k, stride, dil, (H, W) of the INPUT, count per image-forward
    (32, 32, 3, 1, 1, (576, 960), 6), (32, 32, 3, 2, 1, (576, 960), 1), (32, 32, 3, 1, 1, (288, 480), 7),
    (32, 64, 3, 2, 1, (288, 480), 1), (64, 64, 3, 1, 1, (144, 240), 31), (64, 128, 3, 1, 1, (144, 240), 1),
    (128, 128, 3, 1, 1, (144, 240), 5), (128, 128, 3, 1, 2, (144, 240), 6), (320, 128, 3, 1, 1, (144, 240), 1),
]


def timeit(fn, iters=5, warm=2):
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


tot = [0.0, 0.0, 0.0]
for ci, co, k, st, dil, (H, W), cnt in SHAPES:
    x = torch.randn(8, ci, H, W, device=dev, requires_grad=True)
    w = torch.randn(co, ci, k, k, device=dev, requires_grad=True) * 0.05
    pad = dil
    y = F.conv2d(x, w, None, st, pad, dil)
    g = torch.randn_like(y)
    gf = 2.0 * ci * co * k * k * y.shape[2] * y.shape[3] * 8 / 1e9
    tf = timeit(lambda: F.conv2d(x, w, None, st, pad, dil))
    td = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    tw = timeit(lambda: torch.autograd.grad(y, w, g, retain_graph=True))
    tot[0] += tf * cnt; tot[1] += td * cnt; tot[2] += tw * cnt
    mine = ""
    if k == 3 and st == 1 and dil == 1 and ci in (32, 64) and co in (32, 64):
        yy = ops.conv2d_k3(x.detach(), w)          # x without grad: time the weight gradient alone
        th = timeit(lambda: torch.autograd.grad(yy, w, g, retain_graph=True))
        pk, pkt = ops._pack_conv2d(w.detach()), ops._pack_conv2d(w.detach(), True)
        tff = timeit(lambda: ops._conv2d_fwd(x.detach(), pk, co))
        tdd = timeit(lambda: ops._conv2d_fwd(g, pkt, ci))
        mine = f" | HIP fwd {tff:6.3f} ms {gf / tff:6.1f} TF, dgrad {tdd:6.3f} ms {gf / tdd:6.1f} TF, wgrad {th:6.3f} ms {gf / th:6.1f} TF"
    print(f"{ci:3d}->{co:3d} s{st} d{dil} {H}x{W} x{cnt:2d}: fwd {tf:6.3f} ms {gf / tf:6.1f} TF | dgrad {td:6.3f} ms {gf / td:6.1f} TF"
          f" | wgrad {tw:6.3f} ms {gf / tw:6.1f} TF" + mine, flush=True)
    del x, w, y, g
print(f"weighted totals per step (B=8 images): fwd {tot[0]:.1f} ms, dgrad {tot[1]:.1f} ms, wgrad {tot[2]:.1f} ms")
