#!/usr/bin/env python3
"""Per-kernel timing at the SceneFlow shapes (576x960, D=192 -> 1/4-res volume [48,144,240]) on one MI355X.
HIP-event timing on torch's current stream (the stream the kernels are launched on)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd  # noqa: E402

ops = ecm_amd.ops
dev = "cuda"


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


def report(name, ms, gbytes=None, gflop=None, executed=1.0):
    """executed: fraction of the direct-count FLOPs the kernel issues on the matrix cores (Winograd F(2x2,3x3) x direct
    depth: 12/27 in 3-D, 16/36 in 2-D); the roofline fraction is on EXECUTED work, the direct count is printed beside it."""
    extra = ""
    if gbytes:
        extra += f"  {gbytes / ms:8.1f} GB/s ({gbytes / ms / 8000 * 100:.1f}% of 8 TB/s)"
    if gflop and executed != 1.0:
        extra += (f"  {gflop * executed / ms:8.2f} TFLOP/s executed ({gflop * executed / ms / 157.3 * 100:.1f}% of 157.3)"
                  f"  [{gflop / ms:7.2f} TF direct-count equivalent]")
    elif gflop:
        extra += f"  {gflop / ms:8.2f} TFLOP/s ({gflop / ms / 157.3 * 100:.1f}% of 157.3)"
    print(f"{name:44s} {ms:9.3f} ms{extra}", flush=True)


def main():
    torch.manual_seed(0)
    B, h, w, D = 1, 144, 240, 48
    L, R = torch.randn(B, 32, h, w, device=dev), torch.randn(B, 32, h, w, device=dev)
    cost = ops.cost_volume(L, R, D)
    vol_mb = cost.numel() * 4 / 1e6
    report("costvol fwd 576x960 D=192", timeit(lambda: ops.cost_volume(L, R, D)), (vol_mb + 2 * L.numel() * 4 / 1e6))
    g = torch.randn_like(cost)
    report("costvol bwd 576x960", timeit(lambda: ops.CostVolumeConcat.backward(type("c", (), {"dims": (B, 32, h, w, D)}), g)),
           (vol_mb + 2 * L.numel() * 4 / 1e6))
    del g
    L5, R5 = torch.randn(1, 32, 270, 480, device=dev), torch.randn(1, 32, 270, 480, device=dev)
    c5 = ops.cost_volume(L5, R5, 64)
    report("costvol fwd 1080x1920 D=256 (cfg5)", timeit(lambda: ops.cost_volume(L5, R5, 64)),
           (c5.numel() * 4 + 2 * L5.numel() * 4) / 1e6)
    del c5, L5, R5
    torch.cuda.empty_cache()

    V = D * h * w
    for (ci, co, dims, st, label) in [(64, 32, (48, 144, 240), 1, "dres0.0"), (32, 32, (48, 144, 240), 1, "conv 32->32"),
                                      (32, 64, (48, 144, 240), 2, "hg.conv1 s2"), (64, 64, (24, 72, 120), 1, "hg.conv2"),
                                      (64, 64, (24, 72, 120), 2, "hg.conv3 s2"), (64, 64, (12, 36, 60), 1, "hg.conv4"),
                                      (32, 1, (48, 144, 240), 1, "classif.2 (Co=1)")]:
        x = torch.randn(B, ci, *dims, device=dev)
        wt = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
        pk = ops._pack_conv(wt)
        od = [(d - 1) // st + 1 for d in dims]
        gf = 2.0 * 27 * ci * co * od[0] * od[1] * od[2] * B / 1e9
        if co == 1:
            report(f"conv3d {label} {ci}->{co} s{st} {dims}", timeit(lambda: ops.conv3d_k3(x, wt, st)), gflop=gf,
                   gbytes=(x.numel() + x.numel() // ci) * 4 / 1e6)
            gyc = torch.randn(B, 1, *dims, device=dev)
            xg = x.clone().requires_grad_(False)
            report(f"wgrad  {label} {ci}->{co}", timeit(lambda: ops.Conv3dK3.backward(
                type("c", (), {"saved_tensors": (xg, wt), "stride": 1, "needs_input_grad": (False, True, False)}), gyc)),
                gbytes=(x.numel() + x.numel() // ci) * 4 / 1e6)
            continue
        report(f"conv3d {label} {ci}->{co} s{st} {dims} (direct)", timeit(lambda: ops._conv_fwd(x, pk, co, st)), gflop=gf)
        if st == 1 and ops.WINOGRAD:
            pkw = ops._wino_pack(wt, 3, False)
            report(f"conv3d {label} {ci}->{co} s{st} {dims} (Winograd)", timeit(lambda: ops._wino_run(x, pkw, co, 3)), gflop=gf,
                   executed=12.0 / 27.0)
    for (ci, co, dims, label) in [(64, 64, (12, 36, 60), "hg.conv5"), (64, 32, (24, 72, 120), "hg.conv6")]:
        x = torch.randn(B, ci, *dims, device=dev)
        wt = torch.randn(ci, co, 3, 3, 3, device=dev) * 0.05
        pk = ops._pack_deconv(wt)
        gf = 2.0 * 27 * ci * co * dims[0] * dims[1] * dims[2] * B / 1e9
        report(f"deconv3d {label} {ci}->{co} {dims}", timeit(lambda: ops._deconv_fwd(x, pk, co, [2 * d for d in dims])), gflop=gf)

    for (ci, co, dims, st, label) in [(32, 32, (48, 144, 240), 1, "wgrad 32->32"), (64, 32, (48, 144, 240), 1, "wgrad dres0.0"),
                                      (32, 64, (48, 144, 240), 2, "wgrad hg.conv1 s2"), (64, 64, (24, 72, 120), 1, "wgrad hg.conv2"),
                                      (64, 64, (24, 72, 120), 2, "wgrad hg.conv3 s2"), (64, 64, (12, 36, 60), 1, "wgrad hg.conv4")]:
        x = torch.randn(B, ci, *dims, device=dev)
        od = [(d - 1) // st + 1 for d in dims]
        gy = torch.randn(B, co, *od, device=dev)
        gf = 2.0 * 27 * ci * co * od[0] * od[1] * od[2] * B / 1e9
        wino = st == 1 and ops.WINOGRAD and ops.WINOGRAD_WGRAD
        report(f"{label} {ci}->{co} s{st} {dims}" + (" (Winograd)" if wino else ""), timeit(lambda: ops._wgrad(x, gy, co, ci, st)),
               gflop=gf, executed=12.0 / 27.0 if wino else 1.0)
    del x, gy
    x = torch.randn(B, 32, D, h, w, device=dev)
    gm, bt = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    mb = x.numel() * 4 / 1e6
    # fused cluster kernel: 1 read + 1 write (+1 read with skip); at B=1 the 212 MB volume partly lives in the MALL
    report("gn fused fwd (relu) 32ch volume", timeit(lambda: ops.group_norm_act(x, gm, bt, None, True)), 2 * mb)
    report("gn fused fwd (relu,+skip)", timeit(lambda: ops.group_norm_act(x, gm, bt, x, True)), 3 * mb)

    lr, hr = torch.randn(B, 32, h, w, device=dev), torch.randn(B, 32, 4 * h, 4 * w, device=dev)
    W0, W1, W2, W3 = (torch.randn(*s, device=dev) * 0.2 for s in ((32, 66, 1, 1), (16, 32, 1, 1), (8, 16, 1, 1), (1, 8, 1, 1)))
    report("ecm_weights9 fwd 576x960", timeit(lambda: ops.ecm_weights9(lr, hr, W0, W1, W2, W3)),
           (hr.numel() + lr.numel() + 9 * 576 * 960) * 4 / 1e6, gflop=7.7)
    lrg, hrg = lr.clone().requires_grad_(), hr.clone().requires_grad_()
    Wg = [t.clone().requires_grad_() for t in (W0, W1, W2, W3)]
    w9o = ops.ecm_weights9(lrg, hrg, *Wg)
    g9 = torch.randn_like(w9o)
    report("ecm_weights9 bwd 576x960", timeit(lambda: torch.autograd.grad(w9o, [lrg, hrg] + Wg, g9, retain_graph=True)),
           (2 * hr.numel() + 2 * lr.numel() + 18 * 576 * 960) * 4 / 1e6)
    c = torch.randn(3, B, D, h, w, device=dev)
    report("softargmin 3 heads", timeit(lambda: ops.softargmin_heads(c)), c.numel() * 4 / 1e6)
    d = ops.softargmin_heads(c)
    w9 = torch.softmax(torch.randn(B, 9, 4 * h, 4 * w, device=dev), 1)
    report("aggregate9 3 heads", timeit(lambda: ops.ecm_aggregate9(d, w9, 4)), (12 * 576 * 960) * 4 / 1e6)

    model = ecm_amd.get_model("cmfsm").cuda().eval()
    left, right = torch.randn(1, 3, 576, 960, device=dev), torch.randn(1, 3, 576, 960, device=dev)
    with torch.no_grad():
        lr_l, _, hr_l = model.feature_extraction(left)
        lr_r, _, _ = model.feature_extraction(right)
        report("encoder x2 (native 2-D family + GroupNorm; pooling / cat on ATen)", timeit(lambda: (model.feature_extraction(left), model.feature_extraction(right)), 5, 2))
        # 1,069 GFLOP is the reference's op count; the collapsed first conv executes 183 -> ~16 of them,
        # so the rate below is "reference-equivalent", the executed rate is ~16 % lower
        report("hot path fwd (HIP), reference-equivalent FLOPs", timeit(lambda: model.hot_path(lr_l, hr_l, lr_r), 5, 2), gflop=1069)
        report("full cmfsm fwd 576x960", timeit(lambda: model(left, right), 5, 2))


if __name__ == "__main__":
    main()
