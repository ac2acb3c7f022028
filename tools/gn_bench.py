"""GroupNorm at the training batch (B=4, 32 x 48x144x240 per sample = 849 MB per tensor, far beyond the 256 MB MALL):
fused cluster kernels (ops.group_norm_act) vs the two-stage kernels (ecm_gn3d_stats + ecm_gn3d_apply)."""
import ctypes as C
import importlib
import sys

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd")
ops, _lib = pkg.ops, pkg._lib


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


def main():
    dev = torch.device("cuda:0")
    for B, Cc, dims in [(4, 32, (48, 144, 240)), (4, 64, (24, 72, 120)), (4, 64, (12, 36, 60)), (8, 32, (1, 288, 480)),
                        (8, 64, (1, 144, 240)), (8, 128, (1, 144, 240)), (8, 32, (1, 144, 240))]:
        x = torch.randn(B, Cc, *dims, device=dev)
        gm, bt = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        S = x.numel() // (B * Cc)
        mb = x.numel() * 4 / 1e9
        y = torch.empty_like(x)
        stats = torch.empty(B, 32, 2, device=dev)
        nb = _lib.query("ecm_gn3d_scratch_bytes", B, Cc, C.c_longlong(S))
        scratch = torch.empty(nb // 4 + 16, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

        def two_stage():
            _lib.call("ecm_gn3d_stats", p(x), p(stats), p(scratch), C.c_longlong(nb), B, Cc, C.c_longlong(S), C.c_float(1e-5), st)
            _lib.call("ecm_gn3d_apply", p(x), p(stats), p(gm), p(bt), None, p(y), B, Cc, C.c_longlong(S), 1, st)

        def fused():
            _lib.call("ecm_gn3d_fwd", p(x), p(gm), p(bt), None, p(y), p(stats), p(scratch), C.c_longlong(nb), B, Cc,
                      C.c_longlong(S), 1, C.c_float(1e-5), st)

        t2, tf = timeit(two_stage), timeit(fused)
        print(f"GN fwd B={B} C={Cc} {dims}: two-stage {t2:.3f} ms ({3 * mb / t2 * 1e3:.0f} GB/s of 3 passes)   "
              f"fused {tf:.3f} ms ({2 * mb / tf * 1e3:.0f} GB/s of 2 passes)")
        sk = torch.randn_like(x)

        def fused_skip():                                  # y = GroupNorm(x) + skip (the residual forms of the model): 2 reads + 1 write
            cl = ops._gn_cluster(B, dev)
            _lib.call("ecm_gn3d_fwd_p", p(x), p(gm), p(bt), p(sk), p(y), p(stats), p(scratch), C.c_longlong(nb), p(cl),
                      C.c_longlong(cl.numel()), B, Cc, C.c_longlong(S), 0, C.c_float(1e-5), st)
        ts = timeit(fused_skip)
        print(f"GN fwd + skip (cluster kernel): {ts:.3f} ms ({3 * mb / ts * 1e3:.0f} GB/s of 3 passes)")
        del sk
        xg = x.clone().requires_grad_()
        G = torch.randn_like(x)
        yy = ops.group_norm_act(xg, gm.requires_grad_(), bt.requires_grad_(), None, True)
        tb = timeit(lambda: torch.autograd.grad(yy, xg, G, retain_graph=True))
        print(f"GN bwd (fused when it fits): {tb:.3f} ms ({3 * mb / tb * 1e3:.0f} GB/s of 3 passes)")
        del x, y, xg, G, yy


if __name__ == "__main__":
    main()
