"""Stride-2 3-D convolution / transposed convolution forward at the step's sizes (batch 4)."""
import sys, torch
sys.path.insert(0, ".")
import ecm_amd
ops = ecm_amd.ops
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
with torch.no_grad():
    for name, shape, Co in (("conv s2 32->64 48x144x240", (4, 32, 48, 144, 240), 64), ("conv s2 64->64 24x72x120", (4, 64, 24, 72, 120), 64)):
        x = torch.randn(*shape, device="cuda"); w = torch.randn(Co, shape[1], 3, 3, 3, device="cuda") * 0.05
        ms = t(lambda: ops.conv3d_k3(x, w, 2))
        fl = 2.0 * 27 * shape[1] * Co * (shape[2] // 2) * (shape[3] // 2) * (shape[4] // 2) * shape[0]
        print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF", flush=True)
    for name, shape, Co in (("deconv 64->32 24x72x120", (4, 64, 24, 72, 120), 32), ("deconv 64->64 12x36x60", (4, 64, 12, 36, 60), 64)):
        x = torch.randn(*shape, device="cuda"); w = torch.randn(shape[1], Co, 3, 3, 3, device="cuda") * 0.05
        ms = t(lambda: ops.deconv3d_k3s2(x, w))
        fl = 2.0 * 27 * shape[1] * Co * shape[2] * shape[3] * shape[4] * shape[0]
        print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF", flush=True)
    # direct stride-1 kernel too (ECM_WINOGRAD=0 path)
    ops.WINOGRAD = False
    x = torch.randn(4, 32, 48, 144, 240, device="cuda"); w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05
    ms = t(lambda: ops.conv3d_k3(x, w, 1))
    print(f"direct s1 32->32: {ms:.3f} ms", flush=True)
