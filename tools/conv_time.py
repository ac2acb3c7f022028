"""Time one 3-D forward conv shape: conv_time.py Ci Co D H W stride [B ...]  (A/B helper for tile choices)."""
import sys, torch
sys.path.insert(0, ".")
import ecm_amd
ops = ecm_amd.ops
ci, co, D, H, W, st = (int(v) for v in sys.argv[1:7])
for B in [int(v) for v in sys.argv[7:]] or [1, 4]:
    x = torch.randn(B, ci, D, H, W, device="cuda"); w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    pk = ops._pack_conv(w)
    for _ in range(5): ops._conv_fwd(x, pk, co, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): y = ops._conv_fwd(x, pk, co, st)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"conv {ci}->{co} s{st} ({D},{H},{W}) B={B}: {ms:.3f} ms  {2*27*ci*co*y[0,0].numel()*B/ms/1e9:.1f} TFLOP/s")
