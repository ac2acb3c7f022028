"""Phase profile of the GroupNorm cluster forward kernel: wall ticks (100 MHz) that thread 0 of every workgroup spends (1) loading
and reducing its slab, (2) in the rendezvous (publish its partial, collect the cluster's), (3) in the finishing pass (apply,
store, bookkeeping), summed over all workgroups.  Needs a library whose gn3d.hip was compiled with -DGN_PROFILE (adds the
counters and the `ecm_gn3d_profile` entry; NOT part of the shipped library):
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGN_PROFILE -c csrc/gn3d.hip -o /tmp/gn3d_prof.o
  hipcc --offload-arch=gfx950 -shared -fPIC <the other csrc/*.o> /tmp/gn3d_prof.o -o /tmp/libecm_prof.so
  ECM_HIP_LIB=/tmp/libecm_prof.so python tools/gn_phase_profile.py"""
import sys, ctypes as C, torch
sys.path.insert(0, ".")
import ecm_amd
ops = ecm_amd.ops; _lib = ops._lib; lib = _lib.load()
lib.ecm_gn3d_profile.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
def read(reset=1):
    a = (C.c_ulonglong * 4)(); assert lib.ecm_gn3d_profile(a, reset) == 0; return list(a)
for (B, Cc, dims, skip) in [(4, 32, (48, 144, 240), False), (4, 32, (48, 144, 240), True), (4, 64, (24, 72, 120), False), (8, 64, (144, 240), True), (8, 128, (144, 240), False)]:
    x = torch.randn(B, Cc, *dims, device="cuda"); gm, bt = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    sk = torch.randn_like(x) if skip else None
    with torch.no_grad():
        for _ in range(3): ops.group_norm_act(x, gm, bt, sk, not skip)
        torch.cuda.synchronize(); read(1)
        n = 10
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): ops.group_norm_act(x, gm, bt, sk, not skip)
        e.record(); torch.cuda.synchronize()
    p = read(1)
    tot = p[0] + p[1] + p[2]
    print(f"B={B} C={Cc} {dims} skip={skip}: {s.elapsed_time(e)/n:.3f} ms/launch, {p[3]/n:.0f} tickets/launch; per ticket (us): load+reduce {p[0]/p[3]/100:.2f}  rendezvous {p[1]/p[3]/100:.2f}  finish+store {p[2]/p[3]/100:.2f}   shares {p[0]/tot:.2f} / {p[1]/tot:.2f} / {p[2]/tot:.2f}")
