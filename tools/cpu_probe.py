import os, sys, time, torch
sys.path.insert(0, os.getcwd())
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
from oracle import ecm_oracle as O
import ecm_amd
for nt in (16, 8):
    torch.set_num_threads(nt)
    model = ecm_amd.get_model("cmfsm")
    sd = {k: v.detach().clone().requires_grad_() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    left, right = torch.randn(1, 3, 256, 512, generator=g), torch.randn(1, 3, 256, 512, generator=g)
    gt = torch.rand(1, 256, 512, generator=g) * 191
    t0 = time.perf_counter()
    preds = O.cmfsm_forward(left, right, sd)
    t1 = time.perf_counter(); print(nt, "fwd", t1 - t0, flush=True)
    O.train_loss(preds, gt).backward()
    print(nt, "bwd", time.perf_counter() - t1, flush=True)
