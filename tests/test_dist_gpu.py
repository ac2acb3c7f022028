"""The N>1 training path on the REAL model: two ranks (gloo rendezvous, both on cuda:0) each run `cmfsm` forward + backward on
their own stereo pair, `FlatBucketDDP` reduces the flat gradient bucket -- the second step through the overlapped path (the
non-encoder slice reduced on a side stream while the encoder's backward runs) -- and the result must be the gradient of the
reference's ONE masked mean over the global batch (train.py:162-174 after nn.DataParallel's gather), as a single process
computes it on the two pairs together.  Covers what the CPU gloo tests cannot: the HIP autograd Functions (incl. the forked
convolutions and the n-ary gradient sum) under the tensor hooks and side stream of the DDP wrapper."""
import os
import sys
import tempfile

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

H, W, MAXD = 256, 512, 192


def _inputs():
    g = torch.Generator().manual_seed(21)
    left, right = torch.randn(2, 3, H, W, generator=g), torch.randn(2, 3, H, W, generator=g)
    gt = torch.rand(2, H, W, generator=g) * 150.0 + 1.0
    gt[1, : H // 2] = 0.0                         # very different mask counts on the two ranks
    gt[0, :8] = 250.0
    return left, right, gt


def _model(ecm):
    torch.manual_seed(1234)
    return ecm.get_model("cmfsm").cuda().train()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from importlib import import_module
    import ecm_amd as ecm
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ecm.ops.gn_cluster_mode(0)                    # ranks share the device: two-stage GroupNorm (no inter-workgroup waits)
    model = _model(ecm)
    ddp = D.FlatBucketDDP(model, world, late_module="feature_extraction", overlap_wgrad=False)   # (ranks share the device)
    left, right, gt = (t[rank:rank + 1].cuda() for t in _inputs())
    grads, early = [], []
    for _ in range(2):                            # step 0 learns the hook count, step 1 takes the overlapped path
        ddp.zero_grad()
        loss, count = D.masked_smooth_l1_x3_with_count(model(left, right), gt, MAXD)
        ddp.global_mean_loss(loss, count).backward()
        early.append(ddp._early_work is not None)
        ddp.allreduce_gradients()
        torch.cuda.synchronize()
        ecm.ops.check_async_errors()
        grads.append(ddp.flat.detach().cpu().clone())
    order = {id(p): k for k, p in model.named_parameters()}
    torch.save({"grads": grads, "early": early, "order": [order[id(p)] for p in ddp.params]}, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_real_model_gradients_equal_global_batch():
    from importlib import import_module
    import torch.multiprocessing as mp
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import ecm_amd as ecm
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    port = 33500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "g0.pt")), torch.load(os.path.join(d, "g1.pt"))
    assert r0["early"] == [False, True] and r1["early"] == [False, True]
    # single process, both pairs as one batch, ONE masked mean -- with the same GroupNorm kernels as the workers: at random
    # initialisation the gradients of this network move by ~2 % when the rounding of ANY stage changes (two-stage vs cluster
    # GroupNorm: 2.2 %, Winograd vs direct convolutions: 6 %; ReLU / LeakyReLU kinks), so only like is compared with like
    prev_mode = ecm.ops.gn_cluster_mode(0)
    try:
        model = _model(ecm)
        left, right, gt = (t.cuda() for t in _inputs())
        loss, _ = D.masked_smooth_l1_x3_with_count(model(left, right), gt, MAXD)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ecm.ops.gn_cluster_mode(prev_mode)
    byname = dict(model.named_parameters())
    ref = torch.cat([byname[k].grad.flatten() for k in r0["order"]]).cpu()
    scale = float(ref.abs().max())
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)                                  # both ranks hold the same reduced bucket
        assert torch.isfinite(a).all()
        # two batch-1 backward passes summed vs one batch-2 pass: fp32 summation order differs in the weight gradients
        torch.testing.assert_close(a, ref, rtol=2e-3, atol=2e-5 * scale)
    assert torch.equal(r0["grads"][0], r0["grads"][1])            # plain path == overlapped path (same inputs, same weights)


# ---------------------------------------------------------------------------------------------------------------------------
# RCCL itself: a process group of ONE rank over backend "nccl" on the MI355X -- loads librccl, builds a
# communicator and runs every collective of the training step through it (parameter broadcast, the mask-count all-reduce,
# the bucket all-reduce, and on the second step the overlapped path: the non-encoder slice reduced on a side stream under
# the encoder's backward).  A sum over one rank is the identity, so the reduced bucket must equal the plain gradient bit
# for bit.
def _rccl_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from importlib import import_module
    import ecm_amd as ecm
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    model = _model(ecm)
    ddp = D.FlatBucketDDP(model, 1, late_module="feature_extraction", always_reduce=True)
    assert ddp._collective and ddp.overlap
    left, right, gt = (t[:1].cuda() for t in _inputs())
    grads, early = [], []
    for _ in range(2):
        ddp.zero_grad()
        loss, count = D.masked_smooth_l1_x3_with_count(model(left, right), gt, MAXD)
        ddp.global_mean_loss(loss, count).backward()
        early.append(ddp._early_work is not None)
        ddp.allreduce_gradients()
        torch.cuda.synchronize()
        ecm.ops.check_async_errors()
        grads.append(ddp.flat.detach().cpu().clone())
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    dist.barrier(device_ids=[0])
    with open("/proc/self/maps") as f:
        rccl = sorted({line.split()[-1] for line in f if "librccl" in line})
    order = {id(p): k for k, p in model.named_parameters()}
    torch.save({"grads": grads, "early": early, "rccl": rccl, "backend": dist.get_backend(),
                "order": [order[id(p)] for p in ddp.params], "ones": t.cpu()}, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


def test_rccl_world1_flat_bucket_allreduce():
    from importlib import import_module
    import torch.multiprocessing as mp
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import ecm_amd as ecm
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    port = 35500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rccl_worker, args=(port, d), nprocs=1, join=True)
        r = torch.load(os.path.join(d, "rccl.pt"))
    assert r["backend"] == "nccl" and r["rccl"], f"librccl was not mapped by the worker: {r['rccl']}"
    assert r["early"] == [False, True]                            # step 1 took the side-stream (overlapped) path
    assert torch.equal(r["ones"], torch.ones(4))
    model = _model(ecm)
    left, right, gt = (t[:1].cuda() for t in _inputs())
    loss, _ = D.masked_smooth_l1_x3_with_count(model(left, right), gt, MAXD)
    loss.backward()
    torch.cuda.synchronize()
    byname = dict(model.named_parameters())
    ref = torch.cat([byname[k].grad.flatten() for k in r["order"]]).cpu()
    for g in r["grads"]:
        assert torch.isfinite(g).all()
        assert torch.equal(g, ref)                                # sum over one rank = identity, plain and overlapped path


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (as the driver invokes it): the parent starts two fresh ranks,
    relays rank 0's JSON line and exits 0.  On the 1-GPU box the ranks share the device (gloo rendezvous), which the line
    must say: ranks 2, devices 1."""
    import json
    import subprocess
    env = dict(os.environ, ECM_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1",
                        "--height", "256", "--width", "512", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    ndev = torch.cuda.device_count()
    assert out["ranks"] == 2 and out["devices"] == min(2, ndev) and out["n_gpus"] == out["devices"] and out["backend"] == "gloo"
    assert out["config"]["global_batch"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
