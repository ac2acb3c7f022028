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
    ddp = D.FlatBucketDDP(model, world, late_module="feature_extraction")
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
