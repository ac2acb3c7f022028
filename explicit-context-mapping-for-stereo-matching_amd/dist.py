"""Data-parallel harness for the training path: one process per GPU, stereo pairs sharded across ranks, ONE
flat fp32 gradient bucket all-reduced per step (RCCL over xGMI when the backend is "nccl"; gloo on CPU tests).

Replaces the reference's single-process `nn.DataParallel` (train.py:78-81), which re-broadcasts the 21 MB of
parameters and reduces gradients onto device 0 every step.  GroupNorm has no cross-sample statistics, so the
gradient sum is the only exchange on the path.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, device: int | None = None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*). Returns (rank, world, local_rank).
    `device`: the GPU index this rank uses (default LOCAL_RANK); with "nccl" the process group is bound to it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            dev = local if device is None else device
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)      # eager communicator on the right GPU; barrier() needs no guess
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(n_items: int, rank: int, world: int):
    """Contiguous index split of a global batch (DistributedSampler-style, no shuffling): rank -> range."""
    per = n_items // world
    rem = n_items % world
    start = rank * per + min(rank, rem)
    return range(start, start + per + (1 if rank < rem else 0))


class FlatBucketDDP:
    """Owns one flat gradient buffer; after `allreduce_gradients()` every parameter's .grad is a view into it, so the whole
    model's gradient is exchanged with a single all-reduce (5,255,368 floats = 21 MB for cmfsm).

    `zero_grad()` drops the gradients (None) instead of zeroing the bucket: autograd then hands each parameter its freshly
    computed gradient tensor instead of launching one `grad += new` kernel per parameter (≈280 tiny adds per step for
    cmfsm), and `allreduce_gradients()` gathers them into the bucket with ONE multi-tensor copy."""

    def __init__(self, module: torch.nn.Module, world: int | None = None):
        self.module = module
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        for p, v in zip(self.params, self.views):
            p.grad = v
        if self.world > 1:
            self.broadcast_parameters()

    def broadcast_parameters(self, src: int = 0):
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def allreduce_gradients(self):
        """Gather the step's gradients into the bucket, sum over ranks, average (= the gradient of the mean loss over
        the global batch), and leave every .grad pointing at its slice of the bucket for the optimizer."""
        dst, src, unused = [], [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                unused.append(v)                      # parameter not reached by this step's graph
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        if unused:
            torch._foreach_zero_(unused)
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def __call__(self, *a, **k):
        return self.module(*a, **k)


def masked_smooth_l1_x3(preds, gt, maxdisp: int = 192):
    """The reference's training loss (train.py:162,168-174): mask 0<d<maxdisp, smooth-L1 'mean' over the masked pixels
    of each head, weights 0.5/0.7/1.0 -- the fused HIP loss kernel (ops.stereo_loss3: one pass, no `o[mask]` gathers and
    their six device->host syncs per step).  GPU tensors only, like every op of the package (no CPU fallback)."""
    from . import ops
    return ops.stereo_loss3(preds, gt, maxdisp)[0]


class GraphedForward:
    """Inference through one captured HIP graph: the eval forward of a model is a fixed sequence of ~400 kernel launches
    (encoder on MIOpen + the hot path's kernels), which at batch 1 is partly launch-bound; capturing it once and replaying
    it removes the per-launch host cost.  Inputs are copied into the graph's static buffers; the returned tensors are the
    graph's static outputs (valid until the next call).  Shapes are fixed at construction."""

    def __init__(self, model: torch.nn.Module, left: torch.Tensor, right: torch.Tensor, warmup: int = 3):
        assert left.is_cuda and right.is_cuda, "GraphedForward captures a HIP graph: GPU tensors only"
        self.model = model.eval()
        self.left, self.right = left.clone(), right.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():      # warm-up outside capture: lazy one-time work (MIOpen solver
            for _ in range(warmup):                          # picks, LDS attributes, occupancy queries, constant masks)
                self.model(self.left, self.right)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self.model(self.left, self.right)

    def __call__(self, left: torch.Tensor, right: torch.Tensor):
        self.left.copy_(left)
        self.right.copy_(right)
        self.graph.replay()
        return self.out
