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
    model's gradient is exchanged with one all-reduce per bucket slice (5,255,368 floats = 21 MB for cmfsm).

    `zero_grad()` drops the gradients (None) instead of zeroing the bucket: autograd then hands each parameter its freshly
    computed gradient tensor instead of launching one `grad += new` kernel per parameter (≈280 tiny adds per step for
    cmfsm), and the gradients are gathered into the bucket with ONE multi-tensor copy per slice.

    Overlap: backward finishes the hot path (3-D stack, ECM, heads) before it enters the 2-D encoder, so with
    `late_module` = the encoder's attribute name the bucket is split into the encoder's slice and the rest; as soon as the
    gradient has flowed back into ALL of the encoder's outputs (tensor hooks; their number is learnt on the first step) the
    rest-slice is gathered and its all-reduce is started on a side stream, under the encoder's backward.  The encoder slice
    follows at the end of backward.  world == 1 installs nothing.

    Loss semantics: the reference's loss is ONE masked mean over the whole gathered batch (train.py:162-174 behind
    nn.DataParallel's gather), which is not the average of per-rank masked means when the ranks' mask counts differ.
    `global_mean_loss(loss, count)` rescales the rank's loss by count_r / sum(count) (one scalar all-reduce, no host sync)
    so that the SUM of the ranks' gradients is the gradient of the global masked mean; allreduce_gradients() then sums
    without dividing.  Without it, gradients are averaged (equal-count shards)."""

    def __init__(self, module: torch.nn.Module, world: int | None = None, late_module: str | None = "feature_extraction",
                 overlap_wgrad: bool = True, always_reduce: bool = False):
        """`always_reduce`: run the collectives (broadcast, count and bucket all-reduces, the side-stream slice) even in a
        process group of ONE rank -- every step then goes through the communicator exactly as at N > 1 (a 1-GPU box can so
        exercise RCCL: tests/test_dist_gpu.py); needs an initialised process group."""
        self.module = module
        if overlap_wgrad and next(module.parameters()).is_cuda:
            # this class owns every reader of the weight gradients (the bucket gather; the optimizer behind
            # allreduce_gradients()), and joins the side stream before both: weight gradients may run there (ops._on_side)
            from . import ops
            ops.enable_wgrad_overlap(True)
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self._collective = self.world > 1 or (always_reduce and dist.is_initialized())
        named = [(k, p) for k, p in module.named_parameters() if p.requires_grad]
        late = getattr(module, late_module, None) if late_module else None
        late_ids = {id(p) for p in late.parameters()} if late is not None else set()
        # bucket order: late (encoder) slice first, then the early-finishing rest -- each slice contiguous
        named.sort(key=lambda kp: 0 if id(kp[1]) in late_ids else 1)
        self.params = [p for _, p in named]
        self.n_late = sum(1 for p in self.params if id(p) in late_ids)
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for i, p in enumerate(self.params):
            if i == self.n_late:
                self.early_off = off
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.n_late == len(self.params):
            self.early_off = off
        for p, v in zip(self.params, self.views):
            p.grad = v
        self._sum_only = False
        self._early_work = None
        self._armed = False
        self._fired = 0
        self._expected = None
        self._side = torch.cuda.Stream(device=ref.device) if ref.is_cuda else None
        self.overlap = self._collective and late is not None and 0 < self.n_late < len(self.params)
        if self.overlap:
            late.register_forward_hook(self._watch_late_outputs)
        if self._collective:
            self.broadcast_parameters()

    def broadcast_parameters(self, src: int = 0):
        """In-place on the tensors themselves under no_grad (the version counters move, unlike a `.data` write), and any
        packed weight layout cached by ops.frozen_weights() is dropped."""
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t, src)
        from . import ops
        ops.invalidate_packed()

    def zero_grad(self):
        """Start of a step: drops the gradients and ARMS the early all-reduce for exactly one backward.  Gradient
        accumulation (several backwards before allreduce_gradients()) or a second backward over a retained graph therefore
        never start a collective over a bucket that later backwards still write: only the first backward after zero_grad()
        may fire it, and once it has fired any further backward of the step is refused."""
        from . import ops
        ops.pace_side_streams()            # the previous step's backward has finished on the device (ops: host run-ahead)
        for p in self.params:
            p.grad = None
        self._armed = True

    # ---- loss scaling --------------------------------------------------------------------------------------------
    def global_mean_loss(self, loss: torch.Tensor, count: torch.Tensor):
        """loss = this rank's masked MEAN, count = its number of masked elements (device scalar, e.g. stereo_loss3's
        metrics[1]).  Returns the loss rescaled so that summing gradients over ranks gives the gradient of the masked mean
        over the GLOBAL batch.  A rank with an empty mask contributes nothing, and a step whose mask is empty on EVERY rank
        has loss 0 with zero (not NaN) gradients.  Contract: every rank calls this in every step it calls
        allreduce_gradients() in (it holds a collective, and switches that step's reduction from average to sum)."""
        if not self._collective:
            return loss
        count = count.detach().to(loss.dtype)
        total = count.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        self._sum_only = True
        scale = count / total.clamp_min(1.0)          # total == 0 => count == 0 => scale 0 (never 0/0)
        return torch.where(count > 0, loss, torch.zeros_like(loss)) * scale

    # ---- overlap ---------------------------------------------------------------------------------------------------
    def _watch_late_outputs(self, _module, _inputs, outputs):
        if not torch.is_grad_enabled():
            return
        outs = outputs if isinstance(outputs, (tuple, list)) else (outputs,)
        for t in outs:
            if torch.is_tensor(t) and t.requires_grad:
                t.register_hook(self._late_output_grad)

    def _late_output_grad(self, grad):
        if self._early_work is not None:
            raise RuntimeError("FlatBucketDDP: a backward pass ran after the early all-reduce of this step had started "
                               "(gradient accumulation / retain_graph): construct with late_module=None or call "
                               "allreduce_gradients() once per backward")
        self._fired += 1
        if self._armed and self._expected is not None and self._fired == self._expected:
            self._armed = False
            self._start_early()
        elif self._expected is not None and self._fired > self._expected:
            self._armed = False                       # a second backward in this step: no early start, reduce at the end
        return grad

    def _gather(self, lo: int, hi: int):
        if self.flat.is_cuda:
            from . import ops
            ops.join_side_streams()                   # weight gradients are computed on a side stream (ops._on_side)
        dst, src, unused = [], [], []
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is None:
                unused.append(v)                      # parameter not reached by this step's graph
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        if unused:
            torch._foreach_zero_(unused)

    def _start_early(self):
        """All gradients of the non-encoder slice exist: gather them and reduce them under the encoder's backward."""
        if any(p.grad is None for p in self.params[self.n_late:]):
            return                                    # a different graph than the one learnt: reduce everything at the end
        self._gather(self.n_late, len(self.params))
        for p, v in zip(self.params[self.n_late:], self.views[self.n_late:]):
            p.grad = v
        tail = self.flat[self.early_off:]
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._early_work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._early_work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)

    def allreduce_gradients(self):
        """Gather the step's gradients into the bucket, sum over ranks (average unless global_mean_loss pre-scaled the loss),
        and leave every .grad pointing at its slice of the bucket for the optimizer."""
        early_done = self._early_work is not None
        self._gather(0, self.n_late if early_done else len(self.params))      # (joins the weight-gradient stream first)
        if self._collective:
            if early_done:
                if self.n_late:
                    dist.all_reduce(self.flat[:self.early_off], op=dist.ReduceOp.SUM)
                self._early_work.wait()               # the current stream waits for the side-stream reduction
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if not self._sum_only:
                self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        if self.overlap and self._expected is None and self._fired > 0:
            self._expected = self._fired              # hooks per backward, learnt from the first step (one backward per step)
        self._fired, self._early_work, self._sum_only, self._armed = 0, None, False, False

    def __call__(self, *a, **k):
        return self.module(*a, **k)


def masked_smooth_l1_x3(preds, gt, maxdisp: int = 192):
    """The reference's training loss (train.py:162,168-174): mask 0<d<maxdisp, smooth-L1 'mean' over the masked pixels
    of each head, weights 0.5/0.7/1.0 -- the fused HIP loss kernel (ops.stereo_loss3: one pass, no `o[mask]` gathers and
    their six device->host syncs per step).  GPU tensors only, like every op of the package (no CPU fallback)."""
    from . import ops
    return ops.stereo_loss3(preds, gt, maxdisp)[0]


def masked_smooth_l1_x3_with_count(preds, gt, maxdisp: int = 192):
    """-> (loss, mask count as a device scalar): what FlatBucketDDP.global_mean_loss needs (no host sync)."""
    from . import ops
    loss, metrics = ops.stereo_loss3(preds, gt, maxdisp)
    return loss, metrics[1]


class GraphedForward:
    """Inference through one captured HIP graph: the eval forward of a model is a fixed sequence of ~400 kernel launches
    (encoder + hot path, all on this library's kernels), which at batch 1 is partly launch-bound; capturing it once and
    replaying it removes the per-launch host cost.  The weight-packing kernels are captured too (ops._cached_pack never
    serves a cached layout during capture), so a replay follows in-place weight updates such as a later
    `load_state_dict` into the same model.  Inputs are copied into the graph's static buffers; the returned tensors are
    the graph's static outputs (valid until the next call).  Shapes are fixed at construction."""

    def __init__(self, model: torch.nn.Module, left: torch.Tensor, right: torch.Tensor, warmup: int = 3,
                 cluster_groupnorm: bool = False):
        """`cluster_groupnorm`: capture the one-pass (cluster) GroupNorm kernels instead of the two-stage ones a capturing
        stream gets by default.  One read pass less per GroupNorm; only for a graph that is never replayed concurrently with
        another graph or with eager GroupNorm work on another stream (two cluster launches in flight together starve each
        other: csrc/gn3d.hip, GnControl)."""
        assert left.is_cuda and right.is_cuda, "GraphedForward captures a HIP graph: GPU tensors only"
        self.model = model.eval()
        self.left, self.right = left.clone(), right.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():      # warm-up outside capture: lazy one-time work (LDS attributes,
            for _ in range(warmup):                          # occupancy queries, constant masks, allocator growth)
                self.model(self.left, self.right)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        from . import ops
        old = ops.gn_cluster_mode(4) if cluster_groupnorm else None      # matters at capture time only: which kernels are recorded
        try:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.out = self.model(self.left, self.right)
        finally:
            if old is not None:
                ops.gn_cluster_mode(old)

    def __call__(self, left: torch.Tensor, right: torch.Tensor):
        self.left.copy_(left)
        self.right.copy_(right)
        self.graph.replay()
        return self.out
