#!/usr/bin/env python3
"""Headline benchmark: stereo-pairs/s of the cmfsm training step (fwd + bwd + Adam) on SceneFlow-shaped frames
(960x540 padded to 960x576 as the reference's loader does, Flying3d.py:66-72; D=192; batch 4 per GPU) and
ms per cost volume, on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Extra objects: `roofline` (dominant kernel, live HIP-event timing),
`roofline_costvol` (the HBM-bound cost-volume build), `cpu_baseline` (the oracle port on host cores; N=1 only).
Synthetic data (N(0,1) images, U(0,191) ground truth), random-init weights (reference init rule).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0             # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="stereo pairs per GPU per step")
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--maxdisp", type=int, default=192)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="--mode infer only: replay the forward as one captured HIP graph")
    ap.add_argument("--explicit-cost-volume", action="store_true",
                   help="run the reference's explicit op sequence (4-D concat volume + 64->32 Conv3d) instead of the collapsed 2-D form")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(threads: int, H: int, W: int):
    """Oracle (CPU port of the reference op sequence) on a bounded sample of the same workload: ONE pair at the
    bench resolution, forward + backward of the training loss (about 20 s on 16 host threads)."""
    from oracle import ecm_oracle as O
    import ecm_amd
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads or min(avail, 16)            # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1234)
    model = ecm_amd.get_model("cmfsm")           # CPU instance used only as a weight container (reference init rule)
    sd = {k: v.detach().clone().requires_grad_() for k, v in model.state_dict().items()}
    left, right = torch.randn(1, 3, H, W, generator=g), torch.randn(1, 3, H, W, generator=g)
    gt = torch.rand(1, H, W, generator=g) * 191.0
    t0 = time.perf_counter()
    preds = O.cmfsm_forward(left, right, sd)
    t1 = time.perf_counter()
    loss = O.train_loss(preds, gt)
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"1 pair {W}x{H} D=192, fwd+bwd of the training loss, {dt:.1f} s (fwd {t1 - t0:.1f} s); "
                      f"torch CPU fp32, {cores} threads",
            "seconds": dt, "fwd_only_pairs_per_s": 1.0 / (t1 - t0)}


def main():
    args = parse()
    import ecm_amd
    from importlib import import_module
    ecm_dist = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    lib = ecm_amd._lib

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the HIP path has no CPU fallback)"
    # one process per GPU over RCCL ("nccl"); ECM_DIST_BACKEND=gloo lets several ranks rehearse the N>1 path on a
    # box with fewer GPUs than ranks (ranks then share devices round-robin; RCCL itself refuses duplicate devices)
    backend = os.environ.get("ECM_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)
    torch.cuda.set_device(local)
    rank, world, _ = ecm_dist.init_from_env(backend, device=local)
    assert world == args.gpus or world == 1 and args.gpus == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    dev = torch.device("cuda", local)
    # The reference sets cudnn.benchmark=True (train.py:26); on ROCm that means an exhaustive MIOpen search per conv
    # shape (minutes at this size), so the encoder runs on MIOpen's immediate-mode picks instead.
    torch.backends.cudnn.benchmark = False

    if args.explicit_cost_volume:
        import_module("explicit-context-mapping-for-stereo-matching_amd.models").EXPLICIT_COST_VOLUME = True
    torch.manual_seed(0)
    model = ecm_amd.get_model("cmfsm").to(dev)
    B, H, W, D = args.batch, args.height, args.width, args.maxdisp
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    left = torch.randn(B, 3, H, W, generator=g).to(dev)
    right = torch.randn(B, 3, H, W, generator=g).to(dev)
    gt = (torch.rand(B, H, W, generator=g) * 191.0).to(dev)

    if args.mode == "train":
        model.train()
        ddp = ecm_dist.FlatBucketDDP(model, world)
        opt = torch.optim.Adam(ddp.params, lr=1e-3, betas=(0.9, 0.999), fused=True)   # train.py:85-86 (fused: same update rule)

        def step():
            ddp.zero_grad()
            loss = ecm_dist.masked_smooth_l1_x3(model(left, right), gt, D)
            loss.backward()
            ddp.allreduce_gradients()
            opt.step()
            return loss
    else:
        model.eval()
        if args.graph:
            graphed = ecm_dist.GraphedForward(model, left, right)

            def step():
                return graphed(left, right)[2]
        else:
            def step():
                with torch.no_grad():
                    return model(left, right)[2]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier(device_ids=[local]) if backend == "nccl" else torch.distributed.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        note(f"warmup step {i + 1}/{args.warmup} done")
    # time EXACTLY K steps; the dominant kernels are also event-timed per launch on the launch stream
    lib.enable_timer("ecm_conv3d_k3_fwd")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timers = lib.disable_timers()
    last = step()                                          # outside the timed region: the result must be finite
    assert bool(torch.isfinite(last).all()), "non-finite loss / disparity after the timed steps"
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    note(f"timed {args.steps} steps in {dt:.3f} s")
    # second reading (1 GPU only): the same step with the reference's explicit op sequence -- 4-D concat volume built by
    # the cost-volume kernel + 64->32 Conv3d on it -- so both forms are on record in the same JSON line
    explicit = None
    if world == 1 and not args.explicit_cost_volume and not args.graph:
        mdl = import_module("explicit-context-mapping-for-stereo-matching_amd.models")
        mdl.EXPLICIT_COST_VOLUME = True
        try:
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dte = time.perf_counter() - t1
            explicit = {"ms_per_step": 1e3 * dte / args.steps, "value": B * args.steps / dte, "unit": "pairs/s",
                        "note": "same step with ops.cost_volume (explicit [B,2C,D',h,w] tensor) + 64->32 Conv3d"}
        finally:
            mdl.EXPLICIT_COST_VOLUME = False
        note(f"explicit-cost-volume variant: {explicit['ms_per_step']:.1f} ms/step")
    if rank == 0:
        h, w, Dl = H // 4, W // 4, D // 4
        # "ms per cost-volume build" (BASELINE.json's second metric) is the stand-alone build of the reference's full
        # [B,2C,D',h,w] concat volume (ops.cost_volume = cmfsm.py:667-682), timed here on features of the bench's shape;
        # inside the model the volume is never materialised: its convolution collapses to 2-D ones (ops.costvol_conv3d).
        fl, fr = (torch.randn(B, 32, h, w, device=dev) for _ in range(2))
        for _ in range(3):
            ecm_amd.ops.cost_volume(fl, fr, Dl)
        torch.cuda.synchronize()
        lib.enable_timer("ecm_costvol_concat_fwd")
        for _ in range(10):
            ecm_amd.ops.cost_volume(fl, fr, Dl)
        torch.cuda.synchronize()
        timers.update(lib.disable_timers())
        del fl, fr
        # dominant kernel SYMBOL: conv3d_k3_mfma<1,1,4,8,4> = every ecm_conv3d_k3_fwd launch with stride 1 and Co <= 32
        # (forward convs, and in training the stride-1 data gradients that run on the same kernel).
        # int args of the call: (B, Ci, Co, D, H, W, stride).  achieved = sum(algorithmic FLOPs) / sum(duration).
        sel = [(s, e, a) for (s, e, a) in timers["ecm_conv3d_k3_fwd"] if a[6] == 1 and a[2] <= 32]
        conv_total_ms = sum(s.elapsed_time(e) for s, e, _ in sel)
        conv_ms = conv_total_ms / max(1, len(sel))
        conv_flop = sum(2.0 * 27 * a[1] * a[2] * a[3] * a[4] * a[5] * a[0] for _, _, a in sel)
        conv_tf = conv_flop / (conv_total_ms * 1e-3) / 1e12 if sel else 0.0
        main = [(s, e) for (s, e, a) in sel if a[1] == 32 and a[2] == 32]
        main_ms = sum(s.elapsed_time(e) for s, e in main) / max(1, len(main))
        main_tf = 2.0 * 27 * 32 * 32 * Dl * h * w * B / (main_ms * 1e-3) / 1e12 if main else 0.0
        cv = [(s, e) for (s, e, a) in timers["ecm_costvol_concat_fwd"]]
        cv_ms = sum(s.elapsed_time(e) for s, e in cv) / max(1, len(cv))
        cv_bytes = (2 * 32 * Dl * h * w + 2 * 32 * h * w) * 4.0 * B
        cv_gbs = cv_bytes / (cv_ms * 1e-3) / 1e9 if cv else 0.0
        pairs = B * world * args.steps
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs,
        # gfx950 corrections applied as profiles/r01_pmc_traffic.json says), measured at B=1 and scaled by B.
        traffic_cv = traffic_conv = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pm = json.load(f)
            if (H, W, D) == (576, 960, 192):
                traffic_cv = pm["costvol_fwd_v4"]["hbm_bytes_per_launch_B1"] * B
                cv_pm = pm["conv3d_k3_mfma<1,1,4,8,4> 32->32"]
                traffic_conv = cv_pm["hbm_bytes_per_launch_B4_xcd_aware"] if B == 4 and "hbm_bytes_per_launch_B4_xcd_aware" in cv_pm \
                    else cv_pm["hbm_bytes_per_launch_B1"] * B
        except (OSError, KeyError, ValueError):
            pass
        shape_name = {(576, 960): "SceneFlow 960x540 (padded to 576)",
                      (384, 1248): "KITTI-2015 1242x375 (padded to 1248x384)"}.get((H, W), f"synthetic {W}x{H}")
        out = {
            "metric": "stereo-pairs/sec (cmfsm train step fwd+bwd+Adam)" if args.mode == "train"
                      else "stereo-pairs/sec (cmfsm eval forward)",
            "value": pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{shape_name} D={D} batch={B}/GPU "
                                   f"{'fwd+bwd+Adam (train.py path)' if args.mode == 'train' else 'eval forward (test.py path)'}",
                       "arch": "cmfsm", "global_batch": B * world, "parallelism": f"dp{world}",
                       "cost_volume": "explicit 4-D tensor" if args.explicit_cost_volume else "collapsed into 2-D convolutions",
                       "launch": "hip graph replay" if (args.mode == "infer" and args.graph) else "eager"},
            "ms_per_cost_volume": cv_ms / B,
            "roofline": {"kernel": "conv3d_k3_mfma<1,1,4,8,4> (all stride-1, Co<=32 launches: fwd + dgrad)", "bound": "mfma",
                         "achieved": conv_tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": conv_tf / PEAK_F32_MFMA_TFLOPS, "traffic": traffic_conv,
                         "traffic_note": "HBM bytes of one 32->32 launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json)",
                         "launches_timed": len(sel),
                         "avg_launch_ms": conv_ms,
                         "of_which_32to32": {"achieved": main_tf, "frac": main_tf / PEAK_F32_MFMA_TFLOPS,
                                             "avg_launch_ms": main_ms, "launches_timed": len(main)}},
            "roofline_costvol": {"kernel": "costvol_fwd_v4", "bound": "hbm", "achieved": cv_gbs, "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": cv_gbs / PEAK_HBM_GBS, "traffic": traffic_cv,
                                 "launches_timed": len(cv), "avg_launch_ms": cv_ms},
        }
        if explicit is not None:
            out["explicit_cost_volume"] = explicit
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline (oracle port, 1 pair fwd+bwd at the bench resolution) ...")
            cb = cpu_baseline(args.cpu_threads, H, W)
            note(f"cpu baseline done: {cb['seconds']:.1f} s")
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
