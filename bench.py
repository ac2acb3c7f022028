#!/usr/bin/env python3
"""Headline benchmark: stereo-pairs/s of the cmfsm training step (fwd + bwd + Adam) on SceneFlow-shaped frames
(960x540 padded to 960x576 as the reference's loader does, Flying3d.py:66-72; D=192; batch 4 per GPU) and
ms per cost volume, on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks (the parent never touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W             # ... or runs as one rank of an external launcher

Prints ONE JSON line on rank 0.  Extra objects: `roofline` (dominant kernel, live HIP-event timing), `roofline_costvol`
(the HBM-bound cost-volume build), `roofline_worst` (the kernels furthest below their roofline, live-timed in the same
steps), `configs` (N=1 only: the other BASELINE.json configurations that fit one GPU -- eval forward of cfg 1, the KITTI
frame and the 256x512 training crop of cfg 4, the cfg 5 cost-volume microbench -- each with its own steps / ms so the
driver's clock can bound them), `cpu_baseline` (the oracle port on host cores; N=1 only).
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
    ap.add_argument("--no-extras", action="store_true", help="skip the `configs` and explicit-cost-volume readings (N=1)")
    ap.add_argument("--graph", action="store_true", help="--mode infer only: replay the forward as one captured HIP graph")
    ap.add_argument("--explicit-cost-volume", action="store_true",
                   help="run the reference's explicit op sequence (4-D concat volume + 64->32 Conv3d) instead of the collapsed 2-D form")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--trace-steps", action="store_true", help="diagnostic: synchronise and print the wall time of every timed step")
    ap.add_argument("--launch-table", default="", help="write the second pass's per-(entry point, shape) device times to this JSON file")
    return ap.parse_args()


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(threads: int, H: int, W: int):
    """The oracle (CPU port of the reference op sequence) on a bounded sample of the bench workload, on this box's host
    cores: ONE pair at the bench resolution --
      * fwd + bwd of the training loss, one sample (the headline metric's workload; `value`);
      * eval forward under no_grad: one warm-up + three timed runs on N = all available cores (capped at 16, a 1-GPU box's
        share) and one timed run on 8 threads (comparable with SURVEY section 6's 26.8 s of the reference on 8 cores);
      * the cost-volume stage alone (python loop over disparities, cmfsm.py:667-682)."""
    from oracle import ecm_oracle as O
    import ecm_amd
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads or min(avail, 16)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1234)
    model = ecm_amd.get_model("cmfsm")           # CPU instance used only as a weight container (reference init rule)
    sd_plain = {k: v.detach().clone() for k, v in model.state_dict().items()}
    left, right = torch.randn(1, 3, H, W, generator=g), torch.randn(1, 3, H, W, generator=g)
    gt = torch.rand(1, H, W, generator=g) * 191.0
    fwd = []
    with torch.no_grad():
        O.cmfsm_forward(left, right, sd_plain)                     # warm-up
        for _ in range(3):
            t0 = time.perf_counter()
            O.cmfsm_forward(left, right, sd_plain)
            fwd.append(time.perf_counter() - t0)
        fl, fr = torch.randn(1, 32, H // 4, W // 4, generator=g), torch.randn(1, 32, H // 4, W // 4, generator=g)
        t0 = time.perf_counter()
        O.cost_volume(fl, fr, 48)
        cv_s = time.perf_counter() - t0
        n8 = None
        if cores > 8:
            torch.set_num_threads(8)
            t0 = time.perf_counter()
            O.cmfsm_forward(left, right, sd_plain)
            n8 = time.perf_counter() - t0
            torch.set_num_threads(cores)
    sd = {k: v.detach().clone().requires_grad_() for k, v in sd_plain.items()}
    t0 = time.perf_counter()
    loss = O.train_loss(O.cmfsm_forward(left, right, sd), gt)
    loss.backward()
    dt = time.perf_counter() - t0
    fmean = sum(fwd) / len(fwd)
    return {"value": 1.0 / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "sample": f"1 pair {W}x{H} D=192: fwd+bwd of the training loss once ({dt:.1f} s) = `value`; eval forward 1 warm-up + 3 "
                      f"timed ({fmean:.2f} s mean); torch CPU fp32, {cores} threads",
            "seconds": dt,
            "eval_forward": {"pairs_per_s": 1.0 / fmean, "seconds": fwd, "threads": cores,
                             "threads8_seconds": n8, "threads8_pairs_per_s": (1.0 / n8) if n8 else None},
            "cost_volume_ms": cv_s * 1e3}


def shard_feeder_rate(dev, B, n_frames=32):
    import tempfile
    import numpy as np
    from importlib import import_module
    S = import_module("explicit-context-mapping-for-stereo-matching_amd.shards")
    rs = np.random.RandomState(0)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        frames = []
        base = np.concatenate([rs.randint(0, 256, size=(540, 960, 6)).astype(np.float32),
                               (rs.rand(540, 960, 1) * 191.0).astype(np.float32)], 2)
        for i in range(n_frames):                  # distinct frames from one random base (rolled): cheap to make
            frames.append(np.roll(base, 7 * i, axis=1))
        path = S.write_shard(frames, os.path.join(d, "bench.ecms"), "fp16")
        del frames
        reader = S.ShardReader(path)
        for key, split, label in (("eval_full_frames", "test", "540x960 frames -> [B,3,576,960] (Flying3d.py:66-72), 8 B/px over PCIe"),
                                  ("train_crops", "train", "random 256x512 windows (Flying3d.py:51-56): only the window crosses PCIe")):
            feeder = S.ShardFeeder(reader, B, split=split, device=dev, prefetch=2, shuffle=True)
            for _ in feeder:                       # warm-up epoch: page cache, pinned buffers, allocator
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for _ in range(3):
                for batch in feeder:
                    n += B
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[key] = {"value": n / dt, "unit": "pairs/s", "pairs": n, "seconds": dt, "workload": label}
            feeder.close()
        out["shard"] = {"frames": n_frames, "bytes": os.path.getsize(path), "disparity": "fp16"}
    return out


def _timed(fn, steps, warmup, sync):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher's environment: this process becomes the PARENT of N fresh rank processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would, rendezvous on 127.0.0.1), relays rank 0's
    JSON line and exits with the worst child code.  The parent makes no HIP call at all (nothing here touches torch.cuda) and
    no process is ever replaced by another: every rank is a child started from scratch.  If one rank dies the others would
    wait in a collective for ever, so they are ended (by PID) once any rank has failed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this host driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))

    def relay(stream):
        # rank 0's stdout: the JSON line goes to stdout, anything a library prints there (gloo's connection banner) to stderr
        for line in stream:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
    import signal
    import threading

    def end_ranks(signum, _frame):                               # the parent is told to stop: the ranks must not outlive it
        for p in procs:
            if p.poll() is None:
                p.terminate()
        sys.exit(128 + signum)
    signal.signal(signal.SIGTERM, end_ranks)
    signal.signal(signal.SIGINT, end_ranks)
    pump = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    pump.start()
    worst, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        rcs = [p.poll() for p in procs]
        bad = [rc for rc in rcs if rc not in (None, 0)]
        if bad and failed_at is None:
            failed_at = time.time()
            print(f"[bench] a rank exited with {bad[0]}: ending the other ranks", file=sys.stderr, flush=True)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        if failed_at is not None and time.time() - failed_at > 20:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = p.wait()
        worst = worst or (rc if rc > 0 else (128 - rc if rc < 0 else 0))
    pump.join(timeout=10)
    sys.exit(worst)


def _family(name, a):
    """Kernel family of one C-ABI launch (entry point + its integer arguments): the rows of DESIGN section 6's table."""
    if name.startswith("ecm_conv_wino_fwd"):                # (B,Ci,Co,D,H,W,kd)
        return "winograd_conv_3d" if a[6] == 3 else "winograd_conv_2d"
    if name == "ecm_conv_wino_wgrad":                       # (nbytes?,B,Ci,Co,D,H,W,kd): kd is the last int
        return "winograd_wgrad_3d" if a[-1] == 3 else "winograd_wgrad_2d"
    if name in ("ecm_conv3d_k3_fwd", "ecm_conv3d_k3_wgrad") and a[6] == 1:      # (B,Ci,Co,D,H,W,stride): ECM_WINOGRAD=0 only
        return "direct_conv_3d_stride1"
    if name in ("ecm_conv3d_k3_fwd", "ecm_conv3d_k3_wgrad", "ecm_deconv3d_k3s2_fwd"):
        return "stride2_conv_deconv_wgrad_3d"
    if name.startswith("ecm_gn3d"):
        return "groupnorm"
    if name.startswith("ecm_conv3d_c1"):
        return "classifier_32to1"
    if name.startswith("ecm_weights9") or name.startswith("ecm_context_weights"):
        return "ecm_weights"
    if name.startswith("ecm_conv2d") or name.startswith("ecm_deconv2d") or name == "ecm_zero_insert2d":
        return "conv2d_direct_family"
    if "pack_weight" in name or name.startswith("ecm_costvol_class_weights"):
        return "weight_packing"
    if name.startswith("ecm_costvol"):
        return "cost_volume_assemble"
    if name in ("ecm_softargmin_heads_fwd", "ecm_softargmin_heads_bwd", "ecm_aggregate9_fwd", "ecm_aggregate9_bwd",
                "ecm_stereo_loss_fwd", "ecm_stereo_loss_bwd"):
        return "heads_and_loss"
    if name == "ecm_sum_n":
        return "gradient_sums"
    return "other_hip"


def step_breakdown(timers, n_steps):
    """Per-family device time of ONE step in ms, from the second pass's per-launch HIP events (every C-ABI launch of the
    step; what runs on ATen -- pooling, cat, Adam, autograd's own adds -- is not in it and shows up as the difference to
    ms_per_step)."""
    fam = {}
    launches = 0
    for name, evs in timers.items():
        for s_, e_, a in evs:
            k = _family(name, a)
            fam[k] = fam.get(k, 0.0) + s_.elapsed_time(e_)
            launches += 1
    out = {k: v / n_steps for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}
    out["_sum_hip_launches"] = sum(fam.values()) / n_steps
    out["_launches_per_step"] = launches / n_steps
    out["_source"] = (f"HIP events around every C-ABI launch, second pass of {n_steps} steps with the weight gradients on the "
                      "main stream (no overlap), so the sum exceeds the timed step where the side stream hides part of it")
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)              # never returns
    import ecm_amd
    from importlib import import_module
    ecm_dist = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    mdl = import_module("explicit-context-mapping-for-stereo-matching_amd.models")
    lib, ops = ecm_amd._lib, ecm_amd.ops

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the HIP path has no CPU fallback)"
    # one process per GPU over RCCL ("nccl"); ECM_DIST_BACKEND=gloo lets several ranks rehearse the N>1 path on a
    # box with fewer GPUs than ranks (ranks then share devices round-robin; RCCL itself refuses duplicate devices)
    backend = os.environ.get("ECM_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)
    torch.cuda.set_device(local)
    if backend == "nccl" and int(os.environ.get("WORLD_SIZE", "1")) > ndev:
        raise SystemExit(f"bench.py: {os.environ['WORLD_SIZE']} ranks over RCCL need as many GPUs, this box has {ndev} "
                         "(ECM_DIST_BACKEND=gloo rehearses the N>1 path with ranks sharing devices)")
    rank, world, _ = ecm_dist.init_from_env(backend, device=local)
    assert world == args.gpus or world == 1 and args.gpus == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    devices = min(world, ndev)          # distinct GPUs in use: ranks share devices round-robin only under gloo
    shared_device = world > ndev
    if shared_device:
        # several PROCESSES on one device: the one-pass GroupNorm kernels wait across workgroups, which is only safe to
        # rely on when at most a few launches share the device -- take the two-stage kernels (no inter-workgroup waits);
        # and keep the weight gradients on the main stream: with two processes on one device every extra stream is one more
        # hardware queue to time-slice (measured in round 4: 4.3 s per step with the side stream, 49 ms without)
        ops.gn_cluster_mode(0)
    dev = torch.device("cuda", local)
    # The reference sets cudnn.benchmark=True (train.py:26); on ROCm that means an exhaustive MIOpen search per conv
    # shape (minutes at this size), so what is left of the encoder on MIOpen runs on its immediate-mode picks instead.
    torch.backends.cudnn.benchmark = False

    if args.explicit_cost_volume:
        mdl.EXPLICIT_COST_VOLUME = True
    torch.manual_seed(0)
    model = ecm_amd.get_model("cmfsm").to(dev)
    B, H, W, D = args.batch, args.height, args.width, args.maxdisp

    def make_inputs(b, h, w, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        return (torch.randn(b, 3, h, w, generator=g).to(dev), torch.randn(b, 3, h, w, generator=g).to(dev),
                (torch.rand(b, h, w, generator=g) * 191.0).to(dev))

    left, right, gt = make_inputs(B, H, W, 1234 + rank)
    ddp = opt = None
    if args.mode == "train":
        model.train()
        ddp = ecm_dist.FlatBucketDDP(model, world, overlap_wgrad=not shared_device)
        opt = torch.optim.Adam(ddp.params, lr=1e-3, betas=(0.9, 0.999), fused=True)   # train.py:85-86 (fused: same update rule)

    def train_step(l, r, g):
        ddp.zero_grad()
        loss, count = ecm_dist.masked_smooth_l1_x3_with_count(model(l, r), g, D)      # train.py:162-174
        ddp.global_mean_loss(loss, count).backward()                                  # one masked mean over the GLOBAL batch
        ddp.allreduce_gradients()
        opt.step()
        return loss

    if args.mode == "train":
        def step():
            return train_step(left, right, gt)
    else:
        model.eval()
        if not args.graph:
            ops.frozen_weights().__enter__()          # fixed weights for the whole run: pack each layer's weights once
        if args.graph:
            graphed = ecm_dist.GraphedForward(model, left, right, cluster_groupnorm=True)    # one graph, replayed alone

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
    # time EXACTLY K steps, un-instrumented
    barrier()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    if args.trace_steps:                                   # diagnostic mode (a sync per step; not the headline protocol)
        for i in range(args.steps):
            ts = time.perf_counter()
            step()
            torch.cuda.synchronize()
            note(f"step {i}: {(time.perf_counter() - ts) * 1e3:.1f} ms  (allocator: {torch.cuda.memory_reserved() / 2**30:.1f} GiB reserved)")
    else:
        for _ in range(args.steps):
            step()
    barrier()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    device_allocs_timed = int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0))
    note(f"timed region: {device_allocs_timed} device allocations by the caching allocator, reserved "
         f"{ms0.get('reserved_bytes.all.current', 0) / 2**30:.1f} -> {ms1.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB")
    # second pass, outside the timed region: the dominant kernel and the kernels furthest below their roofline are
    # event-timed per launch on the launch stream (torch's current stream, the one every kernel of the step is launched on)
    TIMED = ("ecm_conv_wino_fwd", "ecm_conv3d_k3_fwd", "ecm_conv3d_k3_wgrad", "ecm_conv3d_c1_fwd", "ecm_conv3d_c1_gn_fwd", "ecm_weights9_fwd",
             "ecm_weights9_bwd", "ecm_deconv3d_k3s2_fwd")
    # The timed steps run the weight gradients on a second stream, under the data-gradient / GroupNorm chain (ops._on_side):
    # a launch then shares the device and its duration says nothing about the kernel.  For this pass they go back onto the
    # main stream, so every timed launch has the device to itself (profiles/: the kernel stats of a run with
    # ECM_WGRAD_OVERLAP=0 are the ones these averages agree with).
    overlap_was = ops.enable_wgrad_overlap(False)
    for name in TIMED:
        lib.enable_timer(name)
    lib.enable_all_timers()                                # ... and every other entry point, for step_breakdown_ms
    n_pass2 = min(args.steps, 5)
    for _ in range(n_pass2):                               # every rank: the training step holds a collective
        step()
    torch.cuda.synchronize()
    timers = lib.disable_timers()
    breakdown = step_breakdown(timers, n_pass2)
    if args.launch_table and rank == 0:
        rows = {}
        for name, evs in timers.items():
            for s_, e_, a in evs:
                r = rows.setdefault((name, tuple(a), tuple(getattr(a, "longs", ()))), [0, 0.0])
                r[0] += 1
                r[1] += s_.elapsed_time(e_)
        table = [{"entry": k[0], "family": _family(k[0], k[1]), "int_args": list(k[1]), "size_args": list(k[2]),
                  "launches_per_step": v[0] / n_pass2, "ms_per_step": v[1] / n_pass2, "avg_launch_ms": v[1] / v[0]}
                 for k, v in rows.items()]
        table.sort(key=lambda r: -r["ms_per_step"])
        with open(args.launch_table, "w") as f:
            json.dump({"note": "per (C-ABI entry point, integer arguments) device time of one training step; HIP events, "
                               "weight gradients on the main stream", "steps": n_pass2, "rows": table}, f, indent=0)
    ops.enable_wgrad_overlap(overlap_was)
    ops.check_async_errors()                               # a GroupNorm cluster time-out during the timed steps is fatal
    last = step()                                          # outside the timed region: the result must be finite
    assert bool(torch.isfinite(last).all()), "non-finite loss / disparity after the timed steps"
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    note(f"timed {args.steps} steps in {dt:.3f} s")

    extras = world == 1 and not args.no_extras and args.mode == "train" and not args.explicit_cost_volume
    sync = torch.cuda.synchronize
    # second reading (1 GPU only): the same step with the reference's explicit op sequence -- 4-D concat volume built by
    # the cost-volume kernel + 64->32 Conv3d on it -- so both forms are on record in the same JSON line
    explicit = None
    configs = {}
    if extras:
        mdl.EXPLICIT_COST_VOLUME = True
        try:
            ms = _timed(step, args.steps, 1, sync) * 1e3
            explicit = {"ms_per_step": ms, "value": B / ms * 1e3, "unit": "pairs/s", "steps": args.steps,
                        "note": "same step with ops.cost_volume (explicit [B,2C,D',h,w] tensor) + 64->32 Conv3d"}
        finally:
            mdl.EXPLICIT_COST_VOLUME = False
        note(f"explicit-cost-volume variant: {explicit['ms_per_step']:.1f} ms/step")

        # ---- the other BASELINE.json configurations that fit one GPU -------------------------------------------------
        # cfg 4 (KITTI fine-tune, 1 pair per GPU): the reference trains on 256x512 crops (KITTI.py:84-97) and evaluates the
        # padded 384x1248 frame; both shapes as a full training step, as SURVEY 8d asks
        for key, (b, h, w, st, wu, label) in {
            "cfg4_kitti_384x1248_train": (1, 384, 1248, 5, 2, "KITTI-2015 1242x375 padded to 1248x384, batch 1/GPU, fwd+bwd+Adam"),
            "cfg4_crop_256x512_train": (4, 256, 512, 5, 2, "256x512 training crop (Flying3d.py:51-56 / KITTI.py:84-97), batch 4/GPU, fwd+bwd+Adam"),
        }.items():
            l2, r2, g2 = make_inputs(b, h, w, 77)
            ms = _timed(lambda: train_step(l2, r2, g2), st, wu, sync) * 1e3
            configs[key] = {"ms_per_step": ms, "value": b / ms * 1e3, "unit": "pairs/s", "steps": st, "warmup": wu, "workload": label}
            note(f"{key}: {ms:.1f} ms/step")
            del l2, r2, g2
        # cfg 1: single 960x540 pair, eval forward as test.py runs it (no_grad, crop [:540,:960], EPE of output3)
        model.eval()
        l1, r1, g1 = make_inputs(1, H, W, 78)

        def eval_step():
            with torch.no_grad():
                return ops.eval_epe(model(l1, r1)[2], g1, min(540, H), min(960, W), D)
        frozen = ops.frozen_weights()     # an evaluation loop over fixed weights: packed weight layouts are built once
        frozen.__enter__()
        ms = _timed(eval_step, 10, 3, sync) * 1e3
        configs["cfg1_eval_forward_b1"] = {"ms_per_step": ms, "value": 1e3 / ms, "unit": "pairs/s", "steps": 10, "warmup": 3,
                                           "workload": "single 960x540 pair (padded to 576), eval forward + crop + EPE (test.py:63-94)"}
        lb, rb, _ = make_inputs(B, H, W, 79)

        def eval_step_b():
            with torch.no_grad():
                return model(lb, rb)[2]
        ms = _timed(eval_step_b, 5, 2, sync) * 1e3
        frozen.__exit__(None, None, None)
        configs["cfg1_eval_forward_b4"] = {"ms_per_step": ms, "value": B / ms * 1e3, "unit": "pairs/s", "steps": 5, "warmup": 2,
                                           "workload": f"eval forward, batch {B}"}
        note(f"eval forward: {configs['cfg1_eval_forward_b1']['ms_per_step']:.2f} ms/pair at B=1, "
             f"{configs['cfg1_eval_forward_b4']['value']:.1f} pairs/s at B={B}")
        model.train()
        del l1, r1, g1, lb, rb
        # the other head families at production size, eval forward as test.py runs it (its default --arch is
        # bilinear_cmf_sub_16, test.py:111): trilinear head (a11) and volume-mapping head (a10), with the head kernel
        # event-timed on the launch stream.  Both heads are bound by fp32 exp / FMA issue over H*W*192 logits per head, not by
        # HBM (they read < 1 MB of LR logits and ~20 MB of weight planes): achieved_GBs is on record, the bound is VALU.
        for arch, entry, label in (("bilinear_cmf_sub_16", "ecm_trilinear_softargmin_fwd", "trilinear head (bilinear_cmf.py:447-471)"),
                                   ("cmfsm_sub_16", "ecm_volume_mapping_fwd", "volume-mapping head (cmfsm_sub_16.py:767-848)")):
            torch.manual_seed(0)
            m2 = ecm_amd.get_model(arch).to(dev).eval()
            lv, rv, _ = make_inputs(1, H, W, 80)

            def eval_v():
                with torch.no_grad():
                    return m2(lv, rv)[2]
            with ops.frozen_weights():
                ms = _timed(eval_v, 10, 3, sync) * 1e3
                lib.enable_timer(entry)
                for _ in range(5):
                    eval_v()
                sync()
                ev = lib.disable_timers()[entry]
            head_ms = sum(s_.elapsed_time(e_) for s_, e_, _ in ev) / len(ev)
            nlog = 3 * H * W * D                                     # logits evaluated by the fused head (3 heads)
            configs[f"eval_forward_b1_{arch}"] = {
                "ms_per_step": ms, "value": 1e3 / ms, "unit": "pairs/s", "steps": 10, "warmup": 3,
                "workload": f"{arch}: single 960x540 pair (padded to 576), eval forward",
                "head_kernel": {"entry": entry, "what": label, "avg_launch_ms": head_ms, "launches_timed": len(ev),
                                "bound": "fp32 VALU (exp + FMA per logit)", "logits_per_launch": nlog,
                                "achieved_Glogits_per_s": nlog / head_ms / 1e6,
                                "not_materialised_bytes": 3 * 4.0 * H * W * D}}
            note(f"{arch}: eval forward {ms:.2f} ms/pair, head kernel {head_ms:.3f} ms")
            del m2, lv, rv
        # cfg 5: cost-volume microbench, 1920x1080 D=256 -> L,R [1,32,270,480], D'=64, 2.16 GB per build
        f5l, f5r = (torch.randn(1, 32, 270, 480, device=dev) for _ in range(2))
        for _ in range(5):
            ops.cost_volume(f5l, f5r, 64)
        sync()
        lib.enable_timer("ecm_costvol_concat_fwd")
        for _ in range(20):
            ops.cost_volume(f5l, f5r, 64)
        sync()
        ev = lib.disable_timers()["ecm_costvol_concat_fwd"]
        ms5 = sum(s.elapsed_time(e) for s, e, _ in ev) / len(ev)
        bytes5 = (2 * 32 * 64 * 270 * 480 + 2 * 32 * 270 * 480) * 4.0
        configs["cfg5_cost_volume_1080p_d256"] = {"ms_per_cost_volume": ms5, "steps": 20, "warmup": 5, "bytes": bytes5,
                                                  "achieved_GBs": bytes5 / ms5 / 1e6, "frac_of_hbm_peak": bytes5 / ms5 / 1e6 / PEAK_HBM_GBS,
                                                  "workload": "L,R [1,32,270,480], D'=64 -> [1,64,64,270,480] (2.16 GB)"}
        note(f"cfg5 cost volume: {ms5:.3f} ms")
        del f5l, f5r
        ops.check_async_errors()
        # n4: the packed-shard feeder (memory map -> pinned buffer -> copy stream -> ecm_frame_prep_packed), alone, on a
        # synthetic shard of 960x540 frames: sustained pairs/s next to the step's pairs/s (it must stay ahead of it)
        try:
            configs["n4_shard_feeder"] = shard_feeder_rate(dev, B)
            note(f"shard feeder: {configs['n4_shard_feeder']['eval_full_frames']['value']:.0f} pairs/s full frames, "
                 f"{configs['n4_shard_feeder']['train_crops']['value']:.0f} pairs/s training crops")
        except OSError as e:                       # no scratch space on this box: report, do not fail the bench line
            configs["n4_shard_feeder"] = {"error": str(e)}

    if rank == 0:
        h, w, Dl = H // 4, W // 4, D // 4
        # "ms per cost-volume build" (BASELINE.json's second metric) is the stand-alone build of the reference's full
        # [B,2C,D',h,w] concat volume (ops.cost_volume = cmfsm.py:667-682), timed here on features of the bench's shape;
        # inside the model the volume is never materialised: its convolution collapses to 2-D ones (ops.costvol_conv3d).
        fl, fr = (torch.randn(B, 32, h, w, device=dev) for _ in range(2))
        for _ in range(3):
            ops.cost_volume(fl, fr, Dl)
        torch.cuda.synchronize()
        lib.enable_timer("ecm_costvol_concat_fwd")
        for _ in range(10):
            ops.cost_volume(fl, fr, Dl)
        torch.cuda.synchronize()
        timers.update(lib.disable_timers())
        del fl, fr

        def ms_of(evs):
            return sum(s.elapsed_time(e) for s, e in evs) / max(1, len(evs))

        # dominant kernel SYMBOL: the stride-1 3x3x3 convolutions with <= 32 output channels (forward convs and, in training,
        # their data gradients) = every ecm_conv_wino_fwd launch with kd == 3 and Co <= 32 (conv_wino_mfma<3,2,1,2>), or with
        # ECM_WINOGRAD=0 every ecm_conv3d_k3_fwd launch with stride 1 (conv3d_k3_mfma<1,1,4,8,4>).
        # achieved = sum(ALGORITHMIC FLOPs = 2*27*Ci*Co*voxels, the direct-convolution count of SURVEY 8d) / sum(duration).
        wino = [(s, e, a) for (s, e, a) in timers.get("ecm_conv_wino_fwd", []) if a[6] == 3 and a[2] <= 32]   # (B,Ci,Co,D,H,W,kd)
        direct = [(s, e, a) for (s, e, a) in timers.get("ecm_conv3d_k3_fwd", []) if a[6] == 1 and a[2] <= 32]  # (..., stride)
        sel = wino if wino else direct
        conv_total_ms = sum(s.elapsed_time(e) for s, e, _ in sel)
        conv_ms = conv_total_ms / max(1, len(sel))
        conv_flop = sum(2.0 * 27 * a[1] * a[2] * a[3] * a[4] * a[5] * a[0] for _, _, a in sel)
        conv_tf = conv_flop / (conv_total_ms * 1e-3) / 1e12 if sel else 0.0
        main_l = [(s, e) for (s, e, a) in sel if a[1] == 32 and a[2] == 32]
        main_ms = ms_of(main_l)
        main_tf = 2.0 * 27 * 32 * 32 * Dl * h * w * B / (main_ms * 1e-3) / 1e12 if main_l else 0.0
        executed = 12.0 / 27.0 if wino else 1.0      # Winograd F(2x2,3x3) x direct depth: 16 multiplies per 4 outputs per kd
        conv_kernel = ("conv_wino_mfma<3,2,1,2> (Winograd F(2x2,3x3) in (h,w), direct along d; all stride-1 3x3x3, Co<=32 launches: fwd + dgrad)"
                       if wino else "conv3d_k3_mfma<1,1,4,8,4> (all stride-1, Co<=32 launches: fwd + dgrad)")
        cv = [(s, e) for (s, e, a) in timers["ecm_costvol_concat_fwd"]]
        cv_ms = ms_of(cv)
        cv_bytes = (2 * 32 * Dl * h * w + 2 * 32 * h * w) * 4.0 * B
        cv_gbs = cv_bytes / (cv_ms * 1e-3) / 1e9 if cv else 0.0
        pairs = B * world * args.steps

        # kernels furthest below their roofline, live-timed in the same steps
        worst = {}
        wg2 = [(s, e, a) for (s, e, a) in timers.get("ecm_conv3d_k3_wgrad", []) if a[6] == 2]     # (B,Ci,Co,D,H,W,stride)
        if wg2:
            tot = sum(s.elapsed_time(e) for s, e, _ in wg2)
            flop = sum(2.0 * 27 * a[1] * a[2] * a[0] * ((a[3] - 1) // 2 + 1) * ((a[4] - 1) // 2 + 1) * ((a[5] - 1) // 2 + 1) for _, _, a in wg2)
            tf = flop / (tot * 1e-3) / 1e12
            worst["conv3d_wgrad_stride2"] = {"kernel": "conv3d_wgrad_mfma (stride-2 layers: hourglass conv1/conv3 and the deconvs' weight gradients)",
                                             "bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                             "frac": tf / PEAK_F32_MFMA_TFLOPS, "launches_timed": len(wg2), "avg_launch_ms": tot / len(wg2)}
        # GroupNorm (17 % of the step, HBM-bound): algorithmic bytes = passes x tensor bytes -- forward x -> y (+ skip), backward
        # x, gy -> gx (+ y where the ReLU mask comes from it, + gskip where it is written), statistics-only x
        gn_bytes = gn_ms = 0.0
        gn_n = 0
        gn_big = [0.0, 0.0, 0]
        for n_, evs in timers.items():
            if not n_.startswith("ecm_gn3d") or n_ in ("ecm_gn3d_cluster_preset",):
                continue
            for s_, e_, a in evs:
                if len(a) < 2 or not a.longs:
                    continue
                elems = float(a[0]) * a[1] * a.longs[-1]               # (B, C, ..., S last of the 64-bit arguments)
                pt = a.ptrs
                if "fwd" in n_:
                    passes = 2 + (1 if len(pt) > 3 and pt[3] else 0)   # (x, gamma, beta, skip, y, ...)
                elif "bwd" in n_:
                    passes = 3 + (1 if len(pt) > 4 and pt[4] else 0) + (1 if len(pt) > 7 and pt[7] else 0)   # (.., y, gy, gx, gskip, ..)
                elif "stats" in n_:
                    passes = 1
                else:
                    passes = 2
                t_ = s_.elapsed_time(e_)
                gn_bytes += passes * elems * 4.0
                gn_ms += t_
                gn_n += 1
                if elems == float(B) * 32 * Dl * h * w:                # the 849 MB maps of the 3-D stack at batch 4
                    gn_big[0] += passes * elems * 4.0
                    gn_big[1] += t_
                    gn_big[2] += 1
        if gn_n:
            gbs = gn_bytes / (gn_ms * 1e-3) / 1e9
            worst["groupnorm"] = {"kernel": "gn_fused_fwd / gn_fused_bwd (cluster kernels, one read + one write pass per operand) + gn_stats",
                                  "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                  "launches_timed": gn_n, "ms_per_step": gn_ms / n_pass2,
                                  "of_which_quarter_res_volumes": ({"achieved": gn_big[0] / (gn_big[1] * 1e-3) / 1e9,
                                                                    "frac": gn_big[0] / (gn_big[1] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                                    "launches_timed": gn_big[2]} if gn_big[2] else None),
                                  "yardstick": "ATen add (2 reads + 1 write, 849 MB operands) streams at 6.0 TB/s = 0.75 on this part (profiles/r04_stream_ceilings.txt)"}
        w2 = [(s, e, a) for n_ in ("ecm_conv_wino_fwd", "ecm_conv_wino_fwd_add") for (s, e, a) in timers.get(n_, []) if a[6] == 1]
        if w2:
            # 2-D Winograd F(2x2,3x3) (encoder + class convolutions, forward and data gradient; D = independent planes):
            # direct count 2*9*Ci*Co per output pixel, of which the kernel executes 16/36 on the matrix cores
            tot = sum(s.elapsed_time(e) for s, e, _ in w2)
            flop = sum(2.0 * 9 * a[1] * a[2] * a[0] * a[3] * a[4] * a[5] for _, _, a in w2)
            tf = flop / (tot * 1e-3) / 1e12
            big = [(s, e, a) for (s, e, a) in w2 if a[1] == 32 and a[2] == 32 and a[4] * a[5] == H * W]
            bt = sum(s.elapsed_time(e) for s, e, _ in big)
            bf = sum(2.0 * 9 * 32 * 32 * a[0] * a[3] * a[4] * a[5] for _, _, a in big)
            worst["conv_wino_2d"] = {"kernel": "conv_wino_mfma<1,1,2,4> / <1,2,..> (2-D Winograd: encoder and class convolutions, fwd + dgrad)",
                                     "bound": "mfma", "achieved": tf * 16.0 / 36.0, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                     "frac": tf * 16.0 / 36.0 / PEAK_F32_MFMA_TFLOPS, "executed_over_algorithmic": 16.0 / 36.0,
                                     "algorithmic_equiv_TFLOPs": tf, "launches_timed": len(w2), "avg_launch_ms": tot / len(w2),
                                     "ms_per_step": tot / n_pass2,
                                     "of_which_32to32_fullres": ({"frac": bf / (bt * 1e-3) / 1e12 * 16.0 / 36.0 / PEAK_F32_MFMA_TFLOPS,
                                                                  "avg_launch_ms": bt / len(big), "launches_timed": len(big)} if big else None)}
        c1 = [(s, e, a) for n_ in ("ecm_conv3d_c1_fwd", "ecm_conv3d_c1_gn_fwd") for (s, e, a) in timers.get(n_, [])]   # (B,Ci,D,H,W)
        if c1:
            tot = sum(s.elapsed_time(e) for s, e, _ in c1)
            byt = sum((a[1] + 1) * 4.0 * a[0] * a[2] * a[3] * a[4] for _, _, a in c1)
            gbs = byt / (tot * 1e-3) / 1e9
            worst["conv3d_c1_fwd"] = {"kernel": "conv3d_c1_fwd_v (classifier 32->1; since round 4 with GroupNorm + ReLU applied on load)", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                                      "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "launches_timed": len(c1), "avg_launch_ms": tot / len(c1)}
        ew = [(s, e, a) for (s, e, a) in timers.get("ecm_weights9_fwd", [])]                      # (B,h,w,s)
        if ew:
            tot = sum(s.elapsed_time(e) for s, e, _ in ew)
            # bound = max(HBM time, fp32 time): 32*(H*W + h*w)*4 + 9*H*W*4 bytes; factorised MLP flops per SURVEY 8d
            bound_ms = 0.0
            for _, _, a in ew:
                bb, hh, ww, ss = a[0], a[1], a[2], a[3]
                HW = hh * ss * ww * ss
                byt = bb * (32 * (HW + hh * ww) + 9 * HW) * 4.0
                flop = bb * (2.0 * 32 * 32 * (HW + hh * ww) + 9 * HW * 2.0 * (32 * 16 + 16 * 8 + 8 + 2 * 32))
                bound_ms += max(byt / (PEAK_HBM_GBS * 1e9), flop / (PEAK_F32_MFMA_TFLOPS * 1e12)) * 1e3
            worst["ecm_weights9_fwd"] = {"kernel": "ecm_lr_proj + ecm_weights_fwd_kernel<0>", "bound": "max(hbm, fp32)",
                                         "bound_ms_per_launch": bound_ms / len(ew), "avg_launch_ms": tot / len(ew),
                                         "frac": bound_ms / tot, "launches_timed": len(ew)}

        eb = [(s, e, a) for (s, e, a) in timers.get("ecm_weights9_bwd", [])]                     # (B,h,w,s)
        if eb:
            tot = sum(s.elapsed_time(e) for s, e, _ in eb)
            # fp32 work per full-resolution pixel: 9 neighbours x (forward recompute 1,424 + backward chain 1,296 + weight-gradient
            # outer products 1,424 flop) + per-pixel first-layer products 3 x 2,048 = 43.4 kflop; HBM: hr read twice, ghr written,
            # saved planes + their gradient read
            bound_ms = 0.0
            for _, _, a in eb:
                HW = a[0] * a[1] * a[3] * a[2] * a[3]
                flop = HW * (9.0 * (1424 + 1296 + 1424) + 3 * 2048)
                byt = HW * (3 * 32 + 18) * 4.0
                bound_ms += max(byt / (PEAK_HBM_GBS * 1e9), flop / (PEAK_F32_MFMA_TFLOPS * 1e12)) * 1e3
            worst["ecm_weights9_bwd"] = {"kernel": "bwd_lr_proj + ecm_weights_bwd_kernel_p<0> + ecm_weights_bwd_cells<0> + ecm_weights_bwd_reduce",
                                         "bound": "max(hbm, fp32)", "bound_ms_per_launch": bound_ms / len(eb),
                                         "avg_launch_ms": tot / len(eb), "frac": bound_ms / tot, "launches_timed": len(eb)}
        dc = [(s, e, a) for (s, e, a) in timers.get("ecm_deconv3d_k3s2_fwd", [])]                # (B,Ci,Co,D,H,W,Do,Ho,Wo)
        if dc:
            tot = sum(s.elapsed_time(e) for s, e, _ in dc)
            flop = sum(2.0 * 27 * a[1] * a[2] * a[0] * a[3] * a[4] * a[5] for _, _, a in dc)      # per INPUT voxel: 27 taps x Ci x Co
            tf = flop / (tot * 1e-3) / 1e12
            worst["deconv3d_k3s2"] = {"kernel": "deconv3d_k3s2_mfma (hourglass conv5 / conv6 and the stride-2 convolutions' data gradients)",
                                      "bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf / PEAK_F32_MFMA_TFLOPS, "launches_timed": len(dc), "avg_launch_ms": tot / len(dc)}

        shape_name = {(576, 960): "SceneFlow 960x540 (padded to 576)",
                      (384, 1248): "KITTI-2015 1242x375 (padded to 1248x384)"}.get((H, W), f"synthetic {W}x{H}")
        # roofline.traffic: HBM bytes per launch from rocprofv3 PMC passes of THIS round (profiles/r02_pmc_traffic.json, written
        # by tools/pmc_traffic.py with the calibration of tools/micro/fetch_calib.hip applied); null until that file exists --
        # round 1's figure used an uncalibrated FETCH_SIZE for 4-byte-per-lane buffer loads and read below the algorithmic bytes.
        traffic_cv = traffic_conv = traffic_source = None
        traffic_note = "no calibrated PMC pass on file (profiles/r0N_pmc_traffic.json)"
        try:
            import glob
            pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0?_pmc_traffic.json")))[-1]     # the latest round's
            with open(pmc_file) as f:
                pm = json.load(f)
            if (H, W, D, B) == (576, 960, 192, 4):
                traffic_conv = pm["conv_wino_mfma_32to32_B4" if ops.WINOGRAD and "conv_wino_mfma_32to32_B4" in pm
                                  else "conv3d_k3_mfma_32to32_B4"]["hbm_bytes_per_launch"]
                traffic_cv = pm["costvol_fwd_v4_B4"]["hbm_bytes_per_launch"]
                traffic_note = pm.get("note", "rocprofv3 PMC, calibrated") + f" [{os.path.basename(pmc_file)}]"
                traffic_source = "profiles/" + os.path.basename(pmc_file) + " (a committed rocprofv3 --pmc run, NOT measured in this run)"
        except (OSError, KeyError, ValueError, IndexError):
            pass
        out = {
            "metric": "stereo-pairs/sec (cmfsm train step fwd+bwd+Adam)" if args.mode == "train"
                      else "stereo-pairs/sec (cmfsm eval forward)",
            "value": pairs / dt, "unit": "pairs/s", "n_gpus": devices, "ranks": world, "devices": devices,
            "backend": backend if world > 1 else "none (single process)", "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # the timed region must not go to the driver for memory (ops: host run-ahead bound); both figures are of rank 0
            "device_allocations_in_timed_region": device_allocs_timed,
            "allocator_reserved_gib": round(ms1.get("reserved_bytes.all.current", 0) / 2**30, 1),
            "config": {"workload": f"{shape_name} D={D} batch={B}/GPU "
                                   f"{'fwd+bwd+Adam (train.py path)' if args.mode == 'train' else 'eval forward (test.py path)'}",
                       "arch": "cmfsm", "global_batch": B * world, "parallelism": f"dp{world}",
                       "cost_volume": "explicit 4-D tensor" if args.explicit_cost_volume else "collapsed into class-indexed 2-D convolutions",
                       "launch": "hip graph replay" if (args.mode == "infer" and args.graph) else "eager"},
            "ms_per_cost_volume": cv_ms / B,
            # frac is on EXECUTED matrix-core work: the Winograd kernel issues 12/27 of the direct convolution's multiplies
            # (16 per 2x2 outputs per (kd,ci,co) instead of 36), so executed FLOPs = algorithmic x 12/27 -- the number
            # the PMC count of v_mfma_f32_32x32x2_f32 reproduces (profiles/r0*_pmc_mfma_util.json); the direct-count rate
            # of SURVEY 8d stays on record as algorithmic_equiv
            "roofline": {"kernel": conv_kernel, "bound": "mfma",
                         "achieved": conv_tf * executed, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": conv_tf * executed / PEAK_F32_MFMA_TFLOPS,
                         "algorithmic_equiv": {"achieved": conv_tf, "unit": "TFLOP/s", "x_peak": conv_tf / PEAK_F32_MFMA_TFLOPS,
                                               "note": "direct-convolution FLOP count 2*27*Ci*Co*voxels / time (SURVEY 8d); not a "
                                                       "roofline fraction for a Winograd kernel"},
                         "executed_over_algorithmic": executed,
                         "traffic": traffic_conv, "traffic_source": traffic_source, "traffic_note": traffic_note,
                         "launches_timed": len(sel), "avg_launch_ms": conv_ms,
                         "timing": "HIP events per launch on the launch stream, second pass after the timed steps with the weight "
                                   "gradients back on the main stream (in the timed steps they overlap the rest of backward on a "
                                   "second stream: %s)" % ("on" if overlap_was else "off"),
                         "of_which_32to32": {"achieved": main_tf * executed, "frac": main_tf * executed / PEAK_F32_MFMA_TFLOPS,
                                             "algorithmic_equiv": main_tf,
                                             "avg_launch_ms": main_ms, "launches_timed": len(main_l)}},
            "roofline_costvol": {"kernel": "costvol_fwd_v4", "bound": "hbm", "achieved": cv_gbs, "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": cv_gbs / PEAK_HBM_GBS, "traffic": traffic_cv,
                                 "traffic_source": traffic_source,
                                 "launches_timed": len(cv), "avg_launch_ms": cv_ms},
            "roofline_worst": worst,
            "step_breakdown_ms": breakdown,
        }
        if explicit is not None:
            # the reading with north_star's cost-volume kernel ON the step's path, as top-level scalars
            out["explicit_ms_per_step"] = explicit["ms_per_step"]
            out["explicit_pairs_per_s"] = explicit["value"]
            out["explicit_cost_volume"] = explicit
        if configs:
            out["configs"] = configs
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline (oracle port: 1 pair at the bench resolution; 4 eval forwards + 1 fwd+bwd) ...")
            cb = cpu_baseline(args.cpu_threads, H, W)
            note(f"cpu baseline done: fwd+bwd {cb['seconds']:.1f} s, eval forward {1.0 / cb['eval_forward']['pairs_per_s']:.2f} s")
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
