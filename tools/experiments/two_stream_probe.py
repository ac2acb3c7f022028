#!/usr/bin/env python3
"""Probe: does the device get through two half-batches faster when they run as two concurrent streams (two Python threads,
each a full training step at batch 2 on its own model replica) than through one batch-4 step?  (Kernels of one micro-batch's
HBM-bound passes would fill the other's matrix-bound phases.)"""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ecm_amd
from importlib import import_module
ops = ecm_amd.ops
D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
ops.enable_wgrad_overlap(False)
H, W = 576, 960


def make(B):
    m = ecm_amd.get_model("cmfsm").cuda().train()
    return m, torch.randn(B, 3, H, W, device="cuda"), torch.randn(B, 3, H, W, device="cuda"), torch.rand(B, H, W, device="cuda") * 191


def step(m, l, r, g):
    for p in m.parameters():
        p.grad = None
    D.masked_smooth_l1_x3(m(l, r), g).backward()


def timed(fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


m4 = make(4)
print(f"one stream, batch 4: {timed(lambda: step(*m4)):.1f} ms (fwd+bwd, no optimizer)", flush=True)
del m4
torch.cuda.empty_cache()
ma, mb = make(2), make(2)
print(f"one stream, batch 2 twice in sequence: {timed(lambda: (step(*ma), step(*mb))):.1f} ms", flush=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def worker(model, stream, n, bar):
    with torch.cuda.stream(stream):
        bar.wait()
        for _ in range(n):
            step(*model)


def both(n=4):
    bar = threading.Barrier(2)
    ta = threading.Thread(target=worker, args=(ma, sa, n, bar)); tb = threading.Thread(target=worker, args=(mb, sb, n, bar))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ta.start(); tb.start(); ta.join(); tb.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


both(1)
print(f"two streams / two threads, batch 2 each: {both():.1f} ms per pair of steps", flush=True)
torch.cuda.synchronize()
print("async status", ops._lib.query("ecm_async_status", 1))
