"""How much faster is the encoder (fwd+bwd, 8 images of 576x960) when MIOpen benchmarks its solvers per shape
(cudnn.benchmark=True: miopenFind*) instead of taking its immediate-mode fallback picks?  Prints both timings and writes
the find-db it produced under $MIOPEN_USER_DB_PATH (set by the caller)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import ecm_amd  # noqa: E402


def timeit(fn, iters=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    torch.manual_seed(0)
    enc = ecm_amd.get_model("cmfsm").feature_extraction.cuda().train()
    x = torch.randn(8, 3, 576, 960, device="cuda")

    def step():
        for p in enc.parameters():
            p.grad = None
        f, _, hr = enc(x)
        (f.square().mean() + hr.square().mean()).backward()

    torch.backends.cudnn.benchmark = False
    for _ in range(2):
        step()
    print(f"immediate mode: {timeit(step):.1f} ms per encoder fwd+bwd", flush=True)
    torch.backends.cudnn.benchmark = True
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    print(f"find pass took {time.perf_counter() - t0:.1f} s", flush=True)
    step()
    print(f"after find:     {timeit(step):.1f} ms per encoder fwd+bwd", flush=True)


if __name__ == "__main__":
    main()
