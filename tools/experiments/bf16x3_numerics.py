#!/usr/bin/env python3
"""Numerics of the split-bf16 ("bf16x3") product that VERDICT r3 item 6 proposes for the matrix kernels, emulated on the CPU:
x = x1 + x2 + x3 (three bf16 terms, round-to-nearest each), w likewise; the six products x1w1, x1w2, x2w1, x1w3, x3w1, x2w2 are
exact in fp32 (8 x 8 significant bits) and are accumulated in fp32 as the matrix cores do.  Compared with the plain fp32 dot
product, against fp64, at the reduction length of a 3x3x3 convolution over 32 channels (K = 864) and of the 64-channel one.

usage: python tools/experiments/bf16x3_numerics.py        (prints a table; tests/test_oracle_c.py-style asserts live in
tests/test_bf16x3_numerics.py)"""
import torch


def split3(t):
    a = t.bfloat16().float()
    r = t - a
    b = r.bfloat16().float()
    r = r - b
    c = r.bfloat16().float()
    return a, b, c


def bf16x3_matmul(x, w, terms=6):
    x1, x2, x3 = split3(x)
    w1, w2, w3 = split3(w)
    prods = [(x1, w1), (x1, w2), (x2, w1), (x1, w3), (x3, w1), (x2, w2)][:terms]
    acc = torch.zeros(x.shape[0], w.shape[1], dtype=torch.float32)
    for a, b in reversed(prods):                 # small terms first, as a kernel would order them for accuracy
        acc = acc + a @ b                        # each product exact in fp32; fp32 accumulation
    return acc


def errors(K, n=256, seed=0, terms=6):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, K, generator=g)
    w = torch.randn(K, n, generator=g) * (2.0 / K) ** 0.5
    truth = x.double() @ w.double()
    scale = float(truth.abs().max())
    e32 = float(((x @ w).double() - truth).abs().max()) / scale
    e3 = float((bf16x3_matmul(x, w, terms).double() - truth).abs().max()) / scale
    rep = float(((sum(split3(x)).double() - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max())
    return e32, e3, rep


if __name__ == "__main__":
    torch.set_num_threads(4)
    print("K      fp32 dot     bf16x3 (6 products)   bf16x3 (3 products: x1w1+x1w2+x2w1)   split representation error")
    for K in (864, 1728, 288):
        e32, e6, rep = errors(K)
        _, e3, _ = errors(K, terms=3)
        print(f"{K:5d}  {e32:.3e}    {e6:.3e}             {e3:.3e}                           {rep:.2e}")
