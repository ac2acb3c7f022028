"""Deterministic per-key synthetic weights (TEST INFRASTRUCTURE ONLY).

No trained checkpoint of the reference exists offline, so parity is checked on
synthetic weights that both the build container (where the reference can be
imported to make golden vectors) and the GPU box (where it cannot) can
regenerate bit-identically: each tensor is drawn from a `torch.Generator`
seeded with crc32(key), on CPU, in fp32.
"""
from __future__ import annotations

import math
import zlib

import torch


def _gen(key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode()))
    return g


def tensor_for(key: str, shape) -> torch.Tensor:
    """Conv kernels ~ N(0, 2/fan) (gain chosen so activations stay O(1) through the stack);
    GroupNorm weight ~ 1 + 0.2 N, bias ~ 0.2 N (so affine terms are exercised)."""
    shape = tuple(shape)
    g = _gen(key)
    if len(shape) >= 3:                         # conv / deconv kernels
        taps = 1
        for k in shape[2:]:
            taps *= k
        fan = shape[1] * taps
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan)
    if key.endswith(".weight"):                 # GroupNorm gamma
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    return 0.2 * torch.randn(shape, generator=g)


def make_state_dict(shapes: dict) -> dict:
    return {k: tensor_for(k, s) for k, s in shapes.items()}


def seeded(name: str, *shape, scale: float = 1.0) -> torch.Tensor:
    """A named deterministic input tensor."""
    return torch.randn(*shape, generator=_gen("input:" + name)) * scale
