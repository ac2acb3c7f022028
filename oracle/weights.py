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


def eval_harness_inputs():
    """Deterministic inputs of the eval-harness fixture g9 (tests/golden/make_golden_eval.py and the tests that check it
    build the SAME tensors from this function; only the reference's outputs are stored in the fixture)."""
    import numpy as np
    g = torch.Generator().manual_seed(9)
    B, H, W = 2, 576, 960
    disp = torch.rand(B, H, W, generator=g) * 260.0 - 20.0          # some values < 0 and >= 192: every mask edge is hit
    disp[0, :50, :40] = 0.0                                          # d == 0 separates mask / mask_true
    disp[1, 100:140, :] = 191.99
    disp[:, :, :8] = disp[:, :, :8].abs() + 3.0                      # x - d < 0 near the left border
    out3 = (disp + torch.randn(B, H, W, generator=g) * 2.0).unsqueeze(1)
    h, w = 375, 1242
    o3k = torch.rand(1, 1, 384, 1248, generator=g) * 191.0
    edge = torch.tensor([0.0, 0.00390625, 0.0039, 1.0, 255.99609375, 255.998, 256.0, 300.0, -0.001, -0.5, -1.0, -2.75,
                         1e6, -1e6, 8388607.5, 3e9, -3e9, float("nan"), float("inf"), float("-inf"), 191.999, 0.999,
                         2.0 ** -10, 100.5])
    o3k[0, 0, 9:12, 6:30] = edge.view(1, 24).expand(3, 24)           # inside the un-padded 375 x 1242 window
    frame = np.concatenate([np.random.RandomState(9).randint(0, 256, size=(h, w, 6)).astype(np.float32),
                            (np.random.RandomState(10).rand(h, w, 1) * 200.0).astype(np.float32)], 2)
    return dict(sf_disparity=disp, sf_output3=out3, kitti_output3=o3k, kitti_hw=(h, w), kitti_frame=frame)


def fullframe_frame(kind: str):
    """Raw [H,W,7] float32 frame (left RGB, right RGB as uint8-valued floats, disparity) of the full-frame fixtures
    (tests/golden/make_golden_fullframe.py and the tests that check them build the SAME frame from this function):
    kind "sceneflow" = 540x960 (padded to 576x960 by the loader, Flying3d.py:66-72), "kitti" = 375x1242 (padded to
    384x1248, KITTI.py:98-108).  8x8-blocky random colours + per-pixel noise, the right image = the left one shifted by a
    row-dependent disparity (so the two maps are related), disparity channel = that shift."""
    import numpy as np
    H, W = {"sceneflow": (540, 960), "kitti": (375, 1242)}[kind]
    rs = np.random.RandomState(zlib.crc32(("fullframe:" + kind).encode()) & 0x7FFFFFFF)
    hb, wb = (H + 7) // 8, (W + 7) // 8
    base = np.repeat(np.repeat(rs.randint(32, 224, size=(hb, wb, 3)), 8, 0), 8, 1)[:H, :W]
    left = np.clip(base + rs.randint(-31, 32, size=(H, W, 3)), 0, 255).astype(np.float32)
    disp_rows = (8 + 100 * (np.arange(H) / H)).astype(np.int64)                 # 8 .. 107 px, growing downwards
    cols = np.arange(W)[None, :] + disp_rows[:, None]                           # right[x] = left[x + d]  (x_r = x_l - d)
    right = left[np.arange(H)[:, None], np.clip(cols, 0, W - 1)]
    right = np.clip(right + rs.randint(-4, 5, size=(H, W, 3)), 0, 255).astype(np.float32)
    disp = np.broadcast_to(disp_rows[:, None].astype(np.float32), (H, W)).copy()
    disp[rs.rand(H, W) < 0.05] = 0.0                                            # some invalid pixels (mask edges)
    return np.concatenate([left, right, disp[..., None]], 2).astype(np.float32)
