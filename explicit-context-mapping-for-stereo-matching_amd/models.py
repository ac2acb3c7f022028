"""Drop-in mirror of the reference's `cmf.models` interface for the accelerated path.

Same class names, constructor signatures, attribute tree (=> identical `state_dict()` keys/shapes, so
reference checkpoints load), same `forward` signatures and return shapes as
`cmf/models/cmfsm.py` -- but every hot-path stage (cost volume, 3-D aggregation, soft-argmin, ECM
weights, ECM aggregation) runs on the gfx950 kernels in `ops.py`, and so do the 2-D encoder's convolutions and
GroupNorms; PyTorch-ROCm supplies the pooling / interpolation / concatenation glue around them.

Deliberate deviations (documented in DESIGN.md):
  * device-agnostic construction (no `.cuda()` inside modules, cf. cmfsm.py:98,117,427,671);
  * batch > 1 returns [B,1,H,W] (the reference broadcasts to [B,B,H,W], quirk Q1, cmfsm.py:714);
  * odd hr/lr scale raises ValueError instead of `exit()` (cmfsm.py:448-449);
  * the per-forward debug print + host sync (cmfsm.py:581-583) is dropped.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

NUM_GROUPS = 32  # cmfsm.py:31-33

# True: build the reference's explicit [B,2C,D',h,w] concat volume (ops.cost_volume) and run dres0's first Conv3d on it,
# exactly the reference's op sequence (cmfsm.py:667-684).  False (default): the same result from class-indexed 2-D
# convolutions of the feature maps without the 4-D tensor (ops.costvol_conv3d).  Env ECM_EXPLICIT_COST_VOLUME=1 or
# set `models.EXPLICIT_COST_VOLUME = True`.
import os as _os
EXPLICIT_COST_VOLUME = _os.environ.get("ECM_EXPLICIT_COST_VOLUME", "0") == "1"
# Nothing in this package falls back to a vendor library silently: a layer outside the native kernels raises unless
# ECM_ALLOW_FALLBACK=1, and every time a slower-than-designed path IS taken (that opt-in, or a dilated stage whose map is
# not divisible by its dilation and therefore runs on the direct dilated kernels instead of phase planes) an entry lands
# here and a warning is issued once per cause.
ALLOW_FALLBACK = _os.environ.get("ECM_ALLOW_FALLBACK", "0") == "1"
SLOW_PATH_EVENTS = []
_WARNED = set()


def _note_slow_path(kind, detail):
    SLOW_PATH_EVENTS.append((kind,) + tuple(detail))
    if kind not in _WARNED:
        _WARNED.add(kind)
        import warnings
        warnings.warn(f"ecm: {kind} {detail}: running on the slower native path (recorded in models.SLOW_PATH_EVENTS)")


# ------------------------------------------------------------------------------------------------
# leaf layers: nn.Conv3d / nn.ConvTranspose3d / nn.GroupNorm subclasses so parameter names, shapes and
# layouts are exactly the reference's, with forward() on the HIP kernels.
# ------------------------------------------------------------------------------------------------
class HipConv3d(nn.Conv3d):
    def forward(self, x, fork=False):
        """fork=True: returns (y, x') -- x' is x for the skip connection, its gradient folded into this layer's data-gradient
        kernel (ops._fork_out)."""
        return ops.conv3d_k3(x, self.weight, self.stride[0], fork)


class HipConvTranspose3d(nn.ConvTranspose3d):
    def forward(self, x):
        return ops.deconv3d_k3s2(x, self.weight)


class HipGroupNorm(nn.GroupNorm):
    """GroupNorm(32, C) on 5-D volumes; `fused()` adds the residual/ReLU the reference applies right after."""

    def forward(self, x):
        return ops.group_norm_act(x, self.weight, self.bias, None, False)

    def fused(self, x, skip=None, relu=False, head=0):
        return ops.group_norm_act(x, self.weight, self.bias, skip, relu, head)


class HipReLU(nn.ReLU):
    """Placeholder keeping the reference's Sequential indices; fused into the preceding GroupNorm kernel."""


class EncConv2d(nn.Conv2d):
    """nn.Conv2d of the encoder (same parameters, hence the reference's state_dict keys).  Every layer shape of the
    registered architectures -- 3x3 with stride 1|2 and dilation 1|2|4, the 3-channel stem, the 64/128/320/384-channel
    stages, the 1x1 projections -- runs on the MFMA implicit-GEMM family (ops.conv2d: forward, data and weight gradient);
    a configuration outside it (none in the registered models) RAISES: the vendor-library path (nn.Conv2d's own forward on
    MIOpen) exists only behind ECM_ALLOW_FALLBACK=1, so a slow path can never be taken unseen."""

    def _native(self):
        kh, kw = self.kernel_size
        s, d = self.stride[0], self.dilation[0]
        return (self.groups == 1 and self.bias is None and self.stride[0] == self.stride[1]
                and self.dilation[0] == self.dilation[1] and self.padding == (d * (kh - 1) // 2, d * (kw - 1) // 2)
                and self.padding_mode == "zeros" and ops.conv2d_supported(self.in_channels, self.out_channels, kh, kw, s, d))

    def forward(self, x, fork=False):
        """fork=True: returns (y, x') -- x' is x for the skip connection (ops._fork_out)."""
        if x.dim() == 5:       # [B,C,d*d,H/d,W/d]: the phase planes of a dilation-d layer (feature_extraction._run_layer)
            return ops.conv2d_planes(x, self.weight, fork)
        if self._native():     # raises on CPU tensors, like every op
            return ops.conv2d(x, self.weight, self.stride[0], self.dilation[0], fork=fork)
        if not ALLOW_FALLBACK:
            raise RuntimeError(
                f"EncConv2d({self.in_channels}->{self.out_channels}, k={self.kernel_size}, s={self.stride}, d={self.dilation}, "
                f"groups={self.groups}, bias={self.bias is not None}) is outside the native 2-D family (ops.conv2d_supported); "
                "set ECM_ALLOW_FALLBACK=1 to run it on PyTorch-ROCm's own convolution instead")
        SLOW_PATH_EVENTS.append(("miopen_conv2d", self.in_channels, self.out_channels, self.kernel_size))
        y = super().forward(x)
        return (y, x) if fork else y


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation):
    """2-D conv + GroupNorm of the encoder (cmfsm.py:36-46): the convolution on the MFMA 2-D family (EncConv2d), the
    GroupNorm (and the ReLU / residual add that follows it) on the same fused HIP kernel as the 3-D stack."""
    return nn.Sequential(
        EncConv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                  padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
        HipGroupNorm(NUM_GROUPS, out_planes))


def _is_convbn(m):
    return isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[1], HipGroupNorm)


def _seq_fused(seq, x, start=0):
    """Run an encoder nn.Sequential (from module `start`), folding every `GroupNorm -> ReLU` pair into one fused launch."""
    mods = list(seq)
    i = start
    while i < len(mods):
        m = mods[i]
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        if _is_convbn(m):
            x = m[1].fused(m[0](x), None, relu)
            i += 2 if relu else 1
        elif isinstance(m, HipGroupNorm):
            x = m.fused(x, None, relu)
            i += 2 if relu else 1
        elif isinstance(m, nn.Sequential):
            x = _seq_fused(m, x)
            i += 1
        else:
            x = m(x)
            i += 1
    return x


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad):
    """cmfsm.py:49-58: Sequential(Conv3d(bias=False), GroupNorm(32))."""
    if kernel_size != 3 or pad != 1:
        raise ValueError("the HIP path implements the reference's only configuration: kernel 3, pad 1")
    return nn.Sequential(
        HipConv3d(in_planes, out_planes, kernel_size=3, padding=1, stride=stride, bias=False),
        HipGroupNorm(NUM_GROUPS, out_planes))


def _costvol_dres0(dres0, lr_l, lr_r, ndisp):
    """Cost-volume build + dres0's first convbn_3d + ReLU (cmfsm.py:667-684) as one op: the reference-image half of the
    concat volume is constant along d, so its part of the convolution is a class-indexed set of 2-D convolutions of the
    feature map and only the shifted target-image half is materialised (ops.costvol_conv3d).  `ops.cost_volume` is the
    stand-alone builder of the full [B,2C,D,h,w] tensor (kept for the drop-in boundary and the microbench)."""
    conv, gn = dres0[0][0], dres0[0][1]
    if EXPLICIT_COST_VOLUME:
        return gn.fused(conv(ops.cost_volume(lr_l, lr_r, ndisp)), None, True)
    return gn.fused(ops.costvol_conv3d(lr_l, lr_r, conv.weight, ndisp), None, True)


def _cbn(seq, x, skip=None, relu=False, fork=False):
    """Run a convbn_3d Sequential with the trailing residual/ReLU fused into the GroupNorm kernel.
    fork=True: returns (out, x') where x' stands for x in the skip connection that also consumes it."""
    if fork:
        y, xs = seq[0](x, fork=True)
        return seq[1].fused(y, skip, relu), xs
    return seq[1].fused(seq[0](x), skip, relu)


# The tail of a classifier -- GroupNorm + ReLU + Conv3d(32 -> 1), cmfsm.py:621-634 -- as ONE op that never writes the
# normalised tensor (ops.classifier_tail).  ECM_C1_GN_FUSE=0 / models.C1_GN_FUSE = False: the three stages separately.
C1_GN_FUSE = _os.environ.get("ECM_C1_GN_FUSE", "1") != "0"


def _classifier(clf, x, fork=False):
    """clf = Sequential(convbn_3d(32, 32), ReLU, Conv3d(32, 1)).  -> raw logits [B,D,h,w] (and, with fork, x for the other
    consumer of the classifier's input, cf. _cbn)."""
    conv, gn, last = clf[0][0], clf[0][1], clf[2]
    if fork:
        y, xs = conv(x, fork=True)
    else:
        y, xs = conv(x), None
    if C1_GN_FUSE and y.is_cuda and tuple(last.weight.shape) == (1, 32, 3, 3, 3):
        out = ops.classifier_tail(y, gn.weight, gn.bias, last.weight).squeeze(1)
    else:
        out = last(gn.fused(y, None, True)).squeeze(1)
    return (out, xs) if fork else out


# ------------------------------------------------------------------------------------------------
# encoder (cmfsm.py:61-85, 126-236)
# ------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if self.downsample is None:
            # x feeds conv1 AND the residual add: conv1 hands x back so that its data gradient absorbs the skip gradient
            y, x = self.conv1[0][0](x, fork=True)
            out = self.conv1[0][1].fused(y, None, True)                    # GN + ReLU
        else:
            out = _seq_fused(self.conv1, x)                                # conv + GN + ReLU
            x = _seq_fused(self.downsample, x)
        return self.conv2[1].fused(self.conv2[0](out), x, False)            # conv + GN + residual add (cmfsm.py:76-85)


_INTERP_CACHE = {}


def _interp_matrix(n_in, n_out, device, dtype):
    """[n_out, n_in] weights of 1-D linear interpolation with align_corners=False (F.interpolate's source-index rule)."""
    key = (n_in, n_out, str(device), dtype)
    m = _INTERP_CACHE.get(key)
    if m is None:
        # fp32 source coordinates, exactly as ATen computes them (scale = in/out in float)
        dst = torch.arange(n_out, dtype=torch.float32)
        scale = torch.tensor(float(n_in), dtype=torch.float32) / torch.tensor(float(n_out), dtype=torch.float32)
        src = (scale * (dst + 0.5) - 0.5).clamp_(min=0.0)
        i0 = src.floor().long().clamp_(max=n_in - 1)
        i1 = (i0 + 1).clamp_(max=n_in - 1)
        l1 = (src - i0.float()).double()
        m = torch.zeros(n_out, n_in, dtype=torch.float64)
        m.scatter_add_(1, i0.unsqueeze(1), (1.0 - l1).unsqueeze(1))
        m.scatter_add_(1, i1.unsqueeze(1), l1.unsqueeze(1))
        m = m.to(device=device, dtype=dtype)
        _INTERP_CACHE[key] = m
    return m


def bilinear_upsample(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) of the tiny SPP branch maps (cmfsm.py:208-229) as two
    small matmuls with the separable interpolation matrices: same values, but the backward is a matmul instead of
    ATen's atomic scatter (2 ms per call at 144x240 from a 2x3 map)."""
    H, W = size
    uy = _interp_matrix(x.shape[-2], H, x.device, x.dtype)
    ux = _interp_matrix(x.shape[-1], W, x.device, x.dtype)
    return torch.matmul(uy, torch.matmul(x, ux.t()))


# Encoder variants of the registered architectures (all plain PyTorch):
#   first_tail  "conv": firstconv ends with a bare Conv2d and secondconv starts with GroupNorm+ReLU (cmfsm.py:130-145,
#               cm_sub_4.py, bilinear_cmf.py);  "convbn": firstconv ends with convbn+ReLU (cmfsm_sub_8.py:131-145 ...)
#   layers      (stride, dilation) of layer1..4;  pools: AvgPool sizes of branch1..4;  cat_raw: which map joins the
#   pyramid as "output_raw" (layer2 output, or layer3 output for the /16 nets whose last conv is `lastconv_16`).
_ENCODERS = {
    "cmfsm": dict(first_tail="conv", layers=((1, 1), (2, 1), (1, 1), (1, 2)), pools=(64, 32, 16, 8), raw="layer2"),
    "sub4": dict(first_tail="conv", layers=((1, 1), (2, 1), (1, 2), (1, 4)), pools=(64, 32, 16, 8), raw="layer2"),
    "sub8": dict(first_tail="convbn", layers=((2, 1), (2, 1), (1, 2), (1, 4)), pools=(4, 32, 16, 8), raw="layer2"),
    "sub16": dict(first_tail="convbn", layers=((2, 1), (2, 1), (2, 1), (1, 4)), pools=(4, 2, 16, 8), raw="layer3"),
}


def _pyramid_pools(x, pools):
    """The SPP branches' AvgPool2d(k, stride k) of the SAME map (cmfsm.py:152-170) computed hierarchically: the smallest
    window reads the map once, every larger window that is a multiple of a smaller one pools that result (same windows:
    floor(floor(n/a)/r) == floor(n/(a*r)); only the summation order inside a window changes).  The reference reads the
    128-channel map once per branch.  `pools`: the branches' nn.AvgPool2d modules; returns their outputs in that order."""
    ks = [int(p.kernel_size[0] if isinstance(p.kernel_size, tuple) else p.kernel_size) for p in pools]
    done = {}
    for k in sorted(set(ks)):
        base = max((j for j in done if k % j == 0), default=None)
        done[k] = F.avg_pool2d(x, k, k) if base is None else F.avg_pool2d(done[base], k // base, k // base)
    return [done[k] for k in ks]


class feature_extraction(nn.Module):
    """Returns (low-res 32-ch feature, layer1 output, full-res 32-ch firstconv output)  (cmfsm.py:126-236 and the
    per-architecture copies: cmfsm_sub_8.py:126-236, cmfsm_sub_16.py:127-239, cm_sub_4.py:126-236)."""

    def __init__(self, variant="cmfsm"):
        super().__init__()
        cfg = _ENCODERS[variant]
        self.variant = variant
        self.inplanes = 32
        head = [convbn(3, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True)]
        if cfg["first_tail"] == "conv":
            self.firstconv = nn.Sequential(*head, EncConv2d(32, 32, kernel_size=3, padding=1, stride=1, bias=False))
            self.secondconv = nn.Sequential(
                HipGroupNorm(NUM_GROUPS, 32), nn.ReLU(inplace=True),
                convbn(32, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
                convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        else:
            self.firstconv = nn.Sequential(*head, convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
            self.secondconv = nn.Sequential(
                convbn(32, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
                convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        (s1, d1), (s2, d2), (s3, d3), (s4, d4) = cfg["layers"]
        self.layer1 = self._make_layer(BasicBlock, 32, 3, s1, 1, d1)
        self.layer2 = self._make_layer(BasicBlock, 64, 16, s2, 1, d2)
        self.layer3 = self._make_layer(BasicBlock, 128, 3, s3, 1, d3)
        self.layer4 = self._make_layer(BasicBlock, 128, 3, s4, 1, d4)
        for i, pool in enumerate(cfg["pools"], 1):
            setattr(self, f"branch{i}", nn.Sequential(
                nn.AvgPool2d((pool, pool), stride=(pool, pool)), convbn(128, 32, 1, 1, 0, 1), nn.ReLU(inplace=True)))
        last = nn.Sequential(
            convbn(320 if cfg["raw"] == "layer2" else 384, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
            EncConv2d(128, 32, kernel_size=1, padding=0, stride=1, bias=False))
        setattr(self, "lastconv" if cfg["raw"] == "layer2" else "lastconv_16", last)
        self._raw_is_layer3 = cfg["raw"] == "layer3"

    def _make_layer(self, block, planes, blocks, stride, pad, dilation):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                EncConv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                HipGroupNorm(NUM_GROUPS, planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, pad, dilation)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def _run_layer(layer, x):
        """One residual stage.  If every convolution in it is a 3x3 / stride 1 layer of the same dilation d > 1 (cmfsm's
        layer4, cmfsm.py:150), run it on d*d phase planes, where the convolutions are ordinary 3x3 ones (ops.phase_split)."""
        convs = [m for m in layer.modules() if isinstance(m, nn.Conv2d)]
        d = convs[0].dilation[0]
        ok = (ops.WINOGRAD and d > 1 and x.shape[-2] % d == 0 and x.shape[-1] % d == 0 and x.shape[-1] // d >= 2
              and all(m.kernel_size == (3, 3) and m.stride == (1, 1) and m.dilation == (d, d) and m.padding == (d, d) for m in convs))
        if not ok:
            if d > 1 and ops.WINOGRAD:
                # still this library's kernels (the direct dilated 3x3 family), but not the designed fast path: made visible.
                # Only a map whose height / width is not a multiple of the dilation gets here (e.g. the /16 nets' dilation-4
                # stage on a 384x1248 KITTI frame: 24 x 78); no registered architecture does at the SceneFlow size.
                _note_slow_path("dilated_stage_unphased", (d, tuple(x.shape)))
            return layer(x)
        return ops.phase_merge(layer(ops.phase_split(x, d)), d)

    def forward(self, x, head=None):
        """`head` (extension): the caller only needs the first `head` samples of the full-resolution map (third result); their
        gradient is then folded into the map's gradient in place (ops.fork_head) instead of through a zero-padded copy."""
        output_all = _seq_fused(self.firstconv, x)
        output_head = None
        if head is not None and isinstance(self.secondconv[0], HipGroupNorm):
            # the map's whole-batch consumer is secondconv's GroupNorm + ReLU: that node hands the head slice out and folds
            # its gradient into its own data gradient (ops.GroupNormAct `head`)
            y, output_head = self.secondconv[0].fused(output_all, None, True, head=head)
            output_rt = self.layer1(_seq_fused(self.secondconv, y, start=2))
        else:
            if head is not None:
                output_all, output_head = ops.fork_head(output_all, head)
            output_rt = self.layer1(_seq_fused(self.secondconv, output_all))
        output_raw = self.layer2(output_rt)
        if self._raw_is_layer3:                       # cmfsm_sub_16.py:205-207
            output_raw = self._run_layer(self.layer3, output_raw)
            output_skip = self._run_layer(self.layer4, output_raw)
        else:
            output_skip = self._run_layer(self.layer4, self._run_layer(self.layer3, output_raw))
        size = output_skip.shape[-2:]
        pooled = _pyramid_pools(output_skip, [getattr(self, f"branch{i}")[0] for i in (1, 2, 3, 4)])
        pyramid = [bilinear_upsample(_seq_fused(getattr(self, f"branch{i}"), pooled[i - 1], start=1), size) for i in (4, 3, 2, 1)]
        last = self.lastconv_16 if self._raw_is_layer3 else self.lastconv
        feature = _seq_fused(last, torch.cat([output_raw, output_skip] + pyramid, 1))
        return feature, output_rt, (output_all if output_head is None else output_head)


# ------------------------------------------------------------------------------------------------
# hot-path modules
# ------------------------------------------------------------------------------------------------
class matchshifted(nn.Module):
    """One disparity slice of the cost volume, [B,2C,1,H,W] (cmfsm.py:88-108; unused by the reference's forward)."""

    def forward(self, left, right, shift):
        return ops.cost_volume(left, right, shift + 1)[:, :, shift:shift + 1]


class disparityregression(nn.Module):
    """sum_d x[:,d]*d (cmfsm.py:111-123)."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp

    def forward(self, x):
        return ops.disparity_regression(x)


class hourglass(nn.Module):
    """cmfsm.py:240-303."""

    def __init__(self, inplanes):
        super().__init__()
        c2 = inplanes * 2
        self.conv1 = nn.Sequential(convbn_3d(inplanes, c2, kernel_size=3, stride=2, pad=1), HipReLU(inplace=True))
        self.conv2 = convbn_3d(c2, c2, kernel_size=3, stride=1, pad=1)
        self.conv3 = nn.Sequential(convbn_3d(c2, c2, kernel_size=3, stride=2, pad=1), HipReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(c2, c2, kernel_size=3, stride=1, pad=1), HipReLU(inplace=True))
        self.conv5 = nn.Sequential(
            HipConvTranspose3d(c2, c2, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
            HipGroupNorm(NUM_GROUPS, c2))
        self.conv6 = nn.Sequential(
            HipConvTranspose3d(c2, inplanes, kernel_size=3, padding=1, output_padding=(1, 1, 1), stride=2, bias=False),
            HipGroupNorm(NUM_GROUPS, inplanes))

    def forward(self, x, presqu, postsqu, residual=None, pre_uses=1):
        """`residual` (extension): added to `out` inside the last GroupNorm kernel (cmfsm.py:687,690,693).
        `pre_uses` (extension): how many times the CALLER will consume the returned `pre`; with more than one consumer in all
        (conv3, the skip of conv5 when `presqu` is None, the caller's) their gradients are summed by one kernel (ops.fork)
        instead of one autograd add each, and `pre` is returned as a tuple of `pre_uses` aliases."""
        out = _cbn(self.conv1[0], x, relu=True)                                   # :285
        pre = _cbn(self.conv2, out, skip=postsqu, relu=True)                      # :286-290
        inner = 2 if presqu is None else 1
        views = ops.fork(pre, inner + pre_uses) if pre_uses > 1 else (pre,) * (inner + pre_uses)
        out = _cbn(self.conv3[0], views[0], relu=True)                            # :292
        out = _cbn(self.conv4[0], out, relu=True)                                 # :293
        post = _cbn(self.conv5, out, skip=presqu if presqu is not None else views[1], relu=True)   # :295-299
        out = _cbn(self.conv6, post, skip=residual, relu=False)                   # :301
        return out, (pre if pre_uses <= 1 else tuple(views[inner:])), post


class similarity_measure1(nn.Module):
    """Per-pixel MLP as 1x1 convs 66->32->16->8->1, LeakyReLU between, no bias (cmfsm.py:304-358).
    Parameter holder for the fused ECM-weights kernel; calling it directly runs the plain 1x1 convs."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.conv0 = nn.Conv2d(66, 32, kernel_size=1, bias=False)
        self.relu0 = nn.LeakyReLU(inplace=True)
        self.conv1 = nn.Conv2d(32, 16, kernel_size=1, bias=False)
        self.relu1 = nn.LeakyReLU(inplace=True)
        self.conv2 = nn.Conv2d(16, 8, kernel_size=1, bias=False)
        self.relu2 = nn.LeakyReLU(inplace=True)
        self.conv3 = nn.Conv2d(8, 1, kernel_size=1, bias=False)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.relu0(self.conv0(x))
        x = self.relu1(self.conv1(x))
        x = self.relu2(self.conv2(x))
        return self.conv3(x)


class eight_related_context_mapping(nn.Module):
    """cmfsm.py:431-593 -> nine softmax planes [B,1,H,W] (centre,l,r,t,b,lt,rt,lb,rb)."""

    def __init__(self):
        super().__init__()
        self.similarity1 = similarity_measure1()
        self.sigmoid = nn.Sigmoid()

    def weights(self, lr_feature, hr_feature):
        m = self.similarity1
        return ops.ecm_weights9(lr_feature, hr_feature, m.conv0.weight, m.conv1.weight, m.conv2.weight, m.conv3.weight)

    def forward(self, lr_feature, hr_feature, lr_feature_r=None, hr_feature_r=None):
        w9 = self.weights(lr_feature, hr_feature)          # right-image inputs are ignored (cmfsm.py:474-480)
        return tuple(w9[:, n:n + 1] for n in range(9))


class cmfsm(nn.Module):
    """cmfsm.py:594-774.  forward(left, right) -> (pred1, pred2, pred3), each [B,1,H,W] in pixels."""

    def __init__(self, maxdisp=192):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature_extraction = feature_extraction()
        self.dres0 = nn.Sequential(convbn_3d(64, 32, 3, 1, 1), HipReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1), HipReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), HipReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        for i in (1, 2, 3):
            setattr(self, f"classif{i}", nn.Sequential(
                convbn_3d(32, 32, 3, 1, 1), HipReLU(inplace=True),
                HipConv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)))
        self.mapping_matrix = eight_related_context_mapping()
        # the reference re-initialises every Conv2d/Conv3d with the PSMNet rule (cmfsm.py:638-645)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.Conv3d):
                k = m.kernel_size
                m.weight.data.normal_(0, math.sqrt(2.0 / (k[0] * k[1] * k[2] * m.out_channels)))

    def hot_path(self, lr_l, hr_l, lr_r):
        """Everything after the encoder (cmfsm.py:659-774)."""
        scale = hr_l.shape[-1] // lr_l.shape[-1]
        w9 = self.mapping_matrix.weights(lr_l, hr_l)                                   # :664
        cost0 = _costvol_dres0(self.dres0, lr_l, lr_r, self.maxdisp // scale)          # :667-684
        cost0 = _cbn(self.dres0[2], cost0, relu=True)
        y, cost0 = _cbn(self.dres1[0], cost0, relu=True, fork=True)                    # :685
        cost0 = _cbn(self.dres1[2], y, skip=cost0)
        # out1 / out2 feed the next hourglass AND their classifier: the classifier's first convolution hands them back
        # (fork), so that its data gradient absorbs the gradient arriving through the hourglass
        # cost0 has four consumers (first hourglass + three residual adds): one 4-ary gradient sum instead of three adds
        c_in, c_r1, c_r2, c_r3 = ops.fork(cost0, 4)
        # pre1 has four consumers (two inside dres2, the skips of dres3 and dres4): one 4-ary gradient sum
        out1, (pre1a, pre1b), post1 = self.dres2(c_in, None, None, residual=c_r1, pre_uses=2)   # :686-687
        c1, out1 = _classifier(self.classif1, out1, fork=True)                         # :695
        out2, pre2, post2 = self.dres3(out1, pre1a, post1, residual=c_r2)              # :689-690
        c2, out2 = _classifier(self.classif2, out2, fork=True)                         # :724
        out3, pre3, post3 = self.dres4(out2, pre1b, post2, residual=c_r3)              # :692-693
        c3 = _classifier(self.classif3, out3)                                          # :747
        disp = ops.softargmin_heads(torch.stack([c1, c2, c3], 0))                      # :703-706,725-728,748-753
        preds = ops.ecm_aggregate9(disp, w9, scale)                                    # :709-723 (x3)
        return preds[0].unsqueeze(1), preds[1].unsqueeze(1), preds[2].unsqueeze(1)

    def forward(self, left, right):
        # cmfsm.py:657-658 runs the shared encoder twice; both images go through it as ONE batch here (GroupNorm has no
        # cross-sample statistics, so the result is identical) -- half the launches, better-filled small layers.
        B = left.shape[0]
        if torch.is_grad_enabled():
            ops.pace_side_streams()        # host run-ahead bound: the previous step's side-stream weight gradients have finished
        lr, _, hr = self.feature_extraction(torch.cat([left, right], 0), head=B)      # hr: the left images' map only
        return self.hot_path(lr[:B], hr, lr[B:])


class similarity_measure2(nn.Module):
    """3->3->2->1 1x1-conv MLP on the offset table (cm_sub_4.py: instantiated, never used in forward)."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.conv0 = nn.Conv2d(3, 3, kernel_size=1, bias=False)
        self.relu0 = nn.LeakyReLU(inplace=True)
        self.conv1 = nn.Conv2d(3, 2, kernel_size=1, bias=False)
        self.relu1 = nn.LeakyReLU(inplace=True)
        self.conv2 = nn.Conv2d(2, 1, kernel_size=1, bias=False)
        self.relu2 = nn.LeakyReLU(inplace=True)

    def forward(self, x):
        return self.relu2(self.conv2(self.relu1(self.conv1(self.relu0(self.conv0(x))))))


class six_related_context_mapping(nn.Module):
    """cmfsm_sub_8.py:440-572 (same text in cmfsm_sub_16.py and cm_sub_4/8/16.py):
    forward(lr, hr, lr_r, hr_r) -> ((mapping, mapping_r, mapping_l, mapping_t, mapping_b),
                                    (mapping_target, mapping_target_r, mapping_target_l)), each [B,1,H,W]."""

    def __init__(self, with_similarity2=False):
        super().__init__()
        self.similarity1 = similarity_measure1()
        self.similarity1.relu3 = nn.LeakyReLU(inplace=True)          # cmfsm_sub_8.py:318 (no parameters)
        if with_similarity2:
            self.similarity2 = similarity_measure2()                 # cm_sub_4.py only; unused in forward
        self.fuse = nn.Sequential(nn.Conv2d(2, 1, kernel_size=1, bias=False), nn.LeakyReLU(inplace=True))   # unused

    def planes(self, lr_feature, hr_feature, lr_feature_r, hr_feature_r):
        m = self.similarity1
        ws = (m.conv0.weight, m.conv1.weight, m.conv2.weight, m.conv3.weight)
        return (ops.context_weights(lr_feature, hr_feature, *ws, 1), ops.context_weights(lr_feature_r, hr_feature_r, *ws, 2))

    def forward(self, lr_feature, hr_feature, lr_feature_r, hr_feature_r):
        m5, mt3 = self.planes(lr_feature, hr_feature, lr_feature_r, hr_feature_r)
        return tuple(m5[:, n:n + 1] for n in range(5)), tuple(mt3[:, n:n + 1] for n in range(3))


class _ECMNet(nn.Module):
    """Shared skeleton of the registered architectures: encoder -> cost volume -> dres0/1 -> 1 or 3 hourglasses ->
    classifiers -> head.  Subclasses set ENCODER, HOURGLASSES and HEAD exactly as their reference file does."""
    ENCODER, HOURGLASSES, HEAD, SIM2 = "cmfsm", 3, "eight", False

    def __init__(self, maxdisp=192):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature_extraction = feature_extraction(self.ENCODER)
        self.dres0 = nn.Sequential(convbn_3d(64, 32, 3, 1, 1), HipReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1), HipReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), HipReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1))
        for i in range(self.HOURGLASSES):
            setattr(self, f"dres{i + 2}", hourglass(32))
        for i in range(self.HOURGLASSES):
            setattr(self, f"classif{i + 1}", nn.Sequential(
                convbn_3d(32, 32, 3, 1, 1), HipReLU(inplace=True),
                HipConv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)))
        if self.HEAD in ("five", "volume"):
            self.mapping_matrix = six_related_context_mapping(self.SIM2)
        for m in self.modules():                                      # PSMNet init rule, e.g. cmfsm_sub_8.py:703-711
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.Conv3d):
                k = m.kernel_size
                m.weight.data.normal_(0, math.sqrt(2.0 / (k[0] * k[1] * k[2] * m.out_channels)))

    def hot_path(self, lr_l, hr_l, lr_r, hr_r, out_hw=None):
        scale = hr_l.shape[-1] // lr_l.shape[-1]
        planes = None
        if self.HEAD in ("five", "volume"):
            planes = self.mapping_matrix.planes(lr_l, hr_l, lr_r, hr_r)
        cost0 = _costvol_dres0(self.dres0, lr_l, lr_r, self.maxdisp // scale)
        cost0 = _cbn(self.dres0[2], cost0, relu=True)
        y, cost0 = _cbn(self.dres1[0], cost0, relu=True, fork=True)
        cost0 = _cbn(self.dres1[2], y, skip=cost0)
        c_forks = ops.fork(cost0, self.HOURGLASSES + 1)     # first hourglass + one residual add per hourglass
        heads, x, pre1, post = [], c_forks[0], None, None
        for i in range(self.HOURGLASSES):
            out, pre, post = getattr(self, f"dres{i + 2}")(x, pre1 if i > 0 else None, post, residual=c_forks[i + 1])
            if i == 0:
                pre1 = pre
            clf = getattr(self, f"classif{i + 1}")
            if i + 1 < self.HOURGLASSES:       # `out` also feeds the next hourglass: the classifier's conv hands it back
                c_i, out = _classifier(clf, out, fork=True)
            else:
                c_i = _classifier(clf, out)
            x = out
            heads.append(c_i)
        c = torch.stack(heads, 0)                                        # raw classifier outputs [NH,B,Dl,h,w]
        if self.HEAD == "five":                                          # cmfsm_sub_8.py:757-803 (heads NOT accumulated)
            disp = torch.cat([ops.softargmin_heads(c[k:k + 1]) for k in range(c.shape[0])], 0)
            m5 = planes[0]
            w9 = torch.cat([m5[:, 0:1], m5[:, 2:3], m5[:, 1:2], m5[:, 3:5], torch.zeros_like(m5[:, :4])], 1)
            preds = ops.ecm_aggregate9(disp, w9, scale)                  # planes reordered c,l,r,t,b + 4 zero diagonals
            return tuple(preds[k].unsqueeze(1) for k in range(3))
        if self.HEAD == "volume":                                        # cmfsm_sub_16.py:767-848 / cm_sub_8.py:765-800
            preds = ops.volume_mapping(c, planes[0], planes[1], scale)
        else:                                                            # bilinear_cmf.py:447-471
            H, W = out_hw if out_hw is not None else hr_l.shape[-2:]
            preds = ops.trilinear_softargmin(c, self.maxdisp, H, W)
        preds = [preds[k] for k in range(preds.shape[0])]
        while len(preds) < 3:                                            # cm_sub_*: `return pred1, pred1, pred1`
            preds.append(preds[0])
        return tuple(preds)

    def forward(self, left, right):
        B = left.shape[0]
        if torch.is_grad_enabled():
            ops.pace_side_streams()
        lr, _, hr = self.feature_extraction(torch.cat([left, right], 0))      # one encoder pass for both images
        return self.hot_path(lr[:B], hr[:B], lr[B:], hr[B:], out_hw=left.shape[-2:])


class cmfsm_sub_8(_ECMNet):
    ENCODER, HOURGLASSES, HEAD = "sub8", 3, "five"


class cmfsm_sub_16(_ECMNet):
    ENCODER, HOURGLASSES, HEAD = "sub16", 3, "volume"


class cm_sub_4(_ECMNet):
    ENCODER, HOURGLASSES, HEAD, SIM2 = "sub4", 1, "volume", True


class cm_sub_8(_ECMNet):
    ENCODER, HOURGLASSES, HEAD = "sub8", 1, "volume"


class cm_sub_16(_ECMNet):
    ENCODER, HOURGLASSES, HEAD = "sub16", 1, "volume"


class bilinear_cmf(_ECMNet):
    ENCODER, HOURGLASSES, HEAD = "sub4", 3, "trilinear"


class bilinear_cmf_sub_8(_ECMNet):
    ENCODER, HOURGLASSES, HEAD = "sub8", 3, "trilinear"


class bilinear_cmf_sub_16(_ECMNet):
    ENCODER, HOURGLASSES, HEAD = "sub16", 3, "trilinear"


_MODELS = {"cmfsm": cmfsm, "cmfsm_sub_8": cmfsm_sub_8, "cmfsm_sub_16": cmfsm_sub_16, "cm_sub_4": cm_sub_4,
           "cm_sub_8": cm_sub_8, "cm_sub_16": cm_sub_16, "bilinear_cmf": bilinear_cmf,
           "bilinear_cmf_sub_8": bilinear_cmf_sub_8, "bilinear_cmf_sub_16": bilinear_cmf_sub_16}



def get_model(name):
    """cmf/models/__init__.py:19-41: returns `cls()`; an unknown name prints and returns None like the reference."""
    cls = _MODELS.get(name)
    if cls is None:
        print("Model {} not available".format(name))
        return None
    return cls()
