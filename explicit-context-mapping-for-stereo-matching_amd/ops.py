"""torch.autograd.Function skins over the C ABI of libecm_hip.so.

PyTorch is plumbing here (device memory from its caching allocator, the current HIP stream,
autograd bookkeeping); every forward and backward below is a hand-written gfx950 kernel reached
through `_lib.call`.  There is no eager/CPU fallback: a non-CUDA tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

GN_GROUPS = 32
GN_EPS = 1e-5


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("ecm ops run only on the MI355X HIP path: got a CPU tensor (no CPU fallback exists)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"ecm ops are fp32 (got {t.dtype})")


def _need(cond, msg):
    """Operand-shape contract of an op.  The C ABI sees pointers and sizes only: whatever the sizes do not say about a second
    operand has to be established here, BEFORE the launch -- a kernel reading past a smaller tensor is a device fault, not an
    exception.  RuntimeError, as the torch ops these replace raise on mismatched shapes."""
    if not cond:
        raise RuntimeError(msg() if callable(msg) else msg)


def _c(t):
    """Contiguous and 16-byte aligned (the kernels use float4 accesses)."""
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _empty_like(t):
    """A CONTIGUOUS uninitialised tensor of t's shape: the kernels write dense row-major buffers, and
    `torch.empty_like` would copy the strides of a non-contiguous `t` (e.g. an einsum output viewed as a weight)."""
    return torch.empty(t.shape, dtype=t.dtype, device=t.device)


def _scratch(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------ a1 cost volume
class CostVolumeConcat(torch.autograd.Function):
    """cmfsm.py:667-682 -- [B,C,h,w] x2 -> [B,2C,D,h,w]."""

    @staticmethod
    def forward(ctx, left, right, ndisp):
        _chk(left, right)
        _need(left.dim() == 4 and right.shape == left.shape and ndisp >= 1,
              lambda: f"cost_volume: left {tuple(left.shape)}, right {tuple(right.shape)}, ndisp {ndisp}: two [B,C,h,w] maps of one shape")
        left, right = _c(left), _c(right)
        B, Cc, h, w = left.shape
        cost = torch.empty(B, 2 * Cc, ndisp, h, w, device=left.device, dtype=left.dtype)
        _lib.call("ecm_costvol_concat_fwd", _p(left), _p(right), _p(cost), B, Cc, h, w, ndisp, _stream())
        ctx.dims = (B, Cc, h, w, ndisp)
        return cost

    @staticmethod
    def backward(ctx, gcost):
        B, Cc, h, w, D = ctx.dims
        gcost = _c(gcost)
        gl = torch.empty(B, Cc, h, w, device=gcost.device, dtype=gcost.dtype)
        gr = _empty_like(gl)
        _lib.call("ecm_costvol_concat_bwd", _p(gcost), _p(gl), _p(gr), B, Cc, h, w, D, _stream())
        return gl, gr, None


def cost_volume(left, right, ndisp):
    return CostVolumeConcat.apply(left, right, int(ndisp))


class CostvolConvAssemble(torch.autograd.Function):
    """y[b,co,d,h,x] = P[b,classP(d,x),co,h,x] + Qp[b,classQ(d,x),co,h,x-d+2] (see csrc/costvol_conv.hip);
    backward: gP, gQp = per-class sums of gy over d."""

    @staticmethod
    def forward(ctx, P, Qp, ndisp):
        _chk(P, Qp)
        P, Qp = _c(P), _c(Qp)
        _need(P.dim() == 4 and Qp.dim() == 4, "costvol_conv assemble: P and Qp are [B,classes*Co,h,w(+2)]")
        B, c15, h, w = P.shape
        Co = c15 // 15
        _need(Co >= 1 and c15 == 15 * Co and tuple(Qp.shape) == (B, 6 * Co, h, w + 2) and ndisp >= 2,
              lambda: f"costvol_conv assemble: P {tuple(P.shape)} / Qp {tuple(Qp.shape)} / D {ndisp}: want [B,15Co,h,w], [B,6Co,h,w+2], D >= 2")
        y = torch.empty(B, Co, ndisp, h, w, device=P.device, dtype=P.dtype)
        _lib.call("ecm_costvol_conv_assemble_fwd", _p(P), _p(Qp), _p(y), B, Co, ndisp, h, w, _stream())
        ctx.dims = (B, Co, ndisp, h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, Co, D, h, w = ctx.dims
        gy = _c(gy)
        gP = torch.empty(B, 15 * Co, h, w, device=gy.device, dtype=gy.dtype)
        gQ = torch.empty(B, 6 * Co, h, w + 2, device=gy.device, dtype=gy.dtype)
        _lib.call("ecm_costvol_conv_assemble_bwd", _p(gy), _p(gP), _p(gQ), B, Co, D, h, w, _stream())
        return gP, gQ, None


class ClassWeights(torch.autograd.Function):
    """dres0[0][0].weight [Co,2C,3,3,3] -> (wP [15*Co,C,3,3], wQ [6*Co,C,3,5]): the class-indexed 2-D kernels of the collapsed
    first convolution (csrc/costvol_conv.hip).  Class = which taps pass the wedge `x >= d` of cmfsm.py:678-679 and the depth
    padding: reference half (clamp(d-x,-2,2)+2)*3 + edge with edge = 0 / 1 / 2 for the first / an interior / the last
    disparity plane; target half edge*2 + (x == w-1), a passing tap landing on column kw-kd+2 of the sheared 3x5 kernel."""

    @staticmethod
    def forward(ctx, w):
        _chk(w)
        _need(w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[1] % 2 == 0,
              lambda: f"class weights: w {tuple(w.shape)}: want the first 3x3x3 convolution's [Co,2C,3,3,3]")
        w = _c(w)
        Co, C2 = w.shape[:2]
        Cc = C2 // 2
        wP = torch.empty(15 * Co, Cc, 3, 3, device=w.device, dtype=w.dtype)
        wQ = torch.empty(6 * Co, Cc, 3, 5, device=w.device, dtype=w.dtype)
        _lib.call("ecm_costvol_class_weights_fwd", _p(w), _p(wP), _p(wQ), Co, Cc, _stream())
        ctx.dims = (Co, Cc)
        return wP, wQ

    @staticmethod
    def backward(ctx, gwP, gwQ):
        Co, Cc = ctx.dims
        gw = torch.empty(Co, 2 * Cc, 3, 3, 3, device=gwP.device, dtype=gwP.dtype)
        _lib.call("ecm_costvol_class_weights_bwd", _p(_c(gwP)), _p(_c(gwQ)), _p(gw), Co, Cc, _stream())
        return gw


def costvol_conv3d(left, right, weight, ndisp):
    """conv3d(cost_volume(left, right, ndisp), weight, stride 1, pad 1)  (cmfsm.py:667-684) WITHOUT the 4-D volume: both
    halves of the concat volume are constant along a line in (d, x), so the 3x3x3 convolution collapses to class-indexed
    2-D convolutions of the two feature maps (3x3 on `left`, sheared 3x5 on `right`) -- see csrc/costvol_conv.hip.
    weight: [Co, 2C, 3, 3, 3] (the reference's dres0[0][0].weight)."""
    _need(left.dim() == 4 and right.shape == left.shape and weight.dim() == 5,
          lambda: f"costvol_conv3d: left {tuple(left.shape)}, right {tuple(right.shape)}, weight {tuple(weight.shape)}")
    Cc = left.shape[1]
    if weight.shape[1] != 2 * Cc or ndisp < 2:
        return conv3d_k3(cost_volume(left, right, ndisp), weight, 1)
    wP, wQ = ClassWeights.apply(weight)
    h, w = left.shape[-2:]
    P = conv2d(left, wP, 1, 1, 1, 1, h, w)                      # 3x3, pad 1                                  [B,15Co,h,w]
    # sheared 3x5 on the target features with 2 extra zero columns on the left: pad (1, 2) of F.pad(right, (2, 0)) is a
    # left padding of 4 and a right padding of 2 -> output width w + 2                                        [B,6Co,h,w+2]
    Qp = conv2d(right, wQ, 1, 1, 1, 4, h, w + 2)
    return CostvolConvAssemble.apply(P, Qp, int(ndisp))


# ------------------------------------------------------------------------------------ a8 soft-argmin
class SoftArgminHeads(torch.autograd.Function):
    """c [NH,B,D,h,w] raw classifier outputs -> disp [NH,B,h,w]; head k uses logits c_0+...+c_k
    (cmfsm.py:703-706, 725-728, 748-753)."""

    @staticmethod
    def forward(ctx, c):
        _chk(c)
        _need(c.dim() == 5 and c.numel() > 0, lambda: f"softargmin_heads: c {tuple(c.shape)}: want [heads,B,D,h,w]")
        c = _c(c)
        NH, B, D, h, w = c.shape
        disp = torch.empty(NH, B, h, w, device=c.device, dtype=c.dtype)
        _lib.call("ecm_softargmin_heads_fwd", _p(c), C.c_longlong(B * D * h * w), _p(disp), NH, B, D, h * w, _stream())
        ctx.save_for_backward(c)
        return disp

    @staticmethod
    def backward(ctx, gdisp):
        (c,) = ctx.saved_tensors
        NH, B, D, h, w = c.shape
        gdisp = _c(gdisp)
        gc = _empty_like(c)
        _lib.call("ecm_softargmin_heads_bwd", _p(c), C.c_longlong(B * D * h * w), _p(gdisp), _p(gc), NH, B, D, h * w,
                  _stream())
        return gc


def softargmin_heads(c):
    return SoftArgminHeads.apply(c)


def disparity_regression(x):
    """disparityregression.forward (cmfsm.py:120-123): [B,D,h,w] probabilities -> [B,h,w]. Forward only."""
    _chk(x)
    _need(x.dim() == 4 and x.numel() > 0, lambda: f"disparity_regression: x {tuple(x.shape)}: want [B,D,h,w]")
    x = _c(x)
    B, D, h, w = x.shape
    out = torch.empty(B, h, w, device=x.device, dtype=x.dtype)
    _lib.call("ecm_disparity_regression_fwd", _p(x), _p(out), B, D, h * w, _stream())
    return out


# ------------------------------------------------------------------------------------ a9 aggregation
class ECMAggregate9(torch.autograd.Function):
    """d [NH,B,h,w], w9 [B,9,H,W] -> [NH,B,H,W]   (cmfsm.py:709-723)."""

    @staticmethod
    def forward(ctx, d, w9, scale):
        _chk(d, w9)
        d, w9 = _c(d), _c(w9)
        _need(d.dim() == 4 and scale >= 1, lambda: f"ecm_aggregate9: d {tuple(d.shape)}, scale {scale}: want [heads,B,h,w]")
        NH, B, h, w = d.shape
        _need(tuple(w9.shape) == (B, 9, h * scale, w * scale),
              lambda: f"ecm_aggregate9: w9 {tuple(w9.shape)} for d {tuple(d.shape)} at scale {scale}: want [B,9,h*scale,w*scale]")
        out = torch.empty(NH, B, h * scale, w * scale, device=d.device, dtype=d.dtype)
        _lib.call("ecm_aggregate9_fwd", _p(d), _p(w9), _p(out), NH, B, h, w, scale, _stream())
        ctx.save_for_backward(d, w9)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, gout):
        d, w9 = ctx.saved_tensors
        NH, B, h, w = d.shape
        gout = _c(gout)
        gd, gw9 = _empty_like(d), _empty_like(w9)
        _lib.call("ecm_aggregate9_bwd", _p(d), _p(w9), _p(gout), _p(gd), _p(gw9), NH, B, h, w, ctx.scale, _stream())
        return gd, gw9, None


def ecm_aggregate9(d, w9, scale):
    return ECMAggregate9.apply(d, w9, int(scale))


# ------------------------------------------------------------------------------------ a3 ECM weights
_ECM_MLP_SIZES = (2112, 512, 128, 8)      # similarity_measure1's four 1x1 kernels: 66->32, 32->16, 16->8, 8->1
class ECMWeights9(torch.autograd.Function):
    """eight_related_context_mapping (cmfsm.py:443-593): lr [B,32,h,w], hr [B,32,H,W] -> w9 [B,9,H,W]."""

    @staticmethod
    def forward(ctx, lr, hr, W0, W1, W2, W3):
        _chk(lr, hr, W0, W1, W2, W3)
        _need(lr.dim() == 4 and hr.dim() == 4 and hr.shape[0] == lr.shape[0] and lr.shape[-1] > 0 and lr.shape[-2] > 0,
              lambda: f"context weights: lr {tuple(lr.shape)}, hr {tuple(hr.shape)}: want [B,32,h,w] and [B,32,H,W]")
        _need((W0.numel(), W1.numel(), W2.numel(), W3.numel()) == _ECM_MLP_SIZES,
              lambda: f"context weights: the similarity MLP's kernels hold {_ECM_MLP_SIZES} values (cmfsm.py:400-427), got "
                      f"{(W0.numel(), W1.numel(), W2.numel(), W3.numel())}")
        lr, hr = _c(lr), _c(hr)
        W0, W1, W2, W3 = (_c(t) for t in (W0, W1, W2, W3))
        B, Cc, h, w = lr.shape
        H, W = hr.shape[-2:]
        s = W // w
        if s % 2 != 0:
            raise ValueError("odd scale between hr and lr features (the reference calls exit() here, cmfsm.py:448-449)")
        if Cc != 32 or hr.shape[1] != 32 or H != h * s or W != w * s:
            raise RuntimeError(f"ecm_weights9: unsupported shapes lr {tuple(lr.shape)} hr {tuple(hr.shape)}")
        if s != 4:      # matrix_generation hard-codes scale 4 (cmfsm.py:392): the reference's forward fails in torch.cat otherwise
            raise RuntimeError(f"eight_related_context_mapping exists at scale 4 only (got {s}): the reference's offset tables "
                               "are 4 x 4 and its forward raises a shape error at any other scale")
        w9 = torch.empty(B, 9, H, W, device=lr.device, dtype=lr.dtype)
        nb = _lib.query("ecm_weights9_scratch_bytes", B, h, w)
        scratch = _scratch(nb, lr.device)
        _lib.call("ecm_weights9_fwd", _p(lr), _p(hr), _p(W0), _p(W1), _p(W2), _p(W3), _p(w9), _p(scratch),
                  C.c_longlong(nb), B, h, w, s, _stream())
        ctx.save_for_backward(lr, hr, W0, W1, W2, W3, w9)
        ctx.s = s
        return w9

    @staticmethod
    def backward(ctx, gw9):
        lr, hr, W0, W1, W2, W3, w9 = ctx.saved_tensors
        B, _, h, w = lr.shape
        s = ctx.s
        gw9 = _c(gw9)
        glr, ghr = _empty_like(lr), _empty_like(hr)
        gW = torch.empty(2112 + 512 + 128 + 8, device=lr.device, dtype=lr.dtype)
        nb = _lib.query("ecm_weights9_bwd_scratch_bytes", B, h, w, s)
        scratch = _scratch(nb, lr.device)
        _lib.call("ecm_weights9_bwd", _p(lr), _p(hr), _p(W0), _p(W1), _p(W2), _p(W3), _p(w9), _p(gw9), _p(glr), _p(ghr),
                  _p(gW), _p(scratch), C.c_longlong(nb), B, h, w, s, _stream())
        return (glr, ghr, gW[:2112].view_as(W0), gW[2112:2624].view_as(W1), gW[2624:2752].view_as(W2),
                gW[2752:].view_as(W3))


def ecm_weights9(lr, hr, W0, W1, W2, W3):
    return ECMWeights9.apply(lr, hr, W0, W1, W2, W3)


_VARIANT_PLANES = {0: 9, 1: 5, 2: 3}


class ContextWeights(torch.autograd.Function):
    """General context-mapping weights: variant 0 eight_related (cmfsm.py:431-593), 1/2 six_related on the reference /
    target image (cmfsm_sub_8.py:440-572).  lr [B,32,h,w], hr [B,32,H,W] -> [B,N,H,W]."""

    @staticmethod
    def forward(ctx, lr, hr, W0, W1, W2, W3, variant):
        _chk(lr, hr, W0, W1, W2, W3)
        _need(lr.dim() == 4 and hr.dim() == 4 and hr.shape[0] == lr.shape[0] and lr.shape[-1] > 0 and lr.shape[-2] > 0,
              lambda: f"context weights: lr {tuple(lr.shape)}, hr {tuple(hr.shape)}: want [B,32,h,w] and [B,32,H,W]")
        _need((W0.numel(), W1.numel(), W2.numel(), W3.numel()) == _ECM_MLP_SIZES,
              lambda: f"context weights: the similarity MLP's kernels hold {_ECM_MLP_SIZES} values (cmfsm.py:400-427), got "
                      f"{(W0.numel(), W1.numel(), W2.numel(), W3.numel())}")
        lr, hr = _c(lr), _c(hr)
        W0, W1, W2, W3 = (_c(t) for t in (W0, W1, W2, W3))
        B, Cc, h, w = lr.shape
        H, W = hr.shape[-2:]
        s = W // w
        if s % 2 != 0:
            raise ValueError("odd scale between hr and lr features (the reference calls exit() here)")
        if Cc != 32 or hr.shape[1] != 32 or H != h * s or W != w * s:
            raise RuntimeError(f"context_weights: unsupported shapes lr {tuple(lr.shape)} hr {tuple(hr.shape)}")
        _need(variant in _VARIANT_PLANES, lambda: f"context weights: variant {variant} (0 eight-related, 1 / 2 six-related)")
        out = torch.empty(B, _VARIANT_PLANES[variant], H, W, device=lr.device, dtype=lr.dtype)
        nb = _lib.query("ecm_weights9_scratch_bytes", B, h, w)
        scratch = _scratch(nb, lr.device)
        _lib.call("ecm_context_weights_fwd", _p(lr), _p(hr), _p(W0), _p(W1), _p(W2), _p(W3), _p(out), _p(scratch),
                  C.c_longlong(nb), B, h, w, s, variant, _stream())
        ctx.save_for_backward(lr, hr, W0, W1, W2, W3, out)
        ctx.s, ctx.variant = s, variant
        return out

    @staticmethod
    def backward(ctx, g):
        lr, hr, W0, W1, W2, W3, out = ctx.saved_tensors
        B, _, h, w = lr.shape
        s, variant = ctx.s, ctx.variant
        g = _c(g)
        glr, ghr = _empty_like(lr), _empty_like(hr)
        gW = torch.empty(2112 + 512 + 128 + 8, device=lr.device, dtype=lr.dtype)
        nb = _lib.query("ecm_context_weights_bwd_scratch_bytes", B, h, w, s, variant)
        if nb == 0:
            raise RuntimeError(f"context_weights backward: unsupported scale {s} (the registered architectures' scales are 4, 8, 16)")
        scratch = _scratch(nb, lr.device)
        _lib.call("ecm_context_weights_bwd", _p(lr), _p(hr), _p(W0), _p(W1), _p(W2), _p(W3), _p(out), _p(g), _p(glr),
                  _p(ghr), _p(gW), _p(scratch), C.c_longlong(nb), B, h, w, s, variant, _stream())
        return (glr, ghr, gW[:2112].view_as(W0), gW[2112:2624].view_as(W1), gW[2624:2752].view_as(W2),
                gW[2752:].view_as(W3), None)


def context_weights(lr, hr, W0, W1, W2, W3, variant):
    if variant == 0:
        return ECMWeights9.apply(lr, hr, W0, W1, W2, W3)
    return ContextWeights.apply(lr, hr, W0, W1, W2, W3, int(variant))


class VolumeMapping(torch.autograd.Function):
    """Fused volume-mapping head (cmfsm_sub_16.py:767-801): c [NH,B,Dl,h,w], m5 [B,5,H,W], mt3 [B,3,H,W] -> [NH,B,H,W]."""

    @staticmethod
    def forward(ctx, c, m5, mt3, scale):
        _chk(c, m5, mt3)
        _need(c.dim() == 5 and scale >= 1 and c.numel() > 0, lambda: f"volume_mapping: c {tuple(c.shape)}, scale {scale}")
        c, m5, mt3 = _c(c), _c(m5), _c(mt3)
        NH, B, Dl, h, w = c.shape
        _need(tuple(m5.shape) == (B, 5, h * scale, w * scale) and tuple(mt3.shape) == (B, 3, h * scale, w * scale),
              lambda: f"volume_mapping: m5 {tuple(m5.shape)} / mt3 {tuple(mt3.shape)} for c {tuple(c.shape)} at scale {scale}: "
                      "want [B,5,h*scale,w*scale] and [B,3,h*scale,w*scale]")
        out = torch.empty(NH, B, h * scale, w * scale, device=c.device, dtype=c.dtype)
        _lib.call("ecm_volume_mapping_fwd", _p(c), C.c_longlong(B * Dl * h * w), _p(m5), _p(mt3), _p(out), NH, B, Dl, h, w,
                  scale, _stream())
        ctx.save_for_backward(c, m5, mt3)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        c, m5, mt3 = ctx.saved_tensors
        NH, B, Dl, h, w = c.shape
        g = _c(g)
        gc, gm5, gmt3 = _empty_like(c), _empty_like(m5), _empty_like(mt3)
        nb = _lib.query("ecm_volume_mapping_bwd_scratch_bytes", NH, B, Dl, h, w, ctx.scale)
        scratch = _scratch(nb, c.device)
        _lib.call("ecm_volume_mapping_bwd", _p(c), C.c_longlong(B * Dl * h * w), _p(m5), _p(mt3), _p(g), _p(gc), _p(gm5),
                  _p(gmt3), _p(scratch), C.c_longlong(nb), NH, B, Dl, h, w, ctx.scale, _stream())
        return gc, gm5, gmt3, None


def volume_mapping(c, m5, mt3, scale):
    return VolumeMapping.apply(c, m5, mt3, int(scale))


class TrilinearSoftArgmin(torch.autograd.Function):
    """Fused trilinear head (bilinear_cmf.py:447-471): c [NH,B,Dl,h,w] -> [NH,B,H,W]."""

    @staticmethod
    def forward(ctx, c, Do, H, W):
        _chk(c)
        _need(c.dim() == 5 and c.numel() > 0 and Do >= 1 and H >= 1 and W >= 1,
              lambda: f"trilinear_softargmin: c {tuple(c.shape)} -> ({Do}, {H}, {W})")
        c = _c(c)
        NH, B, Dl, h, w = c.shape
        out = torch.empty(NH, B, H, W, device=c.device, dtype=c.dtype)
        _lib.call("ecm_trilinear_softargmin_fwd", _p(c), C.c_longlong(B * Dl * h * w), _p(out), NH, B, Dl, h, w, Do, H, W,
                  _stream())
        ctx.save_for_backward(c)
        ctx.dims = (Do, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        NH, B, Dl, h, w = c.shape
        Do, H, W = ctx.dims
        g = _c(g)
        gc = _empty_like(c)
        nb = _lib.query("ecm_trilinear_softargmin_bwd_scratch_bytes", NH, B, Dl, h, w, H, W)
        scratch = _scratch(nb, c.device)
        _lib.call("ecm_trilinear_softargmin_bwd", _p(c), C.c_longlong(B * Dl * h * w), _p(g), _p(gc), _p(scratch),
                  C.c_longlong(nb), NH, B, Dl, h, w, Do, H, W, _stream())
        return gc, None, None, None


def trilinear_softargmin(c, Do, H, W):
    return TrilinearSoftArgmin.apply(c, int(Do), int(H), int(W))


# ------------------------------------------------------------------------------------ a5-a7 conv / deconv / GN
def _pack_conv(w, flip_transpose=False):
    Co, Ci = w.shape[0], w.shape[1]
    kin, kout = (Co, Ci) if flip_transpose else (Ci, Co)
    n = _lib.query("ecm_conv3d_packed_floats", kin, kout)
    packed = torch.empty(n, device=w.device, dtype=w.dtype)
    _lib.call("ecm_conv3d_pack_weight", _p(w), _p(packed), Co, Ci, int(flip_transpose), _stream())
    return packed


def _pack_deconv(w):
    """w [A,Bc,3,3,3]: ConvTranspose3d weight [Ci,Co,...] (or a Conv3d weight whose stride-2 dgrad is wanted)."""
    A, Bc = w.shape[0], w.shape[1]
    n = _lib.query("ecm_conv3d_packed_floats", A, Bc)
    packed = torch.empty(n, device=w.device, dtype=w.dtype)
    _lib.call("ecm_deconv3d_pack_weight", _p(w), _p(packed), A, Bc, _stream())
    return packed


# Stride-1 3x3(x3) convolutions -- forward and data gradient -- run on the Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip):
# 2.25x fewer matrix-core multiplies, fp32 throughout.  ECM_WINOGRAD=0 (or ops.WINOGRAD = False) selects the direct
# implicit-GEMM kernels for everything; both are parity-tested.
import os as _os
WINOGRAD = _os.environ.get("ECM_WINOGRAD", "1") != "0"
WINO2D_MIN_CI = int(_os.environ.get("ECM_WINO2D_MIN_CI", "32"))


def _wino_ok(x):
    """Winograd kernels read the patch as pairs of neighbouring columns (rows of at least two elements) and address 32
    channel planes with 32-bit byte offsets (a [D,]H,W plane set of at most 2^24 elements); anything else takes the direct
    kernels."""
    vol = x.shape[-1] * x.shape[-2] * (x.shape[-3] if x.dim() == 5 else 1)
    return WINOGRAD and x.shape[-1] >= 2 and vol * 128 <= 0x80000000
WINOGRAD_WGRAD = _os.environ.get("ECM_WINOGRAD_WGRAD", "1") != "0"


def _wino_pack(w, kd, flip_transpose):
    """w: [Co,Ci,3,3,3] (kd 3) or [Co,Ci,3,3] (kd 1), contiguous."""
    Co, Ci = w.shape[:2]
    kin, kout = (Co, Ci) if flip_transpose else (Ci, Co)

    def build():
        packed = torch.empty(_lib.query("ecm_conv_wino_packed_floats", kin, kout, kd), device=w.device, dtype=w.dtype)
        _lib.call("ecm_conv_wino_pack_weight", _p(w), _p(packed), Co, Ci, kd, int(flip_transpose), _stream())
        return packed
    return _cached_pack(w, ("wT" if flip_transpose else "w") + str(kd), build)


def _wino_pack_both(w, kd):
    """(forward layout, data-gradient layout) from ONE launch: what a training step needs of a layer -- the second one is kept on
    the autograd context for the backward of the same step (`_wino_pack_b`), which otherwise packs it in a dependent 5 us launch
    right in front of the data-gradient kernel (67 such launches per step)."""
    Co, Ci = w.shape[:2]
    nf = _lib.query("ecm_conv_wino_packed_floats", Ci, Co, kd)
    nb = _lib.query("ecm_conv_wino_packed_floats", Co, Ci, kd)
    packed = torch.empty(nf + nb, device=w.device, dtype=w.dtype)
    _lib.call("ecm_conv_wino_pack_weight2", _p(w), _p(packed), Co, Ci, kd, _stream())
    return packed[:nf], (packed[nf:], w._version, w.data_ptr())


def _wino_pack_fwd(ctx, x, w, kd, grad_mode):
    """Forward layout of w; in a pass that will need the data gradient, also the backward layout (stored on ctx).
    `grad_mode`: torch.is_grad_enabled() as the CALLER saw it -- inside autograd.Function.forward grad mode is always off, and
    needs_input_grad reflects the inputs' requires_grad flags even under no_grad, so neither can tell an eval call."""
    ctx.packed_b = None
    if grad_mode and ctx.needs_input_grad[0] and not _FROZEN_DEPTH and not torch.cuda.is_current_stream_capturing():
        pf, ctx.packed_b = _wino_pack_both(w, kd)
        return pf
    return _wino_pack(w, kd, False)


def _wino_pack_b(ctx, w, kd):
    """Data-gradient layout: the one packed in forward if the weight has not been written since, else a fresh one."""
    pb = getattr(ctx, "packed_b", None)
    if pb is not None and pb[1] == w._version and pb[2] == w.data_ptr():
        return pb[0]
    return _wino_pack(w, kd, True)


def _wino_run(x, packed, Co, kd, addend=None):
    """x: [B,Ci,D,H,W] (kd 3; or kd 1 on D independent planes) or [B,Ci,H,W] (kd 1) -> same spatial shape with Co channels."""
    B, Ci = x.shape[:2]
    D, H, W = (x.shape[2:] if x.dim() == 5 else (1,) + tuple(x.shape[2:]))
    y = torch.empty((B, Co) + tuple(x.shape[2:]), device=x.device, dtype=x.dtype)
    if addend is None:
        _lib.call("ecm_conv_wino_fwd", _p(x), _p(packed), _p(y), B, Ci, Co, D, H, W, kd, _stream())
    else:                                     # y = conv(x) + addend in the kernel's epilogue (see _fork_grad)
        if addend.shape != y.shape:
            raise RuntimeError("addend must have the shape of the convolution's output")
        _lib.call("ecm_conv_wino_fwd_add", _p(x), _p(packed), _p(_c(addend)), _p(y), B, Ci, Co, D, H, W, kd, _stream())
    return y


# ---- forks: a tensor consumed by a convolution AND by a skip connection ----------------------------------------------------
# Autograd sums the two gradients with a separate elementwise pass (3 passes over the tensor).  With `fork=True` a
# convolution Function also returns its input (as a view), to be used as the skip operand; its backward then receives the
# skip gradient next to the output gradient and adds it in the epilogue of the data-gradient kernel (Winograd path: one extra
# read instead of read + read + write; other paths: a plain add, i.e. what autograd would have done).
def _fork_out(y, x, fork):
    return (y, x.view_as(x)) if fork else y


def _fork_grad(gx, gskip):
    """data gradient + skip gradient where the kernel could not take the addend"""
    if gskip is None:
        return gx
    return gskip if gx is None else gx + gskip


class Fork(torch.autograd.Function):
    """x -> n views of x, one per consumer; backward adds the consumers' gradients in ONE pass (ecm_sum_n: n reads + 1 write)
    instead of autograd's n-1 binary adds of 3 passes each."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [_c(g) for g in grads if g is not None]
        if not gs:
            return None, None
        while len(gs) > 1:
            take, gs = gs[:4], gs[4:]
            if len(take) == 1:
                gs.append(take[0])
                continue
            out = torch.empty_like(take[0])
            take = take + [None] * (4 - len(take))
            _lib.call("ecm_sum_n", _p(take[0]), _p(take[1]), _p(take[2]), _p(take[3]), _p(out), C.c_longlong(out.numel()),
                      _stream())
            gs.insert(0, out)
        return gs[0], None


class ForkHead(torch.autograd.Function):
    """x -> (alias of x, x[:n]): one consumer of the whole batch and one of its first n samples (the encoder's full-resolution
    map: the next encoder layer takes both images, the ECM weights only the left ones, cmfsm.py:657-664).  Autograd's own
    route for that is a zero-filled full-size tensor, a copy of the slice gradient into it and a full-size add (2.8 GB of
    traffic at batch 4); here the full gradient is copied once and the slice gradient added into the copy's first n samples.
    The incoming gradient itself is never written: autograd does not promise that a node owns it (a tensor hook,
    `retain_grad`, a producer that returns one tensor for two inputs, or a `retain_graph` replay may hold the same object).
    Where the consumer of the whole batch is a GroupNorm (cmfsm's encoder) the fork lives inside that node instead
    (GroupNormAct `head`), which adds into its own fresh output with no copy at all."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        ctx.set_materialize_grads(False)
        return x.view_as(x), x[:n]

    @staticmethod
    def backward(ctx, g_full, g_head):
        if g_head is None:
            return g_full, None
        if g_full is None:
            raise RuntimeError("ForkHead: the full-batch consumer produced no gradient")   # not a configuration of this model
        out = g_full.clone(memory_format=torch.contiguous_format)
        out[:ctx.n].add_(g_head)
        return out, None


def fork_head(x, n):
    """(alias of x, x[:n]) with the slice's gradient folded into the full gradient in place (see ForkHead)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x[:n]
    return ForkHead.apply(x, int(n))


def fork(x, n):
    """n aliases of x for n consumers whose gradients are then summed by one kernel (see Fork)."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return Fork.apply(x, int(n))


def _conv_fwd(x, packed, Co, stride):
    B, Ci, D, H, W = x.shape
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(B, Co, Do, Ho, Wo, device=x.device, dtype=x.dtype)
    _lib.call("ecm_conv3d_k3_fwd", _p(x), _p(packed), _p(y), B, Ci, Co, D, H, W, stride, _stream())
    return y


def _deconv_fwd(x, packed, Co, out_dhw):
    B, Ci, D, H, W = x.shape
    Do, Ho, Wo = out_dhw
    y = torch.empty(B, Co, Do, Ho, Wo, device=x.device, dtype=x.dtype)
    _lib.call("ecm_deconv3d_k3s2_fwd", _p(x), _p(packed), _p(y), B, Ci, Co, D, H, W, Do, Ho, Wo, _stream())
    return y


# Weight gradients on a side stream.  A layer's weight gradient feeds nothing but the optimizer, while its data gradient is on
# the critical path of backward together with the HBM-bound GroupNorm backward passes: launched on a second stream, the
# matrix-bound weight-gradient kernels run under those passes instead of queueing between them (backward at batch 4:
# 86.9 -> 82.3 ms, gradients bit-identical).  Ordering: the side stream waits for the main stream's position at launch (x and
# gy are ready), the operands are recorded on it (the caching allocator must not recycle them under the kernel), and the main
# stream waits for the side stream once per backward pass, from an engine callback that runs when the pass has finished.
# That is safe exactly when nothing reads a weight gradient DURING the pass: it is therefore OFF unless the training harness
# that owns the gradients' consumers turns it on (dist.FlatBucketDDP does: it joins before it gathers) or ECM_WGRAD_OVERLAP=1
# asks for it; non-leaf weights (nn.DataParallel replicas, whose gradients flow on through Broadcast.backward), weights with
# tensor hooks and weights that already HAVE a gradient (a layer used twice in one graph, accumulation over several passes:
# AccumulateGrad then adds on the main stream at once) always stay on the main stream, and so does everything during graph
# capture.  ECM_WGRAD_OVERLAP=0: never.
_WGRAD_ENV = _os.environ.get("ECM_WGRAD_OVERLAP", "")
WGRAD_OVERLAP = _WGRAD_ENV == "1"


def enable_wgrad_overlap(on=True):
    """Called by a harness that guarantees a join (join_side_streams) before any weight gradient is read; returns the
    previous setting.  The environment's ECM_WGRAD_OVERLAP=0 wins."""
    global WGRAD_OVERLAP
    prev = WGRAD_OVERLAP
    WGRAD_OVERLAP = bool(on) and _WGRAD_ENV != "0"
    return prev


_SIDE = {}     # device index -> [side stream, join queued for the running backward pass, weights seen since the join, event behind the last join]

# Host run-ahead (round 4).  The operands of a side-stream launch are recorded on that stream, so the caching allocator hands
# their blocks out again only once the side stream's work has FINISHED on the device -- unlike main-stream blocks, which it
# recycles in stream order however far the host runs ahead.  A training loop that never synchronises (the host enqueues a step
# in 23 ms, the device runs it in 124) therefore found none of the previous steps' blocks free and went to the driver for new
# ones every step: 320 hipMalloc calls and 40.7 -> 137.5 GiB reserved over 15 steps, and a step time that depended on how fast
# the driver could hand out memory (2x slower right after another process had released its own: the driver scrubs what it
# hands out again).  So the host waits, at the start of a training forward (models: hot path) and in front of a pass's first side launch, for the
# event recorded behind the PREVIOUS pass's join: at most one step's operands are ever pending, the allocator reaches its steady
# state in the first step, and the device still has the optimiser step + the queued forward to run while the host catches up
# (no bubble: measured step time unchanged on a fresh device).  ECM_WGRAD_PACE=0 turns the wait off (A/B only).
_WGRAD_PACE = _os.environ.get("ECM_WGRAD_PACE", "1") != "0"


def pace_side_streams():
    """Block the HOST until the weight gradients of the previous backward pass have finished on the device (returns at once
    when there were none, or during stream capture).  Bounds the caching allocator's pending side-stream blocks to one pass."""
    if not _WGRAD_PACE or not _SIDE:
        return
    for dev, st in list(_SIDE.items()):
        ev = st[3]
        if ev is not None and not torch.cuda.is_current_stream_capturing():
            ev.synchronize()


def join_side_streams():
    """Make the current stream wait for every weight gradient launched so far (a no-op when there is none)."""
    for dev, st in list(_SIDE.items()):                      # (a second device's autograd thread may add its entry meanwhile)
        if st[0] is not None:
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(st[0])
            if st[1]:                                        # side work was launched since the last join: mark its end for the pacer
                if st[3] is None:
                    st[3] = torch.cuda.Event()
                st[3].record(cur)
        st[1] = False
        st[2].clear()


def _on_side(fn, w, *operands):
    dev = operands[0].device
    if (not WGRAD_OVERLAP or w is None or not w.is_leaf or w._backward_hooks or torch.cuda.is_current_stream_capturing()
            or torch.is_grad_enabled()          # backward(create_graph=True): AccumulateGrad CLONES the gradient on the main stream
            or not w.is_contiguous()):          # layout outside AccumulateGrad's stealing contract: it clones there too
        return fn()
    st = _SIDE.get(dev.index)
    if st is None:
        st = _SIDE[dev.index] = [torch.cuda.Stream(device=dev), False, set(), None]
    if w.grad is not None or w.data_ptr() in st[2]:
        # The weight already has a gradient (accumulation over several backward passes), or this pass has already produced one
        # for it (a layer used twice in one graph -- the reference calls its encoder once per image): autograd will ADD the
        # two on the main stream as soon as this node returns (AccumulateGrad, or the input buffer of the leaf's node), so
        # both operands of that addition must be complete there.
        join_side_streams()
        st[2].add(w.data_ptr())
        return fn()
    st[2].add(w.data_ptr())
    side, main = st[0], torch.cuda.current_stream(dev)
    if side == main:
        return fn()
    if not st[1]:
        pace_side_streams()               # first side launch of this pass: the previous pass's operands must have been released
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
    for t in operands:
        t.record_stream(side)
    out.record_stream(main)
    if not st[1]:
        st[1] = True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        except RuntimeError:              # not inside a backward pass (a direct call of backward()): join at once
            join_side_streams()
    return out


WGRAD_FIRST = _os.environ.get("ECM_WGRAD_FIRST", "1") == "1"


def _launch_pair(wfn, dfn):
    """Launch order of a layer's weight gradient (wfn, may go to the side stream) and data gradient (dfn, main stream).
    Weight gradient first: the side stream forks BEFORE the data gradient is queued, so the two start together.  The other
    order (ECM_WGRAD_FIRST=0: the side stream waits for the layer's own data gradient and the weight gradient starts with the
    GroupNorm backward that follows) measured 1 ms slower per backward pass (84.5 vs 83.5 ms at batch 4)."""
    if WGRAD_FIRST:
        gw = wfn() if wfn else None
        gx = dfn() if dfn else None
    else:
        gx = dfn() if dfn else None
        gw = wfn() if wfn else None
    return gx, gw


def _wino_wgrad(x, gy, Co, Ci, kd, w=None):
    """w: the weight tensor the gradient is for (decides whether the side stream may be used, see _on_side)."""
    return _on_side(lambda: _wino_wgrad_now(x, gy, Co, Ci, kd), w, x, gy)


def _wino_wgrad_now(x, gy, Co, Ci, kd):
    """Winograd-form weight gradient of a stride-1 3x3(x3) convolution: x [B,Ci,(D,)H,W], gy [B,Co,(D,)H,W]."""
    B = x.shape[0]
    D, H, W = (x.shape[2:] if x.dim() == 5 else (1,) + tuple(x.shape[2:]))
    gw = torch.empty((Co, Ci) + ((3, 3, 3) if kd == 3 else (3, 3)), device=x.device, dtype=x.dtype)
    nb = _lib.query("ecm_conv_wino_wgrad_scratch_bytes", B, Ci, Co, D, H, W, kd)
    scratch = _scratch(nb, x.device)
    _lib.call("ecm_conv_wino_wgrad", _p(x), _p(gy), _p(gw), _p(scratch), C.c_longlong(nb), B, Ci, Co, D, H, W, kd, _stream())
    return gw


def _wgrad(x, gy, Co, Ci, stride, w=None):
    """gw[Co,Ci,3,3,3] = sum gy[b,co,o] x[b,ci,o*stride+k-1]."""
    if stride == 1 and _wino_ok(x) and WINOGRAD_WGRAD:
        return _wino_wgrad(x, gy, Co, Ci, 3, w)
    return _on_side(lambda: _wgrad_direct(x, gy, Co, Ci, stride), w, x, gy)


def _wgrad_direct(x, gy, Co, Ci, stride):
    B, _, D, H, W = x.shape
    gw = torch.empty(Co, Ci, 3, 3, 3, device=x.device, dtype=x.dtype)
    nb = _lib.query("ecm_conv3d_wgrad_scratch_bytes", B, Ci, Co, D, H, W, stride)
    scratch = _scratch(nb, x.device)
    _lib.call("ecm_conv3d_k3_wgrad", _p(x), _p(gy), _p(gw), _p(scratch), C.c_longlong(nb), B, Ci, Co, D, H, W, stride,
              _stream())
    return gw


class Conv3dK3(torch.autograd.Function):
    """nn.Conv3d(k=3, pad=1, stride 1|2, bias=False) (cmfsm.py:52-57)."""

    @staticmethod
    def forward(ctx, x, w, stride, fork=False, grad_mode=True):
        _chk(x, w)
        _need(x.dim() == 5 and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[1] == x.shape[1] and stride in (1, 2)
              and x.numel() > 0 and w.shape[0] > 0,
              lambda: f"conv3d_k3: x {tuple(x.shape)}, w {tuple(w.shape)}, stride {stride}: want [B,Ci,D,H,W], [Co,Ci,3,3,3], stride 1|2")
        ctx.side_ok = w.is_contiguous()        # else the saved weight is a COPY and AccumulateGrad will re-lay the gradient out (see _on_side)
        x, w = _c(x), _c(w)
        ctx.set_materialize_grads(False)
        if _is_c1(w, stride):
            B, Ci, D, H, W = x.shape
            y = torch.empty(B, 1, D, H, W, device=x.device, dtype=x.dtype)
            _lib.call("ecm_conv3d_c1_fwd", _p(x), _p(w), _p(y), B, Ci, D, H, W, _stream())
        elif stride == 1 and _wino_ok(x):
            y = _wino_run(x, _wino_pack_fwd(ctx, x, w, 3, grad_mode), w.shape[0], 3)
        else:
            y = _conv_fwd(x, _pack_conv(w), w.shape[0], stride)
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return _fork_out(y, x, fork)

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, w = ctx.saved_tensors
        if gy is None:                         # only the forked input was used downstream
            return gskip, None, None, None, None
        gy = _c(gy)
        Co, Ci = w.shape[0], w.shape[1]
        def wfn():
            if _is_c1(w, ctx.stride):
                B, _, D, H, W = x.shape
                g = _empty_like(w)
                nb = _lib.query("ecm_conv3d_c1_wgrad_scratch_bytes", B, Ci, D, H, W)
                scratch = _scratch(nb, x.device)
                _lib.call("ecm_conv3d_c1_wgrad", _p(x), _p(gy), _p(g), _p(scratch), C.c_longlong(nb), B, Ci, D, H, W,
                          _stream())
                return g
            return _wgrad(x, gy, Co, Ci, ctx.stride, w if ctx.side_ok else None)

        def dfn():
            if ctx.stride == 1 and _wino_ok(x) and not _is_c1(w, ctx.stride):
                return _wino_run(gy, _wino_pack_b(ctx, w, 3), Ci, 3, addend=gskip)
            if _is_c1(w, ctx.stride):
                g = torch.empty(x.shape, device=x.device, dtype=x.dtype)
                _lib.call("ecm_conv3d_c1_dgrad", _p(gy), _p(w), _p(g), x.shape[0], Ci, x.shape[2], x.shape[3], x.shape[4],
                          _stream())
            elif ctx.stride == 1:
                g = _conv_fwd(gy, _pack_conv(w, True), Ci, 1)
            else:
                g = _deconv_fwd(gy, _pack_deconv(w), Ci, x.shape[2:])
            return _fork_grad(g, gskip)
        gx, gw = _launch_pair(wfn if ctx.needs_input_grad[1] else None, dfn if ctx.needs_input_grad[0] else None)
        return gx, gw, None, None, None


def _is_c1(w, stride):
    """The classifier's 32 -> 1 layer has its own kernels (conv3d_c1.hip)."""
    return w.shape[0] == 1 and stride == 1 and w.shape[1] <= 32 and w.shape[1] % 8 == 0


# Packed-weight cache.  Packing a weight is one small launch per layer and call; a training step changes every weight, so
# there is nothing to cache there, and by default every call packs afresh -- which also follows in-place updates that the
# version counter does not see (`p.data.copy_()`, `dist.broadcast(p.data)`, `m.weight.data.normal_()`), and puts the pack
# kernels INSIDE a captured HIP graph so that replays follow weight updates.  An evaluation loop over fixed weights may
# opt in with `with ops.frozen_weights(): ...`: inside it a packed layout rides on the weight tensor OBJECT, tagged with the
# context's epoch, the tensor's version counter and its storage address, and is reused until one of them moves; leaving the
# outermost context (or `invalidate_packed()`) drops every cached layout.  Keying a global table by data_ptr would be wrong:
# the caching allocator hands the same address to the next tensor of that size.
import contextlib as _contextlib

_FROZEN_DEPTH = 0
_PACK_EPOCH = 0


@_contextlib.contextmanager
def frozen_weights():
    """Promise that no weight is written inside the block by means the version counter misses (`.data` writes): packed
    weight layouts are then built once and reused (~60 fewer launches per eval forward of cmfsm)."""
    global _FROZEN_DEPTH, _PACK_EPOCH
    _FROZEN_DEPTH += 1
    try:
        yield
    finally:
        _FROZEN_DEPTH -= 1
        if _FROZEN_DEPTH == 0:
            _PACK_EPOCH += 1


def invalidate_packed():
    """Drop every cached packed layout (call after writing weights through `.data` inside a frozen_weights() block)."""
    global _PACK_EPOCH
    _PACK_EPOCH += 1


def _cached_pack(w, kind, build):
    if _FROZEN_DEPTH == 0 or torch.cuda.is_current_stream_capturing():
        return build()
    cache = getattr(w, "_ecm_packed", None)
    if cache is None:
        cache = {}
        try:
            w._ecm_packed = cache
        except AttributeError:
            return build()
    tag = (_PACK_EPOCH, w._version, w.data_ptr(), w.device)
    hit = cache.get(kind)
    if hit is not None and hit[0] == tag:
        return hit[1]
    packed = build()
    cache[kind] = (tag, packed)
    return packed


# (kh, kw, stride, dil) -> output-channel tilings (tiles of 32 per workgroup) the kernels are instantiated for; mirrors the
# COTS masks of csrc/conv2d_case_*.hip and the tiling rule c2_plan() of csrc/conv2d_kernel.h
_C2_CASES = {(3, 3, 1, 1): {1, 2, 3, 4}, (3, 3, 1, 2): {2, 4}, (3, 3, 1, 4): {4}, (3, 3, 2, 1): {1, 2, 4},
             (3, 5, 1, 1): {1, 3}, (1, 1, 1, 1): {1, 2, 4}, (1, 1, 2, 1): {1, 2, 4}}


def _cot(Co, kh, kw):
    if Co <= 32:
        return 1
    if Co <= 64:
        return 2
    return 3 if (kh * kw != 1 and Co % 96 == 0 and Co % 128 != 0) else 4


def conv2d_supported(Ci, Co, kh, kw, stride, dil):
    """Is this layer -- forward AND data gradient -- inside the native 2-D family (csrc/conv2d.hip)?"""
    cots = _C2_CASES.get((kh, kw, stride, dil))
    if cots is None or _cot(Co, kh, kw) not in cots:
        return False
    if stride == 1:
        return _cot(Ci, kh, kw) in cots                      # data gradient: same case with Cout' = Ci
    if kh == 3:                                              # stride-2 3x3: transposed-conv kernel, <= 64 output channels
        return Ci <= 64 and Co % 4 == 0
    return _cot(Ci, 1, 1) in _C2_CASES[(1, 1, 1, 1)]         # stride-2 1x1: a stride-1 1x1 on the coarse grid + zero insertion


def _pack2d(w, flip_transpose):
    w = _c(w)
    Co, Ci, kh, kw = w.shape
    kin, kout = (Co, Ci) if flip_transpose else (Ci, Co)

    def build():
        packed = torch.empty(_lib.query("ecm_conv2d_packed_floats_ex", kin, kout, kh, kw), device=w.device, dtype=w.dtype)
        _lib.call("ecm_conv2d_pack_weight_ex", _p(w), _p(packed), Co, Ci, kh, kw, int(flip_transpose), _stream())
        return packed
    return _cached_pack(w, "c2T" if flip_transpose else "c2", build)


def _conv2d_run(x, packed, Co, kh, kw, stride, dil, pad_top, pad_left, Ho, Wo):
    B, Ci, H, W = x.shape
    y = torch.empty(B, Co, Ho, Wo, device=x.device, dtype=x.dtype)
    _lib.call("ecm_conv2d_fwd_ex", _p(x), _p(packed), _p(y), B, Ci, Co, H, W, kh, kw, stride, dil, pad_top, pad_left, Ho, Wo,
              _stream())
    return y


class Conv2dG(torch.autograd.Function):
    """nn.Conv2d(bias=False) of the encoder (convbn, cmfsm.py:36-46; 1x1 projections 152-170, 231-236) and the class
    convolutions of the collapsed cost volume, on the MFMA implicit-GEMM kernels: forward, data gradient (stride 1: the
    same kernel on flipped / transposed weights; stride 2: the transposed-conv kernel) and weight gradient."""

    @staticmethod
    def forward(ctx, x, w, stride, dil, pad_top, pad_left, Ho, Wo, fork=False, grad_mode=True):
        _chk(x, w)
        _need(x.dim() == 4 and w.dim() == 4 and w.shape[1] == x.shape[1] and x.numel() > 0 and w.shape[0] > 0
              and stride >= 1 and dil >= 1 and pad_top >= 0 and pad_left >= 0 and Ho >= 1 and Wo >= 1,
              lambda: f"conv2d: x {tuple(x.shape)}, w {tuple(w.shape)}, stride {stride}, dilation {dil}, pad ({pad_top}, {pad_left}), "
                      f"out ({Ho}, {Wo}): want [B,Ci,H,W] and [Co,Ci,kh,kw]")
        x = _c(x)
        ctx.set_materialize_grads(False)
        Co, Ci, kh, kw = w.shape
        same = (kh, kw, stride, dil, pad_top, pad_left) == (3, 3, 1, 1, 1, 1) and (Ho, Wo) == tuple(x.shape[-2:])
        # Winograd where it is ahead of the direct kernel (tools/wino_time.py): from 32 channels up (32->32 at 576x960:
        # 0.56 vs 0.73 ms); below that (the 3-channel stem) the 16 frequency planes are mostly padding
        ctx.wino_f, ctx.wino_b = _wino_ok(x) and same and Ci >= WINO2D_MIN_CI, _wino_ok(x) and same and Co >= WINO2D_MIN_CI
        ctx.wino_same = same
        if ctx.wino_f:
            y = _wino_run(x, _wino_pack_fwd(ctx, x, _c(w), 1, grad_mode), Co, 1)
        else:
            y = _conv2d_run(x, _pack2d(w, False), Co, kh, kw, stride, dil, pad_top, pad_left, Ho, Wo)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, dil, pad_top, pad_left, Ho, Wo)
        return _fork_out(y, x, fork)

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, w = ctx.saved_tensors
        if gy is None:
            return (gskip,) + (None,) * 9
        stride, dil, pad_top, pad_left, Ho, Wo = ctx.cfg
        Co, Ci, kh, kw = w.shape
        B, _, H, W = x.shape
        gy = _c(gy)
        def wfn():
            if ctx.wino_same and _wino_ok(x) and WINOGRAD_WGRAD:
                return _wino_wgrad(x, gy, Co, Ci, 1, w)

            def direct():
                g = _empty_like(w)
                nb = _lib.query("ecm_conv2d_wgrad_ex_scratch_bytes", B, Ci, Co, Ho, Wo, kh, kw, stride)
                scratch = _scratch(nb, x.device)
                _lib.call("ecm_conv2d_wgrad_ex", _p(x), _p(gy), _p(g), _p(scratch), C.c_longlong(nb), B, Ci, Co, H, W, kh, kw,
                          stride, dil, pad_top, pad_left, Ho, Wo, _stream())
                return g
            return _on_side(direct, w, x, gy)

        def dfn():
            if ctx.wino_b:
                return _wino_run(gy, _wino_pack_b(ctx, _c(w), 1), Ci, 1, addend=gskip)
            if stride == 1:
                # gx[i] = sum_k w[k] gy[i + pad - k*dil]: the conv of gy with the flipped kernel, padding (K-1)*dil - pad
                g = _conv2d_run(gy, _pack2d(w, True), Ci, kh, kw, 1, dil, (kh - 1) * dil - pad_top, (kw - 1) * dil - pad_left,
                                H, W)
            elif kh == 3:
                if pad_top != 1 or pad_left != 1:
                    raise RuntimeError("stride-2 3x3 data gradient: padding 1 only")
                wc = _c(w)

                def build():
                    packed = torch.empty(9 * Co * ((Ci + 31) // 32) * 32, device=w.device, dtype=w.dtype)
                    _lib.call("ecm_deconv2d_pack_weight", _p(wc), _p(packed), Co, Ci, _stream())
                    return packed
                g = torch.empty(B, Ci, H, W, device=x.device, dtype=x.dtype)
                _lib.call("ecm_deconv2d_k3s2_fwd", _p(gy), _p(_cached_pack(wc, "d2", build)), _p(g), B, Co, Ci, Ho, Wo, H, W,
                          _stream())
            else:
                # 1x1, stride 2 (the downsample projections): W^T gy lands on the even positions, zeros elsewhere
                small = _conv2d_run(gy, _pack2d(w, True), Ci, 1, 1, 1, 1, 0, 0, Ho, Wo)
                g = torch.empty(B, Ci, H, W, device=x.device, dtype=x.dtype)
                _lib.call("ecm_zero_insert2d", _p(small), _p(g), C.c_longlong(B * Ci), H, W, Ho, Wo, _stream())
            return _fork_grad(g, gskip)
        gx, gw = _launch_pair(wfn if ctx.needs_input_grad[1] else None, dfn if ctx.needs_input_grad[0] else None)
        return gx, gw, None, None, None, None, None, None, None, None


def conv2d(x, w, stride=1, dil=1, pad_top=None, pad_left=None, Ho=None, Wo=None, fork=False):
    """General native 2-D convolution; default padding "same-style" dil*(k-1)/2 and the matching output size.
    fork=True: returns (y, x') with x' a view of x to be used as the skip operand (see _fork_out)."""
    kh, kw = w.shape[-2:]
    H, W = x.shape[-2:]
    pt = dil * (kh - 1) // 2 if pad_top is None else int(pad_top)
    pl = dil * (kw - 1) // 2 if pad_left is None else int(pad_left)
    if Ho is None:
        Ho = (H + 2 * pt - dil * (kh - 1) - 1) // stride + 1
    if Wo is None:
        Wo = (W + 2 * pl - dil * (kw - 1) - 1) // stride + 1
    return Conv2dG.apply(x, w, int(stride), int(dil), pt, pl, int(Ho), int(Wo), bool(fork), torch.is_grad_enabled())


def conv2d_k3(x, w):
    """3x3 / stride 1 / pad 1 (kept as the name the 32/64-channel encoder layers were introduced under)."""
    return conv2d(x, w, 1, 1)


# ---- dilated 3x3 layers as d*d independent phase planes ---------------------------------------------------------------------
# A dilation-d 3x3 convolution (padding d) touches only pixels of equal (y mod d, x mod d): it is d*d independent ordinary
# 3x3 / pad 1 convolutions of the phase sub-images x[..., p::d, q::d].  A layer whose convolutions all have dilation d
# (cmfsm's layer4, cmfsm.py:150: 128 -> 128, dilation 2) therefore runs on the Winograd kernels with its activations kept as
# [B,C,d*d,H/d,W/d] (GroupNorm and the residual adds see the same elements in another order), at the price of one
# re-arrangement on the way in and one on the way out.
def phase_split(x, d):
    B, Cc, H, W = x.shape
    return x.view(B, Cc, H // d, d, W // d, d).permute(0, 1, 3, 5, 2, 4).reshape(B, Cc, d * d, H // d, W // d)


def phase_merge(x, d):
    B, Cc, _, h, w = x.shape
    return x.view(B, Cc, d, d, h, w).permute(0, 1, 4, 2, 5, 3).reshape(B, Cc, h * d, w * d)


class Conv2dPlanes(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 Conv2d applied to every plane of x [B,Ci,P,h,w] (P = the phase planes of a dilated layer) on
    the Winograd kernels: forward, data gradient and weight gradient."""

    @staticmethod
    def forward(ctx, x, w, fork=False, grad_mode=True):
        _chk(x, w)
        _need(x.dim() == 5 and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and w.shape[1] == x.shape[1] and x.numel() > 0
              and w.shape[0] > 0 and _wino_ok(x),
              lambda: f"conv2d_planes: x {tuple(x.shape)}, w {tuple(w.shape)}: want [B,Ci,P,h,w] and [Co,Ci,3,3] (Winograd path)")
        ctx.side_ok = w.is_contiguous()
        x, w = _c(x), _c(w)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, w)
        return _fork_out(_wino_run(x, _wino_pack_fwd(ctx, x, w, 1, grad_mode), w.shape[0], 1), x, fork)

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, w = ctx.saved_tensors
        if gy is None:
            return gskip, None, None, None
        gy = _c(gy)
        gx, gw = _launch_pair((lambda: _wino_wgrad(x, gy, w.shape[0], w.shape[1], 1, w if ctx.side_ok else None)) if ctx.needs_input_grad[1] else None,
                              (lambda: _wino_run(gy, _wino_pack_b(ctx, w, 1), w.shape[1], 1, addend=gskip)) if ctx.needs_input_grad[0] else None)
        return gx, gw, None, None


def conv2d_planes(x, w, fork=False):
    return Conv2dPlanes.apply(x, w, bool(fork), torch.is_grad_enabled())


def conv3d_k3(x, w, stride=1, fork=False):
    """fork=True: returns (y, x') with x' a view of x to be used as the skip operand (see _fork_out)."""
    return Conv3dK3.apply(x, w, int(stride), bool(fork), torch.is_grad_enabled())


class Deconv3dK3S2(torch.autograd.Function):
    """nn.ConvTranspose3d(k=3, stride=2, pad=1, output_padding=1, bias=False) (cmfsm.py:262-268)."""

    @staticmethod
    def forward(ctx, x, w):
        _chk(x, w)
        _need(x.dim() == 5 and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[0] == x.shape[1] and x.numel() > 0
              and w.shape[1] > 0,
              lambda: f"deconv3d_k3s2: x {tuple(x.shape)}, w {tuple(w.shape)}: want [B,Ci,D,H,W] and ConvTranspose3d's [Ci,Co,3,3,3]")
        ctx.side_ok = w.is_contiguous()
        x, w = _c(x), _c(w)
        B, Ci, D, H, W = x.shape
        y = _deconv_fwd(x, _pack_deconv(w), w.shape[1], (2 * D, 2 * H, 2 * W))
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _c(gy)
        Ci, Co = w.shape[0], w.shape[1]
        # gw[ci,co,k] = sum x[ci,i] gy[co,2i+k-1]: the conv-wgrad with x:=gy (big), gy:=x, "Co":=Ci, "Ci":=Co
        # gx[ci,i] = sum_{co,k} gy[co,2i+k-1] w[ci,co,k]: a stride-2 conv with w read as Conv3d [Cout=Ci,Cin=Co]
        gx, gw = _launch_pair((lambda: _wgrad(gy, x, Ci, Co, 2, w if ctx.side_ok else None)) if ctx.needs_input_grad[1] else None,
                              (lambda: _conv_fwd(gy, _pack_conv(w), Ci, 2)) if ctx.needs_input_grad[0] else None)
        return gx, gw


def deconv3d_k3s2(x, w):
    return Deconv3dK3S2.apply(x, w)


# Exchange memory of the one-pass GroupNorm kernels (cluster slots, ticket counter, per-span counters): kept per (device,
# stream), preset once; the kernels hand it back in the preset state, so no memset precedes a launch (include/ecm_hip.h:
# ecm_gn3d_fwd_p).  Dropped -- hence preset afresh -- whenever a call fails or an asynchronous time-out is reported.
_GN_CLUSTER = {}


def _gn_cluster(B, device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    nb = _lib.query("ecm_gn3d_cluster_bytes", B)
    buf = _GN_CLUSTER.get(key)
    if buf is None or buf.numel() < nb:
        buf = torch.empty(max(int(nb), 1 << 20), dtype=torch.uint8, device=device)
        _lib.call("ecm_gn3d_cluster_preset", _p(buf), C.c_longlong(buf.numel()), _stream())
        _GN_CLUSTER[key] = buf
    return buf


def _gn_call(name, *args):
    try:
        _lib.call(name, *args)
    except RuntimeError:
        _GN_CLUSTER.clear()
        raise


def _gn_capturing():
    """During HIP-graph capture the kept exchange buffer must not be used: `torch.cuda.graph` captures on ONE
    stream for every graph, so the (device, stream) cache would hand a buffer that was allocated in -- and preset by -- the
    FIRST graph to every later capture, whose kernels would then depend on another graph's replay having run (and two graphs
    replayed concurrently would share the buffer).  A captured GroupNorm therefore takes the stateless entry points
    (ecm_gn3d_fwd / _bwd), and the library itself gives a capturing stream the two-stage kernels -- no inter-workgroup
    waits, no exchange memory -- because two graphs replayed concurrently could otherwise starve each other's clusters
    (csrc/gn3d.hip: GnControl)."""
    return torch.cuda.is_current_stream_capturing()


class GroupNormAct(torch.autograd.Function):
    """y = relu?( GroupNorm32(x)*gamma+beta (+ skip) )  (cmfsm.py:58 + ReLU/residual at 287-299, 606-613, 685-693)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, skip, relu, head=0):
        """head > 0 (a fork, cf. _fork_out): also returns x[:head] for a second consumer of the first `head` samples of the
        INPUT (the encoder's full-resolution map: the ECM weights read the left images' half, cmfsm.py:657-664); backward adds
        that consumer's gradient into the first `head` samples of gx -- a tensor this node has just produced and nobody else
        has seen, so the in-place add cannot touch memory another holder relies on."""
        _chk(x, gamma, beta, skip)
        _need(x.dim() >= 3 and x.numel() > 0 and x.shape[1] % GN_GROUPS == 0 and gamma.numel() == x.shape[1] == beta.numel()
              and (skip is None or skip.shape == x.shape) and 0 <= head <= x.shape[0],
              lambda: f"group_norm_act: x {tuple(x.shape)}, gamma {tuple(gamma.shape)}, beta {tuple(beta.shape)}, skip "
                      f"{None if skip is None else tuple(skip.shape)}, head {head}: {GN_GROUPS} groups over C channels, skip of x's shape")
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        skip = _c(skip) if skip is not None else None
        B, Cc = x.shape[:2]
        S = x.numel() // (B * Cc)
        stats = torch.empty(B, GN_GROUPS, 2, device=x.device, dtype=x.dtype)
        nb = _lib.query("ecm_gn3d_scratch_bytes", B, Cc, C.c_longlong(S))
        scratch = _scratch(nb, x.device)
        y = _empty_like(x)
        if _gn_capturing():
            _gn_call("ecm_gn3d_fwd", _p(x), _p(gamma), _p(beta), _p(skip), _p(y), _p(stats), _p(scratch), C.c_longlong(nb),
                     B, Cc, C.c_longlong(S), int(relu), C.c_float(GN_EPS), _stream())
        else:
            cl = _gn_cluster(B, x.device)
            _gn_call("ecm_gn3d_fwd_p", _p(x), _p(gamma), _p(beta), _p(skip), _p(y), _p(stats), _p(scratch), C.c_longlong(nb),
                     _p(cl), C.c_longlong(cl.numel()), B, Cc, C.c_longlong(S), int(relu), C.c_float(GN_EPS), _stream())
        # ReLU mask in backward: recomputed from x unless a skip was added (then the output y is needed)
        ctx.save_for_backward(x, stats, gamma, beta, y if (relu and skip is not None) else None)
        ctx.relu, ctx.has_skip, ctx.head = relu, skip is not None, int(head)
        if head:
            ctx.set_materialize_grads(False)
            return y, x[:head]
        return y

    @staticmethod
    def backward(ctx, gy, g_head=None):
        x, stats, gamma, beta, y = ctx.saved_tensors
        if gy is None:
            raise RuntimeError("GroupNormAct: the normalised output produced no gradient")     # not a configuration of these models
        gy = _c(gy)
        B, Cc = x.shape[:2]
        S = x.numel() // (B * Cc)
        gx = _empty_like(x)
        gskip = None
        if ctx.has_skip:                    # no mask: the skip gradient is gy itself (the same tensor: nobody writes a gradient in place)
            gskip = _empty_like(x) if ctx.relu else gy
        ggamma, gbeta = _empty_like(gamma), _empty_like(gamma)
        nb = _lib.query("ecm_gn3d_scratch_bytes", B, Cc, C.c_longlong(S))
        scratch = _scratch(nb, x.device)
        if _gn_capturing():
            _gn_call("ecm_gn3d_bwd", _p(x), _p(stats), _p(gamma), _p(beta), _p(y), _p(gy), _p(gx),
                     _p(gskip if (ctx.has_skip and ctx.relu) else None), _p(ggamma), _p(gbeta), _p(scratch),
                     C.c_longlong(nb), B, Cc, C.c_longlong(S), int(ctx.relu), _stream())
        else:
            cl = _gn_cluster(B, x.device)
            _gn_call("ecm_gn3d_bwd_p", _p(x), _p(stats), _p(gamma), _p(beta), _p(y), _p(gy), _p(gx),
                     _p(gskip if (ctx.has_skip and ctx.relu) else None), _p(ggamma), _p(gbeta), _p(scratch),
                     C.c_longlong(nb), _p(cl), C.c_longlong(cl.numel()), B, Cc, C.c_longlong(S), int(ctx.relu), _stream())
        if g_head is not None:
            gx[:ctx.head].add_(g_head)          # gx is this node's own fresh output (see forward)
        return gx, ggamma, gbeta, gskip, None, None


class ClassifierTail(torch.autograd.Function):
    """GroupNorm(32) + ReLU + Conv3d(32 -> 1, k3, pad 1) of a classifier (cmfsm.py:621-634: classifN[0][1], classifN[1],
    classifN[2]) with the normalised tensor never materialised: x [B,32,D,H,W] is the RAW output of classifN[0][0]; one read
    pass gives the group statistics (ecm_gn3d_stats) and the 32 -> 1 kernels normalise + rectify while they stage x
    (ecm_conv3d_c1_gn_fwd / _gn_wgrad).  Saves the GroupNorm kernel's write and the convolution's read of the 849 MB tensor
    (batch 4) per head, and the tensor itself.  Backward: gh = dgrad(gy) (w.r.t. the normalised tensor), the GroupNorm backward
    with the ReLU mask recomputed from x."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w):
        _chk(x, gamma, beta, w)
        _need(x.dim() == 5 and x.numel() > 0 and gamma.numel() == 32 == beta.numel(),
              lambda: f"classifier_tail: x {tuple(x.shape)}, gamma {tuple(gamma.shape)}, beta {tuple(beta.shape)}")
        x, gamma, beta, w = _c(x), _c(gamma), _c(beta), _c(w)
        B, Cc, D, H, W = x.shape
        if Cc != 32 or tuple(w.shape) != (1, 32, 3, 3, 3):
            raise RuntimeError(f"classifier_tail: x {tuple(x.shape)}, w {tuple(w.shape)}: the fused tail exists for 32 -> 1 only")
        S = D * H * W
        stats = torch.empty(B, GN_GROUPS, 2, device=x.device, dtype=x.dtype)
        nb = _lib.query("ecm_gn3d_scratch_bytes", B, Cc, C.c_longlong(S))
        scratch = _scratch(nb, x.device)
        _gn_call("ecm_gn3d_stats", _p(x), _p(stats), _p(scratch), C.c_longlong(nb), B, Cc, C.c_longlong(S), C.c_float(GN_EPS), _stream())
        y = torch.empty(B, 1, D, H, W, device=x.device, dtype=x.dtype)
        _lib.call("ecm_conv3d_c1_gn_fwd", _p(x), _p(stats), _p(gamma), _p(beta), _p(w), _p(y), B, Cc, D, H, W, _stream())
        ctx.save_for_backward(x, stats, gamma, beta, w)
        ctx.side_ok = True
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, gamma, beta, w = ctx.saved_tensors
        gy = _c(gy)
        B, Cc, D, H, W = x.shape
        S = D * H * W

        def wfn():
            def now():
                g = _empty_like(w)
                nbw = _lib.query("ecm_conv3d_c1_wgrad_scratch_bytes", B, Cc, D, H, W)
                sc = _scratch(nbw, x.device)
                _lib.call("ecm_conv3d_c1_gn_wgrad", _p(x), _p(stats), _p(gamma), _p(beta), _p(gy), _p(g), _p(sc), C.c_longlong(nbw),
                          B, Cc, D, H, W, _stream())
                return g
            return _on_side(now, w, x, gy, stats, gamma, beta)

        def dfn():
            gh = torch.empty(x.shape, device=x.device, dtype=x.dtype)
            _lib.call("ecm_conv3d_c1_dgrad", _p(gy), _p(w), _p(gh), B, Cc, D, H, W, _stream())
            return gh
        gh, gw = _launch_pair(wfn if ctx.needs_input_grad[3] else None, dfn)
        gx = _empty_like(x)
        ggamma, gbeta = _empty_like(gamma), _empty_like(gamma)
        nb = _lib.query("ecm_gn3d_scratch_bytes", B, Cc, C.c_longlong(S))
        scratch = _scratch(nb, x.device)
        if _gn_capturing():
            _gn_call("ecm_gn3d_bwd", _p(x), _p(stats), _p(gamma), _p(beta), _p(None), _p(gh), _p(gx), _p(None), _p(ggamma),
                     _p(gbeta), _p(scratch), C.c_longlong(nb), B, Cc, C.c_longlong(S), 1, _stream())
        else:
            cl = _gn_cluster(B, x.device)
            _gn_call("ecm_gn3d_bwd_p", _p(x), _p(stats), _p(gamma), _p(beta), _p(None), _p(gh), _p(gx), _p(None), _p(ggamma),
                     _p(gbeta), _p(scratch), C.c_longlong(nb), _p(cl), C.c_longlong(cl.numel()), B, Cc, C.c_longlong(S), 1, _stream())
        return gx, ggamma, gbeta, gw


def classifier_tail(x, gamma, beta, w):
    """relu(GroupNorm32(x)) -> Conv3d(32 -> 1): see ClassifierTail."""
    return ClassifierTail.apply(x, gamma, beta, w)


FLYING3D_MEAN, FLYING3D_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # cmf/loader/Flying3d.py:26-27


def frame_prep(frames, crop_y0, crop_x0, th, tw, split=None, tail=0, mean=FLYING3D_MEAN, std=FLYING3D_STD, want_image=False):
    """Batched GPU form of Flying3d.__getitem__ + transform (cmf/loader/Flying3d.py:49-99): frames [B,H,W,7] float32
    (resident) -> left, right [B,3,th,tw] normalised, disparity [B,th,tw] (and the raw left image in CHW).
    Train: th, tw = 256, 512 and crop_y0/crop_x0 = the window origins the loader draws at random.  Eval on 540x960 frames:
    th, tw = 576, 960, crops 0, split=540, tail=36 (the last 36 rows are appended again).
    `frames` may also be a packed shard: a tuple (rgb6 uint8 [B,H,W,6], disparity fp16|fp32 [B,H,W])."""
    return _frame_prep(frames, crop_y0, crop_x0, th, tw, split, tail, mean, std, want_image, 0)


def frame_prep_kitti_eval(frames, th=384, tw=1248, mean=FLYING3D_MEAN, std=FLYING3D_STD, want_image=False):
    """KITTI.__getitem__'s eval branch (cmf/loader/KITTI.py:98-108): top / left repeat padding to th x tw with the
    disparity zeroed in (and, through the loader's numpy views, next to) the padding.  `frames`: float32 [B,H,W,7] or a
    packed shard tuple as in frame_prep."""
    return _frame_prep(frames, None, None, th, tw, None, 0, mean, std, want_image, 1)


def _frame_prep(frames, crop_y0, crop_x0, th, tw, split, tail, mean, std, want_image, mode):
    packed = isinstance(frames, (tuple, list))
    if packed:
        rgb, dsp = frames
        if not (rgb.is_cuda and dsp.is_cuda):
            raise RuntimeError("ecm ops run only on the MI355X HIP path: got a CPU tensor (no CPU fallback exists)")
        if rgb.dtype != torch.uint8 or dsp.dtype not in (torch.float16, torch.float32) or rgb.shape[-1] != 6 \
                or tuple(rgb.shape[:3]) != tuple(dsp.shape):
            raise RuntimeError(f"packed shard must be (uint8 [B,H,W,6], fp16|fp32 [B,H,W]), got {rgb.dtype} {tuple(rgb.shape)} / "
                               f"{dsp.dtype} {tuple(dsp.shape)}")
        rgb, dsp = rgb.contiguous(), dsp.contiguous()
        B, H, W = dsp.shape
        dev = rgb.device
    else:
        _chk(frames)
        frames = _c(frames)
        B, H, W, ch = frames.shape
        if ch != 7:
            raise RuntimeError(f"frame_prep expects [B,H,W,7] frames (left RGB, right RGB, disparity), got {tuple(frames.shape)}")
        dev = frames.device
    split = th if split is None else int(split)
    ys = (C.c_int * B)(*[int(v) for v in crop_y0]) if crop_y0 is not None else None
    xs = (C.c_int * B)(*[int(v) for v in crop_x0]) if crop_x0 is not None else None
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    left = torch.empty(B, 3, th, tw, device=dev, dtype=torch.float32)
    right = torch.empty_like(left)
    disp = torch.empty(B, th, tw, device=dev, dtype=torch.float32)
    image = torch.empty_like(left) if want_image else None
    if packed:
        _lib.call("ecm_frame_prep_packed", _p(rgb), _p(dsp), int(dsp.dtype == torch.float16), _p(left), _p(right), _p(disp),
                  _p(image), B, H, W, ys, xs, int(th), int(tw), split, int(tail), m3, s3, mode, _stream())
    elif mode == 1:
        _lib.call("ecm_frame_prep_kitti_eval", _p(frames), _p(left), _p(right), _p(disp), _p(image), B, H, W, int(th), int(tw),
                  m3, s3, _stream())
    else:
        _lib.call("ecm_frame_prep", _p(frames), _p(left), _p(right), _p(disp), _p(image), B, H, W, ys, xs, int(th), int(tw),
                  split, int(tail), m3, s3, _stream())
    return (left, right, disp, image) if want_image else (left, right, disp)


def eval_epe(pred, gt, crop_h=540, crop_w=960, maxdisp=192):
    """SceneFlow evaluation of test.py:69-94 on the device: pred [B,1,Hp,Wp] or [B,Hp,Wp] (output3), gt [B,Hg,Wg];
    -> float32[6] = (epe, epe_non, epe_true, n, n_non, n_true), both tensors cropped to [:crop_h, :crop_w]."""
    _chk(pred, gt)
    if pred.dim() == 4:
        pred = pred.squeeze(1)
    pred, gt = _c(pred), _c(gt)
    B, Hp, Wp = pred.shape
    Bg, Hg, Wg = gt.shape
    if B != Bg:
        raise RuntimeError(f"prediction / ground-truth batch sizes differ: {B} vs {Bg}")
    out = torch.empty(6, device=pred.device, dtype=torch.float32)
    n = B * int(crop_h) * int(crop_w)
    nb = _lib.query("ecm_eval_epe_scratch_bytes", C.c_longlong(n))
    scratch = _scratch(nb, pred.device)
    _lib.call("ecm_eval_epe", _p(pred), _p(gt), _p(out), _p(scratch), C.c_longlong(nb), B, Hp, Wp, Hg, Wg, int(crop_h),
              int(crop_w), C.c_float(maxdisp), _stream())
    return out


def disparity_to_uint16(pred, h, w, scale=256.0):
    """KITTI submission image of test_kitti.py:163-168: pred [B,1,Hp,Wp] or [B,Hp,Wp] (output3) -> uint16 [B,max h,max w]
    with out[b,:h[b],:w[b]] = (pred[b,-h[b]:,-w[b]:] * scale).astype(uint16) (numpy cast semantics), zeros elsewhere.
    h, w: per-sample original sizes (ints or sequences)."""
    _chk(pred)
    if pred.dim() == 4:
        pred = pred.squeeze(1)
    pred = _c(pred)
    B, Hp, Wp = pred.shape
    hs = [int(h)] * B if isinstance(h, int) else [int(v) for v in h]
    ws = [int(w)] * B if isinstance(w, int) else [int(v) for v in w]
    Ho, Wo = max(hs), max(ws)
    out = torch.empty(B, Ho, Wo, device=pred.device, dtype=torch.uint16)
    _lib.call("ecm_disp_to_u16", _p(pred), _p(out), B, Hp, Wp, (C.c_int * B)(*hs), (C.c_int * B)(*ws), Ho, Wo,
              C.c_float(scale), _stream())
    return out


class StereoLoss3(torch.autograd.Function):
    """Loss of the training scripts and the KITTI validation metrics in one pass (train.py:162,172-174;
    train_kitti.py:205-216).  Returns (loss, metrics[8]); metrics = [loss, #mask, epe, err3 %, m1, m2, m3, 0] (no grad)."""

    @staticmethod
    def forward(ctx, p1, p2, p3, gt, maxdisp, w1, w2, w3):
        _chk(p1, p2, p3, gt)
        if not (p1.numel() == p2.numel() == p3.numel() == gt.numel()):
            raise RuntimeError(f"prediction / ground-truth sizes differ: {tuple(p1.shape)} {tuple(p2.shape)} "
                               f"{tuple(p3.shape)} vs {tuple(gt.shape)}")
        p1, p2, p3, gt = _c(p1), _c(p2), _c(p3), _c(gt)
        n = gt.numel()
        out = torch.empty(8, device=gt.device, dtype=gt.dtype)
        nb = _lib.query("ecm_stereo_loss_scratch_bytes", C.c_longlong(n))
        scratch = _scratch(nb, gt.device)
        _lib.call("ecm_stereo_loss_fwd", _p(p1), _p(p2), _p(p3), _p(gt), _p(out), _p(scratch), C.c_longlong(nb),
                  C.c_longlong(n), C.c_float(maxdisp), C.c_float(w1), C.c_float(w2), C.c_float(w3), _stream())
        ctx.save_for_backward(p1, p2, p3, gt, out)
        ctx.cfg = (float(maxdisp), float(w1), float(w2), float(w3))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gmetrics):
        p1, p2, p3, gt, out = ctx.saved_tensors
        maxdisp, w1, w2, w3 = ctx.cfg
        gloss = _c(gloss.reshape(1).to(gt.dtype))
        g1, g2, g3 = _empty_like(p1), _empty_like(p2), _empty_like(p3)
        _lib.call("ecm_stereo_loss_bwd", _p(p1), _p(p2), _p(p3), _p(gt), _p(out), _p(gloss), _p(g1), _p(g2), _p(g3),
                  C.c_longlong(gt.numel()), C.c_float(maxdisp), C.c_float(w1), C.c_float(w2), C.c_float(w3), _stream())
        return g1, g2, g3, None, None, None, None, None


def stereo_loss3(preds, gt, maxdisp=192, weights=(0.5, 0.7, 1.0)):
    """-> (loss, metrics[8]) for the three predictions of a model (each [B,1,H,W] or [B,H,W]) and gt [B,H,W]."""
    p1, p2, p3 = preds
    return StereoLoss3.apply(p1, p2, p3, gt, float(maxdisp), *map(float, weights))


def group_norm_act(x, gamma, beta, skip=None, relu=False, head=0):
    """head > 0: returns (y, x[:head]) -- see GroupNormAct.forward."""
    _need(0 <= int(head) <= x.shape[0], lambda: f"group_norm_act: head {head} of a batch of {x.shape[0]}")
    if head and not (torch.is_grad_enabled() and x.requires_grad):
        return GroupNormAct.apply(x, gamma, beta, skip, bool(relu), 0), x[:head]
    return GroupNormAct.apply(x, gamma, beta, skip, bool(relu), int(head))


def gn_cluster_mode(mode=-1):
    """Process-wide switch of the one-pass GroupNorm kernels (include/ecm_hip.h: ecm_gn3d_cluster_mode): 1 = cluster kernels
    (default), 0 = two-stage kernels only -- the choice when several processes share one device; returns the previous mode."""
    return _lib.query("ecm_gn3d_cluster_mode", int(mode))


def check_async_errors(clear=True):
    """Synchronise the current device and raise if a cluster kernel's bounded wait expired since the last check (its
    outputs are NaN).  Call once per step (bench.py and the tests do); every later GroupNorm call raises on its own too."""
    torch.cuda.synchronize()
    rc = _lib.query("ecm_async_status", 1 if clear else 0)
    if rc != 0:
        _GN_CLUSTER.clear()                    # a timed-out cluster leaves the exchange memory in an unknown state
        msg = _lib.load().ecm_error_string(rc)
        raise RuntimeError(f"asynchronous device-side failure ({rc}): {msg.decode() if msg else '?'}")
