#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): per-stage fp32 error of the HIP path against the oracle evaluated in fp64 on the CPU,
next to the error of the oracle's own fp32 evaluation on the SAME inputs -- every stage is fed the fp64 truth of its input
rounded to fp32, so upstream error does not accumulate.  Usage (GPU box):  python tests/diag_stage_error.py [--wino 0|1]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecm_amd  # noqa: E402
from oracle import ecm_oracle as O  # noqa: E402
from oracle.weights import seeded, tensor_for  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--wino", type=int, default=1)
ap.add_argument("--h", type=int, default=256)
ap.add_argument("--w", type=int, default=512)
args = ap.parse_args()
ops = ecm_amd.ops
ops.WINOGRAD = bool(args.wino)
torch.set_num_threads(16)

model = ecm_amd.get_model("cmfsm")
sd32 = {k: tensor_for(k, v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(sd32)
model = model.cuda().eval()
sd64 = {k: v.double() for k, v in sd32.items()}
left, right = seeded("g8.left", 1, 3, args.h, args.w), seeded("g8.right", 1, 3, args.h, args.w)


def rel(a, t):
    a, t = a.double().cpu(), t.double()
    return float((a - t).pow(2).mean().sqrt() / t.pow(2).mean().sqrt()), float((a - t).abs().max() / t.abs().max())


def row(name, hip, o32, t64):
    rh, mh = rel(hip, t64)
    ro, mo = rel(o32, t64)
    print(f"{name:34s} hip rms {rh:.2e} max {mh:.2e} | oracle-fp32 rms {ro:.2e} max {mo:.2e} | ratio {rh / max(ro, 1e-30):5.1f}", flush=True)


with torch.no_grad():
    # 1 encoder
    lr64, _, hr64 = O.feature_extraction(left.double(), sd64)
    lrr64, _, _ = O.feature_extraction(right.double(), sd64)
    lr32, _, hr32 = O.feature_extraction(left, sd32)
    lrh, _, hrh = model.feature_extraction(left.cuda())
    row("encoder lr feature", lrh, lr32, lr64)
    row("encoder hr feature", hrh, hr32, hr64)
    # encoder pieces: firstconv only
    y64 = O._convbn2d(left.double(), sd64, "feature_extraction.firstconv.0", 1, 1, 1)
    y32 = O._convbn2d(left, sd32, "feature_extraction.firstconv.0", 1, 1, 1)
    fc = model.feature_extraction.firstconv
    yh = fc[0][1].fused(fc[0][0](left.cuda()), None, False)
    row("  firstconv.0 conv+GN (3->32)", yh, y32, y64)
    x64 = F.relu(y64)
    z64 = O._convbn2d(x64, sd64, "feature_extraction.firstconv.2", 1, 1, 1)
    z32 = O._convbn2d(x64.float(), sd32, "feature_extraction.firstconv.2", 1, 1, 1)
    zh = fc[2][1].fused(fc[2][0](x64.float().cuda()), None, False)
    row("  firstconv.2 conv+GN (32->32)", zh, z32, z64)
    c64 = F.conv2d(x64, sd64["feature_extraction.firstconv.2.0.weight"], None, 1, 1)
    c32 = F.conv2d(x64.float(), sd32["feature_extraction.firstconv.2.0.weight"], None, 1, 1)
    ch = fc[2][0](x64.float().cuda())
    row("  firstconv.2 conv only", ch, c32, c64)
    gh = fc[2][1].fused(c64.float().cuda(), None, False)
    g32 = O._gn(c64.float(), sd32, "feature_extraction.firstconv.2.1")
    row("  firstconv.2 GN only", gh, g32, z64)
    # 2 ECM weights
    w64 = O.ecm_weights_eight(lr64, hr64, sd64)
    w32 = O.ecm_weights_eight(lr64.float(), hr64.float(), sd32)
    wh = model.mapping_matrix.weights(lr64.float().cuda(), hr64.float().cuda())
    row("ecm weights (9 planes)", wh, w32, w64)
    # 3 dres0 on the cost volume
    cost64 = O.cost_volume(lr64, lrr64, 48)
    d0_64 = O.dres0(cost64, sd64)
    d0_32 = O.dres0(cost64.float(), sd32)
    mdl = sys.modules["explicit-context-mapping-for-stereo-matching_amd.models"]
    a = mdl._costvol_dres0(model.dres0, lr64.float().cuda(), lrr64.float().cuda(), 48)
    d0h = mdl._cbn(model.dres0[2], a, relu=True)
    row("dres0 (collapsed costvol conv)", d0h, d0_32, d0_64)
    mdl.EXPLICIT_COST_VOLUME = True
    a = mdl._costvol_dres0(model.dres0, lr64.float().cuda(), lrr64.float().cuda(), 48)
    mdl.EXPLICIT_COST_VOLUME = False
    row("dres0 (explicit volume)", mdl._cbn(model.dres0[2], a, relu=True), d0_32, d0_64)
    # 4 dres1
    d1_64 = O.dres1(d0_64, sd64) + d0_64
    d1_32 = O.dres1(d0_64.float(), sd32) + d0_64.float()
    xin = d0_64.float().cuda()
    y = mdl._cbn(model.dres1[0], xin, relu=True)
    d1h = mdl._cbn(model.dres1[2], y, skip=xin)
    row("dres1 (+skip)", d1h, d1_32, d1_64)
    # 5 hourglass
    h64 = O.hourglass(d1_64, None, None, sd64, "dres2")
    h32 = O.hourglass(d1_64.float(), None, None, sd32, "dres2")
    hh = model.dres2(d1_64.float().cuda(), None, None)
    for nm, i in (("out", 0), ("pre", 1), ("post", 2)):
        row(f"hourglass dres2 {nm}", hh[i], h32[i], h64[i])
    # 6 classifier
    o1_64 = h64[0] + d1_64
    c64 = O.classif(o1_64, sd64, "classif1")
    c32 = O.classif(o1_64.float(), sd32, "classif1")
    hcl = mdl._cbn(model.classif1[0], o1_64.float().cuda(), relu=True)
    chh = model.classif1[2](hcl)
    row("classif1 (32->32 + 32->1)", chh, c32, c64)
    # 7 heads
    c = c64.squeeze(1)
    p64 = O.ecm_aggregate_eight(O.soft_argmin(c), w64, 4)
    p32 = O.ecm_aggregate_eight(O.soft_argmin(c.float()), w64.float(), 4)
    disp = ops.softargmin_heads(c.float().cuda().unsqueeze(0))
    ph = ops.ecm_aggregate9(disp, w64.float().cuda(), 4)[0].unsqueeze(1)
    print(f"{'soft-argmin + aggregation':34s} hip max |err| {float((ph.double().cpu() - p64).abs().max()):.2e} px mean {float((ph.double().cpu() - p64).abs().mean()):.2e} | "
          f"oracle-fp32 max {float((p32.double() - p64).abs().max()):.2e} mean {float((p32.double() - p64).abs().mean()):.2e}")
    # whole model
    full64 = O.hot_path(lr64, hr64, lrr64, sd64)
    out = model(left.cuda(), right.cuda())
    full32 = O.cmfsm_forward(left, right, sd32)
    for i in range(3):
        d = (out[i].double().cpu() - full64[i]).abs()
        d32 = (full32[i].double() - full64[i]).abs()
        print(f"whole model pred{i + 1}: hip max {float(d.max()):.2e} mean {float(d.mean()):.2e} px | oracle-fp32 max {float(d32.max()):.2e} mean {float(d32.mean()):.2e}")
