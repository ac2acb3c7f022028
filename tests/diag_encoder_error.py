#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): per-stage fp32 error of the ENCODER on the HIP path vs the oracle in fp64 (each stage
fed the fp64 truth of its input).  Usage (GPU box): python tests/diag_encoder_error.py [--wino 0|1]"""
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
args = ap.parse_args()
ops = ecm_amd.ops
ops.WINOGRAD = bool(args.wino)
torch.set_num_threads(16)
mdl = sys.modules["explicit-context-mapping-for-stereo-matching_amd.models"]

model = ecm_amd.get_model("cmfsm")
sd32 = {k: tensor_for(k, v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(sd32)
model = model.cuda().eval()
sd64 = {k: v.double() for k, v in sd32.items()}
fe = model.feature_extraction
k = "feature_extraction"
left = seeded("g8.left", 1, 3, 256, 512)


def rel(a, t):
    a, t = a.double().cpu(), t.double()
    return float((a - t).pow(2).mean().sqrt() / t.pow(2).mean().sqrt())


def row(name, hip, o32, t64):
    rh, ro = rel(hip, t64), rel(o32, t64)
    print(f"{name:44s} hip rms {rh:.2e} | oracle-fp32 rms {ro:.2e} | ratio {rh / max(ro, 1e-30):6.1f}", flush=True)


def c(t):
    return t.float().cuda()


with torch.no_grad():
    x = left.double()
    y = F.relu(O._convbn2d(x, sd64, k + ".firstconv.0", 1, 1, 1))
    y = F.relu(O._convbn2d(y, sd64, k + ".firstconv.2", 1, 1, 1))
    y = F.relu(O._convbn2d(y, sd64, k + ".firstconv.4", 1, 1, 1))
    out_all = F.conv2d(y, sd64[k + ".firstconv.6.weight"], None, 1, 1)
    # secondconv
    def second(inp, sd):
        z = F.relu(O._gn(inp, sd, k + ".secondconv.0"))
        z = F.relu(O._convbn2d(z, sd, k + ".secondconv.2", 2, 1, 1))
        return F.relu(O._convbn2d(z, sd, k + ".secondconv.4", 1, 1, 1))
    s64, s32 = second(out_all, sd64), second(out_all.float(), sd32)
    row("secondconv (GN, s2 conv, conv)", mdl._seq_fused(fe.secondconv, c(out_all)), s32, s64)
    z = F.relu(O._gn(out_all, sd64, k + ".secondconv.0"))
    t64 = O._convbn2d(z, sd64, k + ".secondconv.2", 2, 1, 1)
    t32 = O._convbn2d(z.float(), sd32, k + ".secondconv.2", 2, 1, 1)
    m = fe.secondconv[2]
    row("  stride-2 conv + GN 32->32", m[1].fused(m[0](c(z)), None, False), t32, t64)
    l1_64, l1_32 = O._layer(s64, sd64, k + ".layer1", 3, 1, 1, 1), O._layer(s64.float(), sd32, k + ".layer1", 3, 1, 1, 1)
    row("layer1 (3 blocks 32ch)", fe.layer1(c(s64)), l1_32, l1_64)
    l2_64, l2_32 = O._layer(l1_64, sd64, k + ".layer2", 16, 2, 1, 1), O._layer(l1_64.float(), sd32, k + ".layer2", 16, 2, 1, 1)
    row("layer2 (16 blocks 64ch, s2)", fe.layer2(c(l1_64)), l2_32, l2_64)
    b64 = O._basic_block(l1_64, sd64, k + ".layer2.0", 2, 1, 1)
    b32 = O._basic_block(l1_64.float(), sd32, k + ".layer2.0", 2, 1, 1)
    row("  layer2.0 (s2 conv, conv, 1x1 s2 downsample)", fe.layer2[0](c(l1_64)), b32, b64)
    b64b = O._basic_block(b64, sd64, k + ".layer2.1", 1, 1, 1)
    b32b = O._basic_block(b64.float(), sd32, k + ".layer2.1", 1, 1, 1)
    row("  layer2.1 (two 64->64 convs)", fe.layer2[1](c(b64)), b32b, b64b)
    cc64 = F.conv2d(b64, sd64[k + ".layer2.1.conv1.0.0.weight"], None, 1, 1)
    cc32 = F.conv2d(b64.float(), sd32[k + ".layer2.1.conv1.0.0.weight"], None, 1, 1)
    row("    64->64 conv only", fe.layer2[1].conv1[0][0](c(b64)), cc32, cc64)
    gg64 = O._gn(cc64, sd64, k + ".layer2.1.conv1.0.1")
    gg32 = O._gn(cc64.float(), sd32, k + ".layer2.1.conv1.0.1")
    row("    GN(32 groups of 2 ch) only", fe.layer2[1].conv1[0][1].fused(c(cc64), None, False), gg32, gg64)
    l3_64, l3_32 = O._layer(l2_64, sd64, k + ".layer3", 3, 1, 1, 1), O._layer(l2_64.float(), sd32, k + ".layer3", 3, 1, 1, 1)
    row("layer3 (3 blocks 128ch)", fe._run_layer(fe.layer3, c(l2_64)), l3_32, l3_64)
    l4_64, l4_32 = O._layer(l3_64, sd64, k + ".layer4", 3, 1, 1, 2), O._layer(l3_64.float(), sd32, k + ".layer4", 3, 1, 1, 2)
    row("layer4 (3 blocks 128ch, dilation 2)", fe._run_layer(fe.layer4, c(l3_64)), l4_32, l4_64)
    for name, pool, idx in (("branch1", 64, 1), ("branch4", 8, 4)):
        def br(inp, sd):
            b = F.avg_pool2d(inp, pool, pool)
            b = F.relu(O._convbn2d(b, sd, f"{k}.{name}.1", 1, 0, 1))
            return F.interpolate(b, inp.shape[-2:], mode="bilinear", align_corners=False)
        pooled = mdl._pyramid_pools(c(l4_64), [getattr(fe, f"branch{i}")[0] for i in (1, 2, 3, 4)])[idx - 1]
        hb = mdl.bilinear_upsample(mdl._seq_fused(getattr(fe, name), pooled, start=1), l4_64.shape[-2:])
        row(f"{name} (pool {pool}, 1x1+GN, upsample)", hb, br(l4_64.float(), sd32), br(l4_64, sd64))
        b64_ = F.avg_pool2d(l4_64, pool, pool)
        row(f"  {name} pooling only", pooled, F.avg_pool2d(l4_64.float(), pool, pool), b64_)
        g64_ = O._convbn2d(b64_, sd64, f"{k}.{name}.1", 1, 0, 1)
        g32_ = O._convbn2d(b64_.float(), sd32, f"{k}.{name}.1", 1, 0, 1)
        row(f"  {name} 1x1 conv + GN on the pooled map {tuple(b64_.shape[-2:])}", mdl._seq_fused(getattr(fe, name), c(b64_), start=1).relu(), g32_.relu(), g64_.relu())
    lr64, _, _ = O.feature_extraction(left.double(), sd64)
    lr32, _, _ = O.feature_extraction(left, sd32)
    row("whole encoder: lr feature", fe(left.cuda())[0], lr32, lr64)
