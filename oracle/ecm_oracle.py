"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.

A functional PyTorch-CPU fp32 restatement of the reference's hot path
(cost volume -> 3-D aggregation -> soft-argmin -> explicit context mapping)
for `cmf/models/cmfsm.py` (reference file:line cited per function).  It
follows the reference's op sequence (python loop over disparities, the
NN-upsample + concat + 1x1-conv MLP per neighbour, softmax, nine shifted
accumulations) so that timing it on host cores is a fair "port" CPU baseline.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module.  The shipped package never does.

Parity pinning: every function here is checked against outputs of the
reference's own code (imported in the build container by
`tests/golden/make_golden.py` / `tests/golden/make_golden_archs.py`) stored under
`tests/golden/*.npz`; see `tests/test_oracle_golden.py`.  A second,
ATen-free restatement in plain C (`oracle/ecm_oracle_c.c`, naive loops) is
checked against the same fixtures in `tests/test_oracle_c.py`.

All functions take a plain `dict[str, Tensor]` state dict whose keys equal the
reference `cmfsm().state_dict()` keys.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

GN_GROUPS = 32          # cmfsm.py:33 group_norm_group_num
GN_EPS = 1e-5           # nn.GroupNorm default (cmfsm.py:46,58)
LEAKY = 0.01            # nn.LeakyReLU default slope (cmfsm.py:310-316)
PAD_LOGIT = -100.0      # cmfsm.py:451-452


# ----------------------------------------------------------------------------
# a1: cost volume  (cmfsm.py:667-682)
# ----------------------------------------------------------------------------
def cost_volume(ref_fea: torch.Tensor, tgt_fea: torch.Tensor, ndisp: int) -> torch.Tensor:
    """cost[b, :C, d, y, x] = L[b,:,y,x], cost[b, C:, d, y, x] = R[b,:,y,x-d] for x >= d, else 0."""
    B, C, h, w = ref_fea.shape
    cost = torch.zeros(B, 2 * C, ndisp, h, w, dtype=ref_fea.dtype)
    for d in range(ndisp):              # the reference's python loop, cmfsm.py:673
        if d == 0:
            cost[:, :C, 0] = ref_fea
            cost[:, C:, 0] = tgt_fea
        elif d < w:
            cost[:, :C, d, :, d:] = ref_fea[..., d:]
            cost[:, C:, d, :, d:] = tgt_fea[..., :-d]
    return cost.contiguous()


def matchshifted(left: torch.Tensor, right: torch.Tensor, shift: int) -> torch.Tensor:
    """One disparity slice [B,2C,1,H,W] (cmfsm.py:88-108); a second oracle for `cost_volume`."""
    B, C, H, W = left.shape
    sl = F.pad(left[..., shift:], (shift, 0, 0, 0))
    sr = F.pad(right[..., :W - shift], (shift, 0, 0, 0))
    return torch.cat((sl, sr), 1).view(B, 2 * C, 1, H, W)


# ----------------------------------------------------------------------------
# a5-a7: 3-D aggregation stack
# ----------------------------------------------------------------------------
def _gn(x, sd, key):
    return F.group_norm(x, GN_GROUPS, sd[key + ".weight"], sd[key + ".bias"], GN_EPS)


def convbn_3d(x, sd, key, stride=1):
    """Conv3d(k3,pad1,no bias) + GroupNorm(32)  (cmfsm.py:49-58). `key` names the Sequential."""
    x = F.conv3d(x, sd[key + ".0.weight"], None, stride, 1)
    return _gn(x, sd, key + ".1")


def deconvbn_3d(x, sd, key):
    """ConvTranspose3d(k3,s2,p1,op1,no bias) + GroupNorm(32)  (cmfsm.py:261-281)."""
    x = F.conv_transpose3d(x, sd[key + ".0.weight"], None, 2, 1, 1)
    return _gn(x, sd, key + ".1")


def hourglass(x, presqu, postsqu, sd, key):
    """cmfsm.py:240-303."""
    out = F.relu(convbn_3d(x, sd, key + ".conv1.0", 2))          # :285
    pre = convbn_3d(out, sd, key + ".conv2", 1)                  # :286
    pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)   # :287-290
    out = F.relu(convbn_3d(pre, sd, key + ".conv3.0", 2))        # :292
    out = F.relu(convbn_3d(out, sd, key + ".conv4.0", 1))        # :293
    up = deconvbn_3d(out, sd, key + ".conv5")
    post = F.relu(up + (presqu if presqu is not None else pre))  # :295-299
    out = deconvbn_3d(post, sd, key + ".conv6")                  # :301
    return out, pre, post


def dres0(cost, sd):
    x = F.relu(convbn_3d(cost, sd, "dres0.0"))                   # cmfsm.py:604-608
    return F.relu(convbn_3d(x, sd, "dres0.2"))


def dres1(x, sd):
    y = F.relu(convbn_3d(x, sd, "dres1.0"))                      # cmfsm.py:610-613
    return convbn_3d(y, sd, "dres1.2")


def classif(x, sd, key):
    y = F.relu(convbn_3d(x, sd, key + ".0"))                     # cmfsm.py:621-634
    return F.conv3d(y, sd[key + ".2.weight"], None, 1, 1)


def aggregation_stack(cost, sd):
    """cmfsm.py:684-695,724-725,747-748 -> three [B,D',h,w] logit volumes."""
    cost0 = dres0(cost, sd)
    cost0 = dres1(cost0, sd) + cost0
    out1, pre1, post1 = hourglass(cost0, None, None, sd, "dres2")
    out1 = out1 + cost0
    out2, pre2, post2 = hourglass(out1, pre1, post1, sd, "dres3")
    out2 = out2 + cost0
    out3, pre3, post3 = hourglass(out2, pre1, post2, sd, "dres4")
    out3 = out3 + cost0
    cost1 = classif(out1, sd, "classif1").squeeze(1)
    cost2 = classif(out2, sd, "classif2").squeeze(1) + cost1
    cost3 = classif(out3, sd, "classif3").squeeze(1) + cost2
    return cost1, cost2, cost3


# ----------------------------------------------------------------------------
# a8: soft-argmin  (cmfsm.py:703-706 + disparityregression 111-123)
# ----------------------------------------------------------------------------
def soft_argmin(cost: torch.Tensor) -> torch.Tensor:
    """[B,D,h,w] logits -> [B,h,w] expected disparity index."""
    p = F.softmax(cost, dim=1)
    D = cost.shape[1]
    disp = torch.arange(D, dtype=cost.dtype).view(1, D, 1, 1)
    disp = disp.repeat(cost.shape[0], 1, cost.shape[2], cost.shape[3])   # :121
    return torch.sum(p * disp, 1)


# ----------------------------------------------------------------------------
# a3: eight-related context mapping  (cmfsm.py:304-358, 391-428, 431-593)
# ----------------------------------------------------------------------------
def offset_tables(scale: int = 4):
    """The nine [1,2,s,s] tables of matrix_generation (cmfsm.py:391-428), indices 0..8."""
    s = scale
    half = torch.cat([torch.arange(-(s // 2), 0), torch.arange(1, s // 2 + 1)]).float()   # [-2,-1,1,2]
    centre = torch.stack([half.view(1, s).expand(s, s), half.view(s, 1).expand(s, s)], 0)  # ch0 varies in x, ch1 in y
    up = torch.arange(1, s + 1).float()
    tabs = [centre.clone() for _ in range(9)]
    tabs[1][0] = (s - up + 1).view(1, s).expand(s, s)       # :410
    tabs[2][0] = up.view(1, s).expand(s, s)                 # :411
    tabs[3][1] = (s - up + 1).view(s, 1).expand(s, s)       # :419
    tabs[4][1] = up.view(s, 1).expand(s, s)                 # :420
    tabs[5][0], tabs[5][1] = tabs[2][0], tabs[3][1]         # :412,421
    tabs[6][0], tabs[6][1] = tabs[1][0], tabs[3][1]         # :413,422
    tabs[7][0], tabs[7][1] = tabs[2][0], tabs[4][1]         # :414,423
    tabs[8][0], tabs[8][1] = tabs[1][0], tabs[4][1]         # :415,424
    return [t.unsqueeze(0).clone() for t in tabs]


# (dy, dx) of the LR cell each plane looks at and the table index it uses IN forward().
# Return order of cmfsm.py:551,585-593: centre, l, r, t, b, lt, rt, lb, rb.
# Table aliasing: forward builds distance_matrix5..8 from self.distance_matrix1..4 (cmfsm.py:459-462).
EIGHT_NEIGHBOURS = ((0, 0, 0), (0, -1, 1), (0, 1, 2), (-1, 0, 3), (1, 0, 4),
                    (-1, -1, 1), (-1, 1, 2), (1, -1, 3), (1, 1, 4))


def similarity_mlp(x, sd, key, final_act=False):
    """similarity_measure1 (cmfsm.py:304-358): 1x1 convs 66->32->16->8->1, LeakyReLU between, no bias."""
    x = F.leaky_relu(F.conv2d(x, sd[key + ".conv0.weight"]), LEAKY)
    x = F.leaky_relu(F.conv2d(x, sd[key + ".conv1.weight"]), LEAKY)
    x = F.leaky_relu(F.conv2d(x, sd[key + ".conv2.weight"]), LEAKY)
    x = F.conv2d(x, sd[key + ".conv3.weight"])
    return F.leaky_relu(x, LEAKY) if final_act else x        # sub_8/sub_16 add relu3 (cmfsm_sub_8.py:318,342)


def nn_upsample(x, s):
    """unsqueeze/expand/view nearest-neighbour upsample used everywhere (e.g. cmfsm.py:465-468, 709-712)."""
    return x.repeat_interleave(s, -1).repeat_interleave(s, -2)


def _slices(dy, dx, s, H, W):
    """Row/col slices: (dst region of HR pixels, src region of the shifted upsampled LR map)."""
    def one(d, n):
        if d < 0:
            return slice(s, n), slice(0, n - s)
        if d > 0:
            return slice(0, n - s), slice(s, n)
        return slice(0, n), slice(0, n)
    (ry, sy), (rx, sx) = one(dy, H), one(dx, W)
    return ry, rx, sy, sx


def ecm_weights_eight(lr, hr, sd, key="mapping_matrix.similarity1"):
    """eight_related_context_mapping.forward (cmfsm.py:443-593) -> [B,9,H,W] softmax planes."""
    B, C, H, W = hr.shape
    s = W // lr.shape[-1]
    if s % 2 != 0:
        raise ValueError("odd scale (reference calls exit(), cmfsm.py:448-449)")
    tabs = [t.repeat(B, 1, H // s, W // s) for t in offset_tables(s)]          # :454-462
    lr_up = nn_upsample(lr, s)                                                  # :465-468
    logits = []
    for dy, dx, t in EIGHT_NEIGHBOURS:
        ry, rx, sy, sx = _slices(dy, dx, s, H, W)
        rep = torch.cat([lr_up[:, :, sy, sx], hr[:, :, ry, rx], tabs[t][:, :, sy, sx]], 1)   # e.g. :484
        val = similarity_mlp(rep, sd, key)
        full = torch.full((B, 1, H, W), PAD_LOGIT, dtype=hr.dtype)             # padding1/2 :451-452
        full[:, :, ry, rx] = val
        logits.append(full)
    return F.softmax(torch.cat(logits, 1), dim=1)                               # :551-552


# ----------------------------------------------------------------------------
# a9: NN-upsample + 9-neighbour aggregation  (cmfsm.py:709-723)
# ----------------------------------------------------------------------------
def ecm_aggregate_eight(d_lr: torch.Tensor, w9: torch.Tensor, scale: int) -> torch.Tensor:
    """d_lr [B,h,w], w9 [B,9,H,W] -> [B,1,H,W]  (the reference's [i,i] diagonal for B>1, quirk Q1)."""
    s = scale
    pred = (s * nn_upsample(d_lr, s)).unsqueeze(1)                              # :709-712
    H, W = pred.shape[-2:]
    out = pred * w9[:, 0:1]                                                     # :714
    for n, (dy, dx, _) in enumerate(EIGHT_NEIGHBOURS):
        if n == 0:
            continue
        ry, rx, sy, sx = _slices(dy, dx, s, H, W)
        out[:, :, ry, rx] += pred[:, :, sy, sx] * w9[:, n:n + 1, ry, rx]        # :715-723
    return out


# ----------------------------------------------------------------------------
# 2-D encoder (NOT hot path; needed so the whole model can be checked end to end)
# feature_extraction, cmfsm.py:126-236
# ----------------------------------------------------------------------------
def _convbn2d(x, sd, key, stride, pad, dil):
    x = F.conv2d(x, sd[key + ".0.weight"], None, stride, dil if dil > 1 else pad, dil)     # :36-46
    return _gn(x, sd, key + ".1")


def _basic_block(x, sd, key, stride, pad, dil):
    out = F.relu(_convbn2d(x, sd, key + ".conv1.0", stride, pad, dil))                      # :76-85
    out = _convbn2d(out, sd, key + ".conv2", 1, pad, dil)
    if key + ".downsample.0.weight" in sd:
        x = _gn(F.conv2d(x, sd[key + ".downsample.0.weight"], None, stride), sd, key + ".downsample.1")
    return out + x


def _layer(x, sd, key, blocks, stride, pad, dil):
    for i in range(blocks):
        x = _basic_block(x, sd, f"{key}.{i}", stride if i == 0 else 1, pad, dil)
    return x


def feature_extraction(x, sd, key="feature_extraction"):
    k = key
    y = F.relu(_convbn2d(x, sd, k + ".firstconv.0", 1, 1, 1))
    y = F.relu(_convbn2d(y, sd, k + ".firstconv.2", 1, 1, 1))
    y = F.relu(_convbn2d(y, sd, k + ".firstconv.4", 1, 1, 1))
    out_all = F.conv2d(y, sd[k + ".firstconv.6.weight"], None, 1, 1)                        # :138
    y = F.relu(_gn(out_all, sd, k + ".secondconv.0"))                                       # :140-141
    y = F.relu(_convbn2d(y, sd, k + ".secondconv.2", 2, 1, 1))
    y = F.relu(_convbn2d(y, sd, k + ".secondconv.4", 1, 1, 1))
    out_rt = _layer(y, sd, k + ".layer1", 3, 1, 1, 1)
    out_raw = _layer(out_rt, sd, k + ".layer2", 16, 2, 1, 1)
    y = _layer(out_raw, sd, k + ".layer3", 3, 1, 1, 1)
    out_skip = _layer(y, sd, k + ".layer4", 3, 1, 1, 2)
    size = out_skip.shape[-2:]
    branches = []
    for name, pool in (("branch1", 64), ("branch2", 32), ("branch3", 16), ("branch4", 8)):
        b = F.avg_pool2d(out_skip, pool, pool)
        b = F.relu(_convbn2d(b, sd, f"{k}.{name}.1", 1, 0, 1))
        branches.append(F.interpolate(b, size, mode="bilinear", align_corners=False))
    feat = torch.cat((out_raw, out_skip, branches[3], branches[2], branches[1], branches[0]), 1)   # :231-233
    feat = F.relu(_convbn2d(feat, sd, k + ".lastconv.0", 1, 1, 1))
    feat = F.conv2d(feat, sd[k + ".lastconv.2.weight"])
    return feat, out_rt, out_all


# ----------------------------------------------------------------------------
# whole model, cmfsm.forward (cmfsm.py:655-774)
# ----------------------------------------------------------------------------
def hot_path(lr_l, hr_l, lr_r, sd, maxdisp=192):
    """Everything after the encoder: returns (pred1, pred2, pred3) each [B,1,H,W]."""
    s = hr_l.shape[-1] // lr_l.shape[-1]
    w9 = ecm_weights_eight(lr_l, hr_l, sd)
    cost = cost_volume(lr_l, lr_r, maxdisp // s)
    c1, c2, c3 = aggregation_stack(cost, sd)
    return tuple(ecm_aggregate_eight(soft_argmin(c), w9, s) for c in (c1, c2, c3))


def cmfsm_forward(left, right, sd, maxdisp=192):
    lr_l, _, hr_l = feature_extraction(left, sd)
    lr_r, _, _ = feature_extraction(right, sd)
    return hot_path(lr_l, hr_l, lr_r, sd, maxdisp)


def train_loss(preds, gt, maxdisp=192):
    """train.py:162,168-174: masked smooth-L1 on the three heads, weights 0.5/0.7/1.0."""
    mask = (gt < maxdisp) & (gt > 0)
    o1, o2, o3 = (p.squeeze(1) for p in preds)
    return (0.5 * F.smooth_l1_loss(o1[mask], gt[mask], reduction="mean")
            + 0.7 * F.smooth_l1_loss(o2[mask], gt[mask], reduction="mean")
            + F.smooth_l1_loss(o3[mask], gt[mask], reduction="mean"))


FLYING3D_MEAN, FLYING3D_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # Flying3d.py:26-27


def flying3d_sample(frame, split="train", crop=(0, 0)):
    """cmf/loader/Flying3d.py:49-99 for one float32 frame [H,W,7] (left RGB, right RGB, disparity; flying3ddata.py:34-39)
    -> (left [3,h,w], right [3,h,w], disparity [h,w], image [3,h,w]).  `crop` = the (x1, y1) the loader draws with
    random.randint in train mode (row, column of the 256x512 window).  torchvision's ToTensor on a float ndarray is a
    HWC->CHW transpose (no scaling) and Normalize is `(t - mean[:,None,None]) / std[:,None,None]` in the tensor's dtype;
    torchvision is absent from this image, so these two documented semantics are restated here (parity unpinned for
    them -- the crop/pad/scale arithmetic is the loader's own numpy code, line by line)."""
    import numpy as np
    data = np.asarray(frame)
    if split == "train":
        x1, y1 = crop
        data = data[x1:x1 + 256, y1:y1 + 512, :]                  # :51-56
    else:
        padding = data[-36:, ...]                                  # :66
        data = np.concatenate([data[0:540, 0:960, :], padding], 0)  # :67-71
    left = data[..., 0:3] / 255                                    # :73
    image = torch.from_numpy(np.ascontiguousarray(data[..., 0:3].transpose(2, 0, 1)))      # :75-76 ToTensor on floats
    right = data[..., 3:6] / 255                                   # :78
    disparity = data[..., 6]                                       # :79
    mean = torch.tensor(FLYING3D_MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(FLYING3D_STD, dtype=torch.float32)[:, None, None]

    def trans(a):                                                  # :90-96 ToTensor + Normalize, then .float()
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return ((t - mean.to(t.dtype)) / std.to(t.dtype)).float()

    return trans(left), trans(right), torch.from_numpy(np.ascontiguousarray(disparity)).float(), image


def kitti_metrics(pred3, gt, maxdisp=192):
    """train_kitti.py:213-216: end-point error and the 3-px / 5 % error rate (in %) of the last head over the mask.
    Pinned by fixture g10 (the reference's own statements executed in the build container, make_golden_eval.py)."""
    mask = (gt < maxdisp) & (gt > 0)
    o3 = pred3.squeeze(1) if pred3.dim() == gt.dim() + 1 else pred3
    err = torch.abs(o3[mask] - gt[mask])
    epe = torch.mean(err)
    good = torch.where((err < 3) | (err < 0.05 * gt[mask]), torch.ones_like(err), torch.zeros_like(err))
    return epe, 100 - torch.sum(good) / torch.sum(mask) * 100


# ----------------------------------------------------------------------------
# Eval leg of the harness: test.py:69-94, test_kitti.py:163-168, KITTI.py:98-108
# Pinned by tests/golden/g9_eval_harness.npz, which make_golden.py produces by EXECUTING those reference lines.
# ----------------------------------------------------------------------------
def sceneflow_eval_epe(output3, disparity, maxdisp=192):
    """test.py:69-94: crop ground truth and prediction to [:540,:960], three masks, mean |error| under each.
    output3 [B,1,H,W] (or [B,H,W]), disparity [B,Hg,Wg] -> (loss, loss_non, loss_true) as python floats."""
    disparity = disparity[:, :540, :960]                                                        # :69
    local = torch.arange(disparity.shape[-1]).repeat(disparity.shape[0], disparity.shape[1], 1).view_as(disparity).float()  # :70
    mask_non = (disparity < maxdisp) & (disparity >= 0) & ((local - disparity) >= 0)           # :71
    mask_true = (disparity < maxdisp) & (disparity > 0) & ((local - disparity) >= 0)           # :72
    mask = (disparity < maxdisp) & (disparity >= 0)                                             # :73
    o = output3.squeeze(1) if output3.dim() == 4 else output3
    o = o[:, :540, :960]                                                                        # :83
    loss = torch.mean(torch.abs(o[mask] - disparity[mask]))                                     # :92
    loss_non = torch.mean(torch.abs(o[mask_non] - disparity[mask_non]))                         # :93
    loss_true = torch.mean(torch.abs(o[mask_true] - disparity[mask_true]))                      # :94
    return loss.item(), loss_non.item(), loss_true.item()


def kitti_disparity_uint16(output3, h, w):
    """test_kitti.py:163-168 for sample 0: output3*256 -> numpy uint16 (C cast), un-pad the top/left padding."""
    import numpy as np
    o = output3.squeeze(1) if output3.dim() == 4 else output3
    pre = (o * 256).data.cpu().numpy().astype("uint16")                                         # :163-164
    pre = pre[0, -h:, -w:]                                                                      # :165
    return np.reshape(pre, [h, w])                                                              # :168


def kitti_eval_pad(frame, th=384, tw=1248):
    """KITTI.__getitem__, eval branch (cmf/loader/KITTI.py:98-108) on one [H,W,7] frame.  `padding_h` / `padding_w` are
    numpy VIEWS, so zeroing their disparity channel also zeroes it in the array they were cut from -- kept as is.
    Returns the padded [th,tw,7] array (the caller's `frame` is modified like the loader's `data`)."""
    import numpy as np
    data = frame
    h, w = data.shape[0], data.shape[1]                                                         # :99
    padding_h = data[:(th - h), :, :]                                                           # :103
    padding_h[:, :, 6] = 0                                                                      # :104
    data = np.concatenate([padding_h, data], 0)                                                 # :105
    padding_w = data[:, :(tw - w), :]                                                           # :106
    padding_w[:, :, 6] = 0                                                                      # :107
    return np.concatenate([padding_w, data], 1)                                                 # :108


def kitti_eval_sample(frame, th=384, tw=1248):
    """KITTI.__getitem__ eval branch + transform (KITTI.py:98-126): -> (left, right [3,th,tw], disparity [th,tw]); the
    ToTensor / Normalize semantics are restated as in flying3d_sample (torchvision is absent: unpinned for those two)."""
    import numpy as np
    data = kitti_eval_pad(np.array(frame, copy=True), th, tw)
    mean = torch.tensor(FLYING3D_MEAN, dtype=torch.float32)[:, None, None]                      # KITTI.py:54-55: same stats
    std = torch.tensor(FLYING3D_STD, dtype=torch.float32)[:, None, None]

    def trans(a):
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return ((t - mean.to(t.dtype)) / std.to(t.dtype)).float()

    return trans(data[..., 0:3] / 255), trans(data[..., 3:6] / 255), torch.from_numpy(np.ascontiguousarray(data[..., 6])).float()


# ============================================================================
# The other registered architectures
# ============================================================================
# six_related_context_mapping (cmfsm_sub_8.py:440-572; identical text in cmfsm_sub_16.py, cm_sub_4/8/16.py):
# reference-image planes in return order [c, r, l, t, b] (cmfsm_sub_8.py:566), right uses table 1 and left table 2
# (:503,525); target-image planes [c, r, l] (:568); zero padding (:461-462); output softmax*logit (:572).
SIX_LEFT = ((0, 0, 0), (0, 1, 1), (0, -1, 2), (-1, 0, 3), (1, 0, 4))
SIX_RIGHT = ((0, 0, 0), (0, 1, 1), (0, -1, 2))


def _six_planes(lr, hr, sd, key, neighbours):
    B, C, H, W = hr.shape
    s = W // lr.shape[-1]
    tabs = [t.repeat(B, 1, H // s, W // s) for t in offset_tables(s)]        # generic even scale (:455-470)
    lr_up = nn_upsample(lr, s)
    logits = []
    for dy, dx, t in neighbours:
        ry, rx, sy, sx = _slices(dy, dx, s, H, W)
        rep = torch.cat([lr_up[:, :, sy, sx], hr[:, :, ry, rx], tabs[t][:, :, sy, sx]], 1)
        val = similarity_mlp(rep, sd, key, final_act=True)                    # relu3, cmfsm_sub_8.py:318,342
        full = torch.zeros(B, 1, H, W, dtype=hr.dtype)                         # zero padding participates in the softmax
        full[:, :, ry, rx] = val
        logits.append(full)
    allp = torch.cat(logits, 1)
    return F.softmax(allp, dim=1) * allp


def ecm_weights_six(lr, hr, lr_r, hr_r, sd, key="mapping_matrix.similarity1"):
    """-> (m5 [B,5,H,W] for the reference image, mt3 [B,3,H,W] for the target image)."""
    return _six_planes(lr, hr, sd, key, SIX_LEFT), _six_planes(lr_r, hr_r, sd, key, SIX_RIGHT)


def ecm_aggregate_five(d_lr, m5, scale):
    """cmfsm_sub_8.py:763-772: d_lr [B,h,w], m5 [B,5,H,W] (c,r,l,t,b) -> [B,1,H,W]."""
    s = scale
    pred = (s * nn_upsample(d_lr, s)).unsqueeze(1)
    H, W = pred.shape[-2:]
    out = pred * m5[:, 0:1]
    for n, (dy, dx, _) in enumerate(SIX_LEFT):
        if n == 0:
            continue
        ry, rx, sy, sx = _slices(dy, dx, s, H, W)
        out[:, :, ry, rx] = out[:, :, ry, rx] + pred[:, :, sy, sx] * m5[:, n:n + 1, ry, rx]
    return out


def volume_mapping(cost_lr, m5, mt3, scale, maxdisp=192):
    """cmfsm_sub_16.py:767-801 (same in cm_sub_*): cost_lr [B,Dl,h,w] (already accumulated over heads),
    m5 [B,5,H,W], mt3 [B,3,H,W] -> disparity [B,H,W].  Follows the reference op sequence incl. the python loop."""
    s = scale
    cost = nn_upsample(cost_lr, s).repeat_interleave(s, 1)                       # :768-773  [B,maxdisp,H,W]
    H, W = cost.shape[-2:]
    fused = cost * m5[:, 0:1]
    for n, (dy, dx, _) in enumerate(SIX_LEFT):                                    # :774-778
        if n == 0:
            continue
        ry, rx, sy, sx = _slices(dy, dx, s, H, W)
        fused[:, :, ry, rx] = fused[:, :, ry, rx] + cost[:, :, sy, sx] * m5[:, n:n + 1, ry, rx]
    vols = [torch.ones_like(cost) for _ in range(3)]                              # :782-784
    for d in range(maxdisp):                                                      # :785-794
        for v, plane in zip(vols, (mt3[:, 0], mt3[:, 1], mt3[:, 2])):
            if d == 0:
                v[:, 0] = plane
            elif d < W:
                v[:, d, :, d:] = plane[:, :, :-d]
    t0, tr, tl = vols
    out = fused * t0                                                              # :796
    out[:, :-s] = out[:, :-s] + fused[:, s:] * tl[:, :-s]                         # :797
    out[:, s:] = out[:, s:] + fused[:, :-s] * tr[:, s:]                           # :798
    return soft_argmin(out)                                                       # :800-801


def trilinear_head(cost_lr, maxdisp, H, W):
    """bilinear_cmf.py:447-471: cost_lr [B,Dl,h,w] -> disparity [B,H,W]."""
    up = F.interpolate(cost_lr.unsqueeze(1), [maxdisp, H, W], mode="trilinear", align_corners=False).squeeze(1)
    return soft_argmin(up)


ARCH_SPEC = {  # name: (head kind, hourglasses, accumulate logits across heads)
    "cmfsm": ("eight", 3, True), "cmfsm_sub_8": ("five", 3, False), "cmfsm_sub_16": ("volume", 3, True),
    "cm_sub_4": ("volume", 1, False), "cm_sub_8": ("volume", 1, False), "cm_sub_16": ("volume", 1, False),
    "bilinear_cmf": ("trilinear", 3, True), "bilinear_cmf_sub_8": ("trilinear", 3, True),
    "bilinear_cmf_sub_16": ("trilinear", 3, True),
}


def hot_path_arch(arch, lr_l, hr_l, lr_r, hr_r, sd, maxdisp=192):
    """Post-encoder path of any registered architecture -> 3 predictions (cm_sub_* repeat pred1, cm_sub_4.py:784)."""
    kind, nhg, accumulate = ARCH_SPEC[arch]
    s = hr_l.shape[-1] // lr_l.shape[-1]
    cost = cost_volume(lr_l, lr_r, maxdisp // s)
    cost0 = dres0(cost, sd)
    cost0 = dres1(cost0, sd) + cost0
    outs, pre1, post = [], None, None
    x = cost0
    for i in range(nhg):
        key = f"dres{i + 2}"
        out, pre, post_new = hourglass(x, pre1 if i > 0 else None, post, sd, key)
        if i == 0:
            pre1 = pre
        post = post_new
        x = out + cost0
        outs.append(x)
    logits = []
    for i, o in enumerate(outs):
        c = classif(o, sd, f"classif{i + 1}").squeeze(1)
        if accumulate and logits:
            c = c + logits[-1]
        logits.append(c)
    if kind == "eight":
        w9 = ecm_weights_eight(lr_l, hr_l, sd)
        preds = [ecm_aggregate_eight(soft_argmin(c), w9, s) for c in logits]
    elif kind == "five":
        m5, _ = ecm_weights_six(lr_l, hr_l, lr_r, hr_r, sd)
        preds = [ecm_aggregate_five(soft_argmin(c), m5, s) for c in logits]
    elif kind == "volume":
        m5, mt3 = ecm_weights_six(lr_l, hr_l, lr_r, hr_r, sd)
        preds = [volume_mapping(c, m5, mt3, s, maxdisp) for c in logits]
    else:
        H, W = hr_l.shape[-2:]
        preds = [trilinear_head(c, maxdisp, H, W) for c in logits]
    while len(preds) < 3:
        preds.append(preds[0])
    return tuple(preds)
