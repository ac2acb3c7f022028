"""End-to-end gradients on an fp64 yardstick.

At random initialisation the parameter gradients of this network are sums of millions of terms with ReLU / LeakyReLU /
smooth-L1 kinks, so two correct fp32 evaluations differ by a few tenths of a percent.  Instead of accepting "5 % of the
largest element", the fixtures (tests/golden/make_golden_fp64.py) hold the REFERENCE's own run in fp64 -- the truth -- and
how far the reference's own fp32 run is from it.  The HIP path must be as close to the truth as another fp32 evaluation
can be: within K = 4 x the reference-fp32's own distance plus a floor of 1e-3 of the quantity's scale (the floor covers
parameters where the reference's fp32 run happens to land unusually close)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ecm_oracle as O
from oracle.weights import make_state_dict, seeded, tensor_for

pytestmark = pytest.mark.gpu
K = 4.0
FLOOR = 1e-3


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


def _z(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def _check_params(model, z, label, skip_prefix=None, population=False):
    """Every parameter's gradient against the reference's fp64 run.  Per parameter: norm and a random projection (a norm cannot
    see a permuted gradient; the projection can) within K x the reference-fp32's own distance from fp64 (+ FLOOR), and the
    stored full tensors element-wise likewise.

    `population` (the full-frame fixtures g12 / g12k): at random initialisation the gradients of the first encoder layers carry
    ~1e-3 of relative noise from ReLU / LeakyReLU decisions that flip with ANY change of rounding, the reference's own fp32 run
    included -- and for a single parameter that run's distance from fp64 is one draw of that noise (1.5e-4 relative for
    `firstconv.0.1.bias` at 384x1248, where the same run is 2.5e-3 off on `firstconv.2.1.bias` two layers on).  K x one lucky
    draw is not a bound.  So there a parameter's norm may also be as far off as the reference-fp32's WORST parameter (x 1.25),
    and what is held tight is the population: median and rms of the relative norm errors over all parameters within 1.5 x the
    reference-fp32's (measured: median 1.0e-4 vs 1.1e-4, rms 4.4e-4 vs 4.4e-4 at 384x1248; 1.4e-4 vs 1.0e-4, 3.8e-4 vs 4.5e-4
    at 576x960)."""
    worst, bad = [], []
    n_full = 0
    rel_hip, rel_ref = [], []
    names = [k for k, _ in model.named_parameters() if not (skip_prefix and k.startswith(skip_prefix))]
    ref_worst = max(abs(float(z["gn32_" + k.replace(".", "_")]) - float(z["gn64_" + k.replace(".", "_")]))
                    / max(float(z["gn64_" + k.replace(".", "_")]), 1e-30) for k in names) if population else 0.0
    for k, p in model.named_parameters():
        if skip_prefix and k.startswith(skip_prefix):
            continue
        kk = k.replace(".", "_")
        n64, n32 = float(z["gn64_" + kk]), float(z["gn32_" + kk])
        p64, p32 = float(z["gp64_" + kk]), float(z["gp32_" + kk])
        g = p.grad.detach().double().cpu() if p.grad is not None else torch.zeros(p.shape, dtype=torch.float64)
        n_hip = float(g.norm())
        p_hip = float((g * seeded("proj:" + k, *g.shape).double()).sum())
        tol_n = max(K * abs(n32 - n64) + FLOOR * n64, 1.25 * ref_worst * n64) + 1e-12
        tol_p = K * abs(p32 - p64) + 2e-2 * n64 + 1e-12          # a projection error is ~ |e| x N(0,1): floor at 2 % of |g|
        if abs(n_hip - n64) > tol_n:
            bad.append(f"{label} {k}: |g| {n_hip:.6e} vs fp64 {n64:.6e} (reference fp32 {n32:.6e}); tol {tol_n:.2e}")
        if abs(p_hip - p64) > tol_p:
            bad.append(f"{label} {k}: projection {p_hip:.6e} vs fp64 {p64:.6e} (reference fp32 {p32:.6e})")
        rel_hip.append(abs(n_hip - n64) / max(n64, 1e-30))
        rel_ref.append(abs(n32 - n64) / max(n64, 1e-30))
        worst.append((abs(n_hip - n64) / max(abs(n32 - n64), FLOOR * n64 / K + 1e-30), k))
        if "g64_" + kk in z:                                     # full tensors
            t64 = torch.from_numpy(z["g64_" + kk])
            e = (g - t64).abs()
            scale = float(t64.abs().max())
            rms64 = float(t64.pow(2).mean().sqrt())
            emax32, erms32 = float(z["e32max_" + kk]), float(z["e32rms_" + kk])
            if float(e.max()) > K * emax32 + FLOOR * scale:
                bad.append(f"{label} {k}: max |hip - fp64| {float(e.max()):.3e} vs the reference fp32's {emax32:.3e} (scale {scale:.3e})")
            if float(e.pow(2).mean().sqrt()) > K * erms32 + FLOOR * rms64:
                bad.append(f"{label} {k}: rms |hip - fp64| {float(e.pow(2).mean().sqrt()):.3e} vs the reference fp32's {erms32:.3e}")
            n_full += 1
    if population:
        rh, rr = np.array(rel_hip), np.array(rel_ref)
        stats = (float(np.median(rh)), float(np.median(rr)), float(np.sqrt((rh ** 2).mean())), float(np.sqrt((rr ** 2).mean())))
        print(f"{label}: relative |g| error over {len(rh)} parameters: median hip {stats[0]:.2e} / reference fp32 {stats[1]:.2e}, "
              f"rms {stats[2]:.2e} / {stats[3]:.2e}, max {rh.max():.2e} / {rr.max():.2e}")
        if stats[0] > 1.5 * stats[1] + 1e-5 or stats[2] > 1.5 * stats[3] + 1e-5:
            bad.append(f"{label}: population of relative norm errors: median {stats[0]:.2e} vs {stats[1]:.2e}, rms {stats[2]:.2e} vs {stats[3]:.2e}")
    assert not bad, "\n".join(bad)
    return n_full, sorted(worst)[-3:]


def test_full_cmfsm_train_step_against_fp64_reference(ecm):
    """train.py:162-181 on the 256x512 fixture: predictions, loss and EVERY parameter's gradient vs the reference in fp64."""
    z = _z("g8d_full_cmfsm_256x512_fp64")
    model = ecm.get_model("cmfsm")
    model.load_state_dict({k: tensor_for(k, v.shape) for k, v in model.state_dict().items()})
    model = model.cuda().train()
    left, right = seeded("g8.left", 1, 3, 256, 512).cuda(), seeded("g8.right", 1, 3, 256, 512).cuda()
    gt = (torch.rand(1, 256, 512, generator=torch.Generator().manual_seed(8)) * 191.0).cuda()
    o = model(left, right)
    loss = O.train_loss(o, gt)
    loss.backward()
    for i in (1, 2, 3):
        d64 = (o[i - 1].detach().double().cpu()[..., ::4, ::4] - torch.from_numpy(z[f"o{i}_64"])).abs()
        # SURVEY section 7's stated end-to-end tolerance is max 2e-2 px, mean 1e-3 px (disparities span 0..191 px); the
        # reference's own fp32 run sits at max 7e-4 / mean 1.3e-4 from this fp64 truth and the HIP path at 3e-4 / 5e-5
        # (tests/diag_stage_error.py), so the bound asserted here is ten times tighter than the stated one
        assert float(d64.max()) <= 2e-3 and float(d64.mean()) <= 1e-4 * i + 1e-4, (i, float(d64.max()), float(d64.mean()))
    l64, l32 = float(z["loss_64"]), float(z["loss_32"])
    assert abs(float(loss.detach()) - l64) <= K * abs(l32 - l64) + 1e-5 * l64, (float(loss.detach()), l64, l32)
    n_full, worst = _check_params(model, z, "cmfsm")
    assert n_full >= 14, n_full
    print("worst norm-error ratios vs the yardstick:", worst)


def test_full_frame_train_step_against_fp64_reference(ecm):
    """BASELINE config 2's training step for ONE sample at the SceneFlow frame size (576x960; fixture g12, round 4): the raw
    frame of oracle/weights.py:fullframe_frame goes through the HIP frame preparation, the model and the fused loss kernel;
    predictions, loss and EVERY parameter's gradient are held against the reference's own fp64 run with the reference-fp32's
    distance from it as the yardstick -- the same assertion as at 256x512, at the size the headline is quoted on."""
    from oracle.weights import fullframe_frame
    z = _z("g12_full_cmfsm_576x960_fp64")
    model = ecm.get_model("cmfsm")
    model.load_state_dict({k: tensor_for(k, v.shape) for k, v in model.state_dict().items()})
    model = model.cuda().train()
    frame = torch.from_numpy(fullframe_frame("sceneflow")[None].copy()).cuda()
    left, right, gt = ecm.ops.frame_prep(frame, [0], [0], 576, 960, split=540, tail=36)
    o = model(left, right)
    loss, metrics = ecm.ops.stereo_loss3(o, gt, 192)                 # train.py:162-174 in one kernel
    loss.backward()
    ecm.ops.check_async_errors()
    for i in (1, 2, 3):
        d64 = (o[i - 1].detach().double().cpu()[..., ::4, ::4] - torch.from_numpy(z[f"o{i}_64"])).abs()
        r32 = (torch.from_numpy(z[f"o{i}_32"]).double() - torch.from_numpy(z[f"o{i}_64"])).abs()
        assert float(d64.max()) <= max(2e-3, K * float(r32.max())) and float(d64.mean()) <= max(1e-4 * i + 1e-4, K * float(r32.mean())), \
            (i, float(d64.max()), float(d64.mean()), float(r32.max()), float(r32.mean()))
    l64, l32 = float(z["loss_64"]), float(z["loss_32"])
    assert abs(float(loss.detach()) - l64) <= K * abs(l32 - l64) + 1e-5 * abs(l64), (float(loss.detach()), l64, l32)
    n_full, worst = _check_params(model, z, "cmfsm 576x960", population=True)
    assert n_full >= 14, n_full
    print("worst norm-error ratios vs the yardstick (576x960):", worst)


def test_kitti_frame_train_step_against_fp64_reference(ecm):
    """BASELINE config 4's per-GPU workload (fixture g12k, round 4): the KITTI-shaped raw frame (375x1242) through the HIP
    top / left repeat padding to 384x1248 (KITTI.py:98-108), the model and the fused loss (train_kitti.py:186-207); predictions,
    loss and every parameter's gradient against the reference's fp64 run, the reference-fp32's distance as the yardstick."""
    from oracle.weights import fullframe_frame
    z = _z("g12k_full_cmfsm_384x1248_fp64")
    model = ecm.get_model("cmfsm")
    model.load_state_dict({k: tensor_for(k, v.shape) for k, v in model.state_dict().items()})
    model = model.cuda().train()
    frame = torch.from_numpy(fullframe_frame("kitti")[None].copy()).cuda()
    left, right, gt = ecm.ops.frame_prep_kitti_eval(frame, 384, 1248)
    o = model(left, right)
    loss, metrics = ecm.ops.stereo_loss3(o, gt, 192)
    loss.backward()
    ecm.ops.check_async_errors()
    for i in (1, 2, 3):
        d64 = (o[i - 1].detach().double().cpu()[..., ::4, ::4] - torch.from_numpy(z[f"o{i}_64"])).abs()
        r32 = (torch.from_numpy(z[f"o{i}_32"]).double() - torch.from_numpy(z[f"o{i}_64"])).abs()
        assert float(d64.max()) <= max(2e-3, K * float(r32.max())) and float(d64.mean()) <= max(1e-4 * i + 1e-4, K * float(r32.mean())), \
            (i, float(d64.max()), float(d64.mean()), float(r32.max()), float(r32.mean()))
    l64, l32 = float(z["loss_64"]), float(z["loss_32"])
    assert abs(float(loss.detach()) - l64) <= K * abs(l32 - l64) + 1e-5 * abs(l64), (float(loss.detach()), l64, l32)
    n_full, worst = _check_params(model, z, "cmfsm 384x1248", population=True)
    assert n_full >= 14, n_full
    print("worst norm-error ratios vs the yardstick (384x1248):", worst)


_ARCHS64 = {"cmfsm_sub_16": (16, 4, 4, 8), "bilinear_cmf_sub_16": (16, 4, 4, 4), "cmfsm_sub_8": (8, 4, 8, 8)}   # s, h, w, full tensors


@pytest.mark.parametrize("arch", sorted(_ARCHS64))
def test_arch_hot_path_against_fp64_reference(ecm, arch):
    """Post-encoder path of three more registered architectures behind the stub encoder -- cmfsm_sub_16 (six-related weights
    on both images + the fused volume-mapping head, cmfsm_sub_16.py:722-850), bilinear_cmf_sub_16 (test.py:111's default:
    trilinear soft-argmin head, bilinear_cmf.py:447-471) and cmfsm_sub_8 (5-neighbour aggregation, cmfsm_sub_8.py:440-572):
    predictions, feature gradients and every parameter's gradient vs the reference run in fp64."""
    s, h, w, min_full = _ARCHS64[arch]
    z = _z(f"arch_{arch}_fp64")
    model = ecm.get_model(arch)
    sd = {k: tensor_for(k, v.shape) for k, v in model.state_dict().items() if not k.startswith("feature_extraction")}
    model.load_state_dict(sd, strict=False)
    model = model.cuda()
    feats = [seeded(f"{arch}.{n}", 1, 32, *sz).cuda().requires_grad_() for n, sz in
             (("lr_l", (h, w)), ("hr_l", (s * h, s * w)), ("lr_r", (h, w)), ("hr_r", (s * h, s * w)))]
    preds = model.hot_path(*feats)
    loss = sum((p * seeded(f"{arch}.G{i}", *p.shape).cuda()).sum() for i, p in enumerate(preds))
    loss.backward()
    for i, p in enumerate(preds, 1):
        d = (p.detach().double().cpu() - torch.from_numpy(z[f"pred{i}_64"])).abs()
        assert float(d.max()) <= 2e-2 and float(d.mean()) <= 1e-3, (i, float(d.max()), float(d.mean()))
    n_feat = 0
    for nm, t in zip(("g_lr_l", "g_hr_l", "g_lr_r", "g_hr_r"), feats):
        if nm + "_64" not in z:                                  # the reference leaves this feature map unused ...
            assert t.grad is None or float(t.grad.abs().max()) == 0.0, nm   # ... and so does the HIP path
            continue
        t64 = torch.from_numpy(z[nm + "_64"])
        e = (t.grad.double().cpu() - t64).abs()
        assert float(e.max()) <= K * float(z["e32max_" + nm]) + FLOOR * float(t64.abs().max()), \
            (nm, float(e.max()), float(z["e32max_" + nm]), float(t64.abs().max()))
        n_feat += 1
    assert n_feat >= 2, n_feat
    n_full, worst = _check_params(model, z, arch, skip_prefix="feature_extraction")
    assert n_full >= min_full, n_full
    print("worst norm-error ratios vs the yardstick:", worst)
