"""GPU parity of the harness legs around the model, through the C ABI:
  * SceneFlow evaluation (test.py:69-94) and the KITTI submission image (test_kitti.py:163-168: integer output, bit-exact)
    against fixture g9 (produced by executing the reference's statements) and against the oracle restatements;
  * packed uint8 + fp16/fp32 shards decoded by the frame-preparation kernel: bit-exact against the float32-frame path;
  * the KITTI eval padding (cmf/loader/KITTI.py:98-108), bit-exact against the oracle restatement and fixture g9."""
import hashlib
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ecm_oracle as O
from oracle.weights import eval_harness_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


@pytest.fixture(scope="module")
def g9():
    with np.load(os.path.join(GOLDEN, "g9_eval_harness.npz")) as z:
        return {k: z[k] for k in z.files}


def test_sceneflow_eval_epe_golden(ecm, g9):
    inp = eval_harness_inputs()
    out = ecm.ops.eval_epe(inp["sf_output3"].cuda(), inp["sf_disparity"].cuda()).cpu().double().numpy()
    np.testing.assert_allclose(out[:3], g9["epe"][:3], rtol=2e-6)          # float32 means (double partial sums here)
    assert np.array_equal(out[3:], g9["epe"][3:])                          # the three mask counts: exact


@pytest.mark.parametrize("B,Hp,Wp,Hg,Wg,ch,cw", [(1, 16, 24, 16, 24, 16, 24), (3, 40, 70, 37, 65, 30, 61), (2, 576, 960, 576, 960, 540, 960)])
def test_sceneflow_eval_epe_vs_oracle(ecm, B, Hp, Wp, Hg, Wg, ch, cw):
    g = torch.Generator().manual_seed(B * 1000 + Hp)
    gt = torch.rand(B, Hg, Wg, generator=g) * 240.0 - 24.0
    pred = torch.rand(B, 1, Hp, Wp, generator=g) * 192.0
    got = ecm.ops.eval_epe(pred.cuda(), gt.cuda(), ch, cw).cpu()
    # oracle: test.py's lines with its hard-coded 540 x 960 crop -> feed it tensors already cropped to (ch, cw)
    want = O.sceneflow_eval_epe(pred[:, :, :ch, :cw], gt[:, :ch, :cw])
    np.testing.assert_allclose(got[:3].numpy(), np.array(want), rtol=2e-6)


def test_empty_mask_gives_nan_like_the_reference(ecm):
    gt = torch.full((1, 8, 8), 500.0)
    out = ecm.ops.eval_epe(torch.zeros(1, 8, 8).cuda(), gt.cuda(), 8, 8).cpu()
    assert torch.isnan(out[:3]).all() and (out[3:] == 0).all()


def test_kitti_uint16_image_golden_bit_exact(ecm, g9):
    inp = eval_harness_inputs()
    h, w = inp["kitti_hw"]
    img = ecm.ops.disparity_to_uint16(inp["kitti_output3"].cuda(), h, w).cpu().numpy()
    assert img.dtype == np.uint16 and img.shape == (1, h, w)
    assert hashlib.sha256(np.ascontiguousarray(img[0]).tobytes()).digest() == g9["u16_sha256"].tobytes()
    assert np.array_equal(img[0, :8], g9["u16_head"]) and np.array_equal(img[0, ::5, ::5], g9["u16_sub"])


def test_kitti_uint16_ragged_batch_vs_oracle(ecm):
    g = torch.Generator().manual_seed(77)
    pred = torch.rand(3, 1, 384, 1248, generator=g) * 191.0
    hs, ws = [375, 370, 384], [1242, 1226, 1248]
    img = ecm.ops.disparity_to_uint16(pred.cuda(), hs, ws).cpu().numpy()
    assert img.shape == (3, 384, 1248)
    for b in range(3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = O.kitti_disparity_uint16(pred[b:b + 1], hs[b], ws[b])
        assert np.array_equal(img[b, :hs[b], :ws[b]], want)
        assert not img[b, hs[b]:].any() and not img[b, :, ws[b]:].any()


def _frames(B, H, W, seed):
    rs = np.random.RandomState(seed)
    rgb = rs.randint(0, 256, size=(B, H, W, 6)).astype(np.uint8)
    disp = (rs.rand(B, H, W) * 250.0).astype(np.float32)
    frames = np.concatenate([rgb.astype(np.float32), disp[..., None]], 3)           # flying3ddata.py:34-39
    return rgb, disp, frames


@pytest.mark.parametrize("half", [False, True])
def test_packed_shard_train_crop_bit_exact(ecm, half):
    rgb, disp, frames = _frames(3, 300, 600, 5)
    ys, xs = [0, 44, 13], [88, 0, 61]
    a = ecm.ops.frame_prep(torch.from_numpy(frames).cuda(), ys, xs, 256, 512, want_image=True)
    d_in = torch.from_numpy(disp).cuda()
    b = ecm.ops.frame_prep((torch.from_numpy(rgb).cuda(), d_in.half() if half else d_in), ys, xs, 256, 512, want_image=True)
    for k in (0, 1, 3):
        assert torch.equal(a[k], b[k]), k                  # colour outputs: bit-identical
    want_d = a[2].half().float() if half else a[2]          # fp16 shard: the fp32 disparity rounded to fp16
    assert torch.equal(b[2], want_d)


def test_packed_shard_eval_pad_bit_exact_vs_oracle(ecm):
    rgb, disp, frames = _frames(2, 540, 960, 6)
    got = ecm.ops.frame_prep((torch.from_numpy(rgb).cuda(), torch.from_numpy(disp).cuda()), [0, 0], [0, 0], 576, 960, split=540,
                             tail=36)
    for b in range(2):
        l, r, d, _ = O.flying3d_sample(frames[b], "test")
        assert torch.equal(got[0][b].cpu(), l) and torch.equal(got[1][b].cpu(), r) and torch.equal(got[2][b].cpu(), d)


@pytest.mark.parametrize("packed", [False, True])
def test_kitti_eval_pad_bit_exact(ecm, g9, packed):
    inp = eval_harness_inputs()
    frame = inp["kitti_frame"]                               # [375,1242,7]
    if packed:
        src = (torch.from_numpy(frame[None, ..., :6].astype(np.uint8)).cuda(), torch.from_numpy(frame[None, ..., 6].copy()).cuda())
    else:
        src = torch.from_numpy(frame[None].copy()).cuda()
    left, right, disp, image = ecm.ops.frame_prep_kitti_eval(src, want_image=True)
    assert left.shape == (1, 3, 384, 1248)
    l, r, d = O.kitti_eval_sample(frame)
    assert torch.equal(left[0].cpu(), l) and torch.equal(right[0].cpu(), r) and torch.equal(disp[0].cpu(), d)
    # and against the reference's own padded array (fixture): disparity zero pattern, subsamples, colour checksums
    dn = disp[0].cpu().numpy()
    assert np.array_equal(np.packbits(dn != 0), g9["pad_disp_nonzero"])
    assert np.array_equal(dn[::3, ::5], g9["pad_disp_sub"])
    raw = image[0].cpu().numpy().transpose(1, 2, 0)          # raw left image, HWC
    assert np.array_equal(raw[::7, ::11].astype(np.uint8), g9["pad_rgb_sub"][..., :3])


def test_eval_step_reproduces_test_py(ecm):
    """One evaluation step as test.py:63-94 runs it -- eval-pad the frames, forward under no_grad, crop [:540,:960],
    EPE of output3 -- on the HIP path end to end (frame prep, model, metric), finite and consistent with torch's own
    masked mean on the device."""
    rgb, disp, _ = _frames(1, 540, 960, 8)
    disp = disp * 0.7
    left, right, gt = ecm.ops.frame_prep((torch.from_numpy(rgb).cuda(), torch.from_numpy(disp).cuda()), [0], [0], 576, 960,
                                         split=540, tail=36)
    torch.manual_seed(0)
    model = ecm.get_model("cmfsm").cuda().eval()
    with torch.no_grad():
        o3 = model(left, right)[2]
    out = ecm.ops.eval_epe(o3, gt).cpu()
    assert torch.isfinite(out).all()
    want = O.sceneflow_eval_epe(o3.cpu(), gt.cpu())
    np.testing.assert_allclose(out[:3].numpy(), np.array(want), rtol=1e-5)
    ecm.ops.check_async_errors()


def test_stereo_loss_and_kitti_metrics_vs_reference_statements(ecm):
    """ops.stereo_loss3 (loss.hip) against the reference's own loss / EPE / 3-px statements (train_kitti.py:186, 196-216,
    executed in the build container: fixture g10)."""
    from conftest import load_golden
    from oracle.weights import seeded
    g = load_golden("g10_kitti_metrics")
    B, H, W = 2, 37, 53
    gt = seeded("g10.gt", B, H, W).abs() * 120.0
    gt[:, ::7, ::5] = 0.0
    outs = [(gt + seeded(f"g10.p{i}", B, H, W) * s).unsqueeze(1).cuda() for i, s in ((1, 4.0), (2, 2.0), (3, 3.0))]
    loss, met = ecm.ops.stereo_loss3(outs, gt.cuda())
    torch.testing.assert_close(loss.cpu(), g["loss"], rtol=1e-5, atol=1e-6)
    assert float(met[1]) == float(g["n_mask"])
    torch.testing.assert_close(met[2].cpu(), g["epe"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(met[3].cpu(), g["loss_3"], rtol=1e-5, atol=1e-4)
