"""The reference's CALLERS, as they wrap and drive the model: `nn.DataParallel(model, device_ids=...)`,
`.cuda()`, Adam over `model.parameters()`, the masked smooth-L1 loss of train.py:162-174 in plain torch, a checkpoint dict
`{'epoch','model_state','optimizer_state'}` with `module.`-prefixed keys (train.py:78-86, 228-233) reloaded the way test.py:40-52
does -- and the GroupNorm cluster kernels next to a foreign long-lived kernel that holds CUs (what an RCCL ring looks like to
the scheduler)."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT

pytestmark = pytest.mark.gpu
SPIN_SO = os.path.join(ROOT, "tests", "c_host", "libspin_host.so")


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


def _pair(seed, h=256, w=512):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1, 3, h, w, generator=g).cuda(), torch.randn(1, 3, h, w, generator=g).cuda(),
            (torch.rand(1, h, w, generator=g) * 200.0 - 4.0).cuda())


def _train_py_loss(outputs, disparity):
    """train.py:162-174, verbatim semantics in plain torch (boolean-mask gathers and all)."""
    mask = (disparity < 192) & (disparity > 0)
    o1, o2, o3 = (torch.squeeze(o, 1) for o in outputs)
    return (0.5 * F.smooth_l1_loss(o1[mask], disparity[mask], reduction="mean")
            + 0.7 * F.smooth_l1_loss(o2[mask], disparity[mask], reduction="mean")
            + F.smooth_l1_loss(o3[mask], disparity[mask], reduction="mean"))


def test_dataparallel_train_step_checkpoint_roundtrip(ecm, tmp_path):
    torch.manual_seed(11)
    model = torch.nn.DataParallel(ecm.get_model("cmfsm"), device_ids=[0])          # train.py:78-79 on a 1-GPU box
    model.cuda()                                                                  # train.py:81
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))  # train.py:85-86
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    left, right, gt = _pair(5)
    model.train()
    optimizer.zero_grad()
    outputs = model(left, right)
    assert all(o.shape == (1, 1, 256, 512) for o in outputs)
    loss = _train_py_loss(outputs, gt)
    loss.backward()
    optimizer.step()
    assert torch.isfinite(loss)
    ecm.ops.check_async_errors()
    state = {"epoch": 3, "model_state": model.state_dict(), "optimizer_state": optimizer.state_dict()}    # train.py:228-231
    assert all(k.startswith("module.") for k in state["model_state"])
    assert len(state["model_state"]) == 272
    changed = sum(int(not torch.equal(before[k], v)) for k, v in state["model_state"].items())
    assert changed == 272, f"only {changed} of 272 tensors moved in the step"
    path = str(tmp_path / "3_cmfsm_flying3d_best_model.pkl")
    torch.save(state, path)

    l2, r2, _ = _pair(6)
    model.eval()
    with torch.no_grad():
        want = [o.clone() for o in model(l2, r2)]
    # test.py:38-52: fresh model, wrapped, .cuda(0), load_state_dict(checkpoint['model_state'])
    fresh = torch.nn.DataParallel(ecm.get_model("cmfsm"), device_ids=[0])
    fresh.cuda(0)
    checkpoint = torch.load(path)
    fresh.load_state_dict(checkpoint["model_state"])
    assert checkpoint["epoch"] == 3
    fresh.eval()
    with torch.no_grad():
        got = fresh(l2, r2)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # the optimizer state reloads too (train.py:106-107)
    opt2 = torch.optim.Adam(fresh.parameters(), lr=1e-3, betas=(0.9, 0.999))
    opt2.load_state_dict(checkpoint["optimizer_state"])
    assert len(opt2.state_dict()["state"]) == len(optimizer.state_dict()["state"])
    # an UNWRAPPED model takes the same file after stripping the prefix
    plain = ecm.get_model("cmfsm").cuda().eval()
    plain.load_state_dict({k[len("module."):]: v for k, v in checkpoint["model_state"].items()})
    with torch.no_grad():
        got = plain(l2, r2)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    ecm.ops.check_async_errors()


def test_unsupported_encoder_layer_raises_instead_of_falling_back(ecm):
    import importlib
    mdl = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.models")
    conv = mdl.EncConv2d(32, 32, kernel_size=5, padding=2, bias=False).cuda()      # 5x5: outside the native family
    x = torch.randn(1, 32, 16, 16, device="cuda")
    assert not mdl.ALLOW_FALLBACK
    with pytest.raises(RuntimeError, match="ECM_ALLOW_FALLBACK"):
        conv(x)
    mdl.ALLOW_FALLBACK = True
    try:
        n = len(mdl.SLOW_PATH_EVENTS)
        y = conv(x)
        assert y.shape == x.shape and len(mdl.SLOW_PATH_EVENTS) == n + 1
    finally:
        mdl.ALLOW_FALLBACK = False


def test_no_slow_path_in_the_registered_models_at_bench_sizes(ecm):
    """cmfsm at the SceneFlow and KITTI frame sizes never takes a slower-than-designed path (nothing lands in
    models.SLOW_PATH_EVENTS)."""
    import importlib
    mdl = importlib.import_module("explicit-context-mapping-for-stereo-matching_amd.models")
    n = len(mdl.SLOW_PATH_EVENTS)
    model = ecm.get_model("cmfsm").cuda().eval()
    with torch.no_grad():
        for h, w in ((576, 960), (384, 1248)):
            model(torch.randn(1, 3, h, w, device="cuda"), torch.randn(1, 3, h, w, device="cuda"))
    assert len(mdl.SLOW_PATH_EVENTS) == n, mdl.SLOW_PATH_EVENTS[n:]


@pytest.mark.parametrize("wgs,lds", [(48, 0), (64, 96 * 1024), (256, 0)])
def test_groupnorm_clusters_next_to_a_foreign_long_lived_kernel(ecm, wgs, lds):
    """A spin kernel holds `wgs` workgroups (with `lds` bytes of LDS each: whole CUs when large) for ~60 ms on a side stream
    while the ticket-cluster GroupNorm kernels -- whose members wait for each other across workgroups -- run forward and
    backward on the main stream, repeatedly.  They must finish (not run into their bounded waits), stay finite and leave no
    asynchronous error; results equal the undisturbed run bit for bit."""
    if not os.path.exists(SPIN_SO):
        pytest.skip("tests/c_host/libspin_host.so not built (run __graft_entry__.build())")
    spin = ctypes.CDLL(SPIN_SO)
    spin.spin_launch.restype = ctypes.c_int
    spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    ops = ecm.ops
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(2, 32, 24, 144, 240, device="cuda", generator=g)
    gm, bt = torch.rand(32, device="cuda", generator=g) + 0.5, torch.randn(32, device="cuda", generator=g)
    gy = torch.randn(x.shape, device="cuda", generator=g)

    def run():
        xg = x.clone().requires_grad_()
        y = ops.group_norm_act(xg, gm, bt, None, True)
        y.backward(gy)
        return y.detach(), xg.grad

    y0, g0 = run()
    torch.cuda.synchronize()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        rc = spin.spin_launch(wgs, lds, 60.0, sink.data_ptr(), side.cuda_stream)
    assert rc == 0, rc
    t0.record()
    outs = [run() for _ in range(8)]                         # ~8 x (fwd 0.2 ms + bwd 0.3 ms): all inside the spin's 60 ms
    t1.record()
    torch.cuda.synchronize()
    ops.check_async_errors()
    assert t0.elapsed_time(t1) < 1500.0, "the cluster launches ran into their bounded waits"
    for y, gx in outs:
        assert torch.isfinite(y).all() and torch.isfinite(gx).all()
        assert torch.equal(y, y0) and torch.equal(gx, g0)
