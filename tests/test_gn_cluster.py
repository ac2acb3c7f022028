"""Robustness of the one-pass GroupNorm cluster kernels (csrc/gn3d.hip; reference op: nn.GroupNorm of convbn_3d,
cmfsm.py:58): workgroups of one launch wait for each other, so the tests cover what happens when the launch does NOT have
the device to itself -- two streams issuing GroupNorm concurrently -- and what happens when a cluster can never complete
(fault injection): a clean error (ECM_EASYNC -> RuntimeError), never a hang and never silent NaN with rc 0."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


def _case(seed, shape):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(shape, device="cuda", generator=g) * 1.3 + 0.1
    gm = torch.rand(shape[1], device="cuda", generator=g) + 0.5
    bt = torch.randn(shape[1], device="cuda", generator=g) * 0.1
    return x, gm, bt


def test_groupnorm_two_streams_concurrently(ecm):
    """GroupNorm forward + backward issued back to back on two streams of ONE device (each launch sized for the whole
    device): every result must equal F.group_norm, and no cluster wait may expire."""
    shape = (2, 32, 24, 144, 240)                      # clusters of 26 / 41 workgroups per span
    cases = [_case(100 + i, shape) for i in range(2)]
    refs = []
    for x, gm, bt in cases:
        xr = x.clone().requires_grad_()
        r = F.relu(F.group_norm(xr, 32, gm, bt, 1e-5))
        r.backward(torch.ones_like(r))
        refs.append((r.detach(), xr.grad))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = [[], []]
    for rep in range(6):                               # interleave the issue order so the launches overlap on the device
        for i in (0, 1):
            x, gm, bt = cases[i]
            with torch.cuda.stream(streams[i]):
                xg = x.clone().requires_grad_()
                y = ecm.ops.group_norm_act(xg, gm, bt, None, True)
                y.backward(torch.ones_like(y))
                outs[i].append((y.detach(), xg.grad))
    ecm.ops.check_async_errors()                       # synchronises; raises if any bounded wait expired
    for i in (0, 1):
        for y, gx in outs[i]:
            torch.testing.assert_close(y, refs[i][0], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(gx, refs[i][1], rtol=1e-3, atol=1e-4)


def test_groupnorm_two_stage_mode_matches(ecm):
    """ecm_gn3d_cluster_mode(0) routes to the two-stage kernels (no inter-workgroup waits): same values."""
    x, gm, bt = _case(7, (1, 64, 8, 48, 96))
    y1 = ecm.ops.group_norm_act(x, gm, bt, None, True)
    old = ecm.ops.gn_cluster_mode(0)
    try:
        y0 = ecm.ops.group_norm_act(x, gm, bt, None, True)
    finally:
        ecm.ops.gn_cluster_mode(old)
    torch.testing.assert_close(y0, y1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(y1, F.relu(F.group_norm(x, 32, gm, bt, 1e-5)), rtol=1e-4, atol=1e-5)


def test_cluster_timeout_is_a_clean_error(ecm):
    """Fault injection (mode 3: the grid is one workgroup short of a cluster, so no cluster can ever complete): the launch
    returns, its outputs are NaN, the sticky error word is set, the NEXT GroupNorm call raises, check_async_errors raises
    and clears, and the path works again afterwards."""
    lib = ecm._lib
    x, gm, bt = _case(9, (1, 32, 8, 48, 96))           # 9216 float4 per channel -> 2 workgroups per cluster
    old_mode = ecm.ops.gn_cluster_mode(3)
    old_poll = lib.query("ecm_gn3d_poll_ms", 50)
    try:
        y = ecm.ops.group_norm_act(x, gm, bt, None, False)
        torch.cuda.synchronize()
        assert torch.isnan(y).any(), "a cluster that never completed must poison its outputs"
        assert lib.query("ecm_async_status", 0) == -4
        with pytest.raises(RuntimeError, match="timed out"):
            ecm.ops.group_norm_act(x, gm, bt, None, False)
        with pytest.raises(RuntimeError, match="timed out"):
            ecm.ops.check_async_errors()
        assert lib.query("ecm_async_status", 0) == 0   # cleared by the check
    finally:
        lib.query("ecm_async_status", 1)
        ecm.ops.gn_cluster_mode(old_mode)
        lib.query("ecm_gn3d_poll_ms", old_poll)
    y = ecm.ops.group_norm_act(x, gm, bt, None, False)
    torch.testing.assert_close(y, F.group_norm(x, 32, gm, bt, 1e-5), rtol=1e-4, atol=1e-5)
    ecm.ops.check_async_errors()


@pytest.mark.parametrize("shape", [(2, 32, 24, 144, 240), (8, 64, 144, 240), (1, 32, 8, 48, 96), (3, 128, 36, 60)])
def test_exchange_memory_is_handed_back_preset(ecm, shape):
    """The one-pass kernels restore their exchange memory (slots, ticket counter, per-span counters) to the all-ones preset
    before they finish, so the copy `ops` keeps per (device, stream) needs no memset between launches: after forward and
    backward, repeated, every byte of it is 0xFF again and the results stay correct."""
    ops = ecm.ops
    x, gm, bt = _case(sum(shape), shape)
    ref = F.relu(F.group_norm(x, 32, gm, bt, 1e-5))
    for rep in range(3):
        xg = x.clone().requires_grad_()
        y = ops.group_norm_act(xg, gm, bt, None, True)
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        torch.testing.assert_close(y.detach(), ref, rtol=1e-4, atol=1e-5)
        assert ops._GN_CLUSTER, "the cluster path keeps its exchange memory"
        for buf in ops._GN_CLUSTER.values():
            assert bool((buf == 0xFF).all()), f"exchange memory not restored after repetition {rep}"
    ops.check_async_errors()


def test_two_captured_graphs_replayed_in_any_order(ecm):
    """the GroupNorm exchange buffer used to be cached per (device, stream) during capture too -- every
    torch.cuda.graph capture runs on the same stream, so a second graph found the first graph's buffer and captured no
    preset: replaying it FIRST ran the cluster kernels on uninitialised memory (2 s time-outs, NaN, sticky ECM_EASYNC).
    Captured GroupNorms no longer touch the kept buffer (stateless entry points), and a capturing stream gets the two-stage
    kernels, so that two graphs replayed CONCURRENTLY cannot starve each other's clusters either."""
    from importlib import import_module
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    torch.manual_seed(4)
    model = ecm.get_model("cmfsm").cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    l1, r1 = (torch.randn(1, 3, 256, 512, device="cuda", generator=g) for _ in range(2))
    l2, r2 = (torch.randn(1, 3, 256, 256, device="cuda", generator=g) for _ in range(2))
    old = ecm.ops.gn_cluster_mode(0)                      # a captured GroupNorm runs the two-stage kernels: like against like
    try:
        with torch.no_grad():
            want1 = [o.clone() for o in model(l1, r1)]
            want2 = [o.clone() for o in model(l2, r2)]
    finally:
        ecm.ops.gn_cluster_mode(old)
    ecm.ops._GN_CLUSTER.clear()                           # as in a fresh process: nothing preset before the captures
    g1 = D.GraphedForward(model, l1, r1, warmup=1)
    print("graph 1 captured", flush=True)
    g2 = D.GraphedForward(model, l2, r2, warmup=1)
    print("graph 2 captured", flush=True)
    out2 = [o.clone() for o in g2(l2, r2)]                # the SECOND graph first
    out1 = [o.clone() for o in g1(l1, r1)]
    torch.cuda.synchronize()
    ecm.ops.check_async_errors()
    for a, b in zip(out1 + out2, want1 + want2):
        assert torch.isfinite(a).all()
        torch.testing.assert_close(a, b, rtol=0, atol=1e-4)
    print("ordered replays ok", flush=True)
    # two graphs replayed concurrently on different streams share nothing
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for s in (s1, s2):
        s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        g1.graph.replay()
    with torch.cuda.stream(s2):
        g2.graph.replay()
    torch.cuda.synchronize()
    ecm.ops.check_async_errors()
    for a, b in zip(list(g1.out) + list(g2.out), want1 + want2):
        torch.testing.assert_close(a, b, rtol=0, atol=1e-4)


def test_two_eager_forwards_of_different_shapes_on_two_streams(ecm):
    """Two model forwards of DIFFERENT sizes issued on two streams of one process (cluster launches of different geometry in
    flight together): this is what starved the clusters in round 4 before cluster launches were ordered across streams on
    the host.  Must finish far inside the bounded waits, equal the one-stream results, and leave no asynchronous error."""
    import time
    torch.manual_seed(4)
    model = ecm.get_model("cmfsm").cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(2)
    a = [torch.randn(1, 3, 256, 512, device="cuda", generator=g) for _ in range(2)]
    b = [torch.randn(1, 3, 256, 256, device="cuda", generator=g) for _ in range(2)]
    with torch.no_grad():
        want_a, want_b = model(*a)[2].clone(), model(*b)[2].clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    with torch.no_grad():
        for _ in range(3):
            with torch.cuda.stream(s1):
                oa = model(*a)[2]
            with torch.cuda.stream(s2):
                ob = model(*b)[2]
            outs.append((oa, ob))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ecm.ops.check_async_errors()
    assert dt < 20.0, f"{dt:.1f} s for six small forwards: cluster launches ran into their bounded waits"
    for oa, ob in outs:
        torch.testing.assert_close(oa, want_a, rtol=0, atol=1e-4)
        torch.testing.assert_close(ob, want_b, rtol=0, atol=1e-4)
