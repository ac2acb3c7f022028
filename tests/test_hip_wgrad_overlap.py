"""Weight gradients on a side stream (ops._on_side): same gradients bit for bit, and the GroupNorm cluster kernels keep working
while another stream's kernels hold CUs (the ticket counter of gn3d.hip used to stay un-reset when a workgroup was placed
only after all tickets had gone -- found through exactly this configuration)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecm():
    assert torch.cuda.is_available()
    import ecm_amd
    return ecm_amd


def _step(ecm, model, left, right, gt):
    from importlib import import_module
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    for p in model.parameters():
        p.grad = None
    loss = D.masked_smooth_l1_x3(model(left, right), gt)
    loss.backward()
    torch.cuda.synchronize()
    assert ecm.ops._lib.query("ecm_async_status", 1) == 0, "a GroupNorm cluster launch timed out"
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


def test_side_stream_weight_gradients_are_bit_identical(ecm):
    ops = ecm.ops
    torch.manual_seed(3)
    model = ecm.get_model("cmfsm").cuda().train()
    B, H, W = 2, 576, 960
    left, right = torch.randn(B, 3, H, W, device="cuda"), torch.randn(B, 3, H, W, device="cuda")
    gt = torch.rand(B, H, W, device="cuda") * 191
    prev = ops.enable_wgrad_overlap(False)
    try:
        ref = _step(ecm, model, left, right, gt)
        ops.enable_wgrad_overlap(True)
        if not ops.WGRAD_OVERLAP:
            pytest.skip("ECM_WGRAD_OVERLAP=0 in the environment")
        for _ in range(3):                                     # several passes: the side stream's queue builds up
            got = _step(ecm, model, left, right, gt)
        assert ops._SIDE, "no weight gradient went to the side stream"
        assert got.keys() == ref.keys()
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
    finally:
        ops.enable_wgrad_overlap(prev)


def test_non_leaf_and_hooked_weights_stay_on_the_main_stream(ecm):
    """A weight that is the output of another op (its gradient flows on during the pass) or carries a tensor hook must not be
    computed on the side stream."""
    ops = ecm.ops
    prev = ops.enable_wgrad_overlap(True)
    try:
        if not ops.WGRAD_OVERLAP:
            pytest.skip("ECM_WGRAD_OVERLAP=0 in the environment")
        ops._SIDE.clear()
        x = torch.randn(1, 32, 8, 16, 32, device="cuda", requires_grad=True)
        base = (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05).requires_grad_()
        w = base * 2.0                                         # non-leaf
        ops.conv3d_k3(x, w, 1).sum().backward()
        assert not ops._SIDE
        seen = []
        leaf = (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05).requires_grad_()
        leaf.register_hook(lambda g: seen.append(float(g.abs().sum())))
        ops.conv3d_k3(x, leaf, 1).sum().backward()
        assert not ops._SIDE and len(seen) == 1
        ref = torch.autograd.grad(torch.nn.functional.conv3d(x, leaf.detach().requires_grad_(), padding=1).sum(), x)[0]
        plain = (leaf.detach().clone()).requires_grad_()
        ops.conv3d_k3(x, plain, 1).sum().backward()            # leaf, no hook: side stream
        assert ops._SIDE
        torch.cuda.synchronize()
        assert torch.allclose(plain.grad, leaf.grad, rtol=0, atol=0)
        assert ref.shape == x.shape
    finally:
        ops.enable_wgrad_overlap(prev)


def test_layer_used_twice_accumulates_correctly(ecm):
    """The reference calls the shared encoder once per image (cmfsm.py:657-658): a weight then receives two gradients in one
    pass and AccumulateGrad adds them on the main stream -- the second use must not leave either operand of that addition in
    flight on the side stream."""
    ops = ecm.ops
    prev = ops.enable_wgrad_overlap(False)
    try:
        torch.manual_seed(5)
        w = (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05).requires_grad_()
        xs = [torch.randn(2, 32, 24, 72, 120, device="cuda") for _ in range(3)]

        def run():
            w.grad = None
            sum(ops.conv3d_k3(x, w, 1).square().sum() for x in xs).backward()
            torch.cuda.synchronize()
            return w.grad.clone()
        ref = run()
        ops.enable_wgrad_overlap(True)
        if not ops.WGRAD_OVERLAP:
            pytest.skip("ECM_WGRAD_OVERLAP=0 in the environment")
        for _ in range(3):
            assert torch.equal(run(), ref)
    finally:
        ops.enable_wgrad_overlap(prev)


def test_gradients_are_complete_on_the_current_stream_right_after_backward(ecm):
    """The join itself: the autograd-engine callback queued by ops._on_side must make the CURRENT stream wait for
    the side stream at the end of backward.  Here nothing synchronises the device between backward() and the readers: the
    gradients are cloned / consumed by an optimizer step on the current stream straight away, with a long queue on the side
    stream (several big weight-gradient kernels behind each other), and must equal the no-overlap run bit for bit."""
    ops = ecm.ops
    torch.manual_seed(9)
    ws = [(torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05).requires_grad_() for _ in range(6)]
    x0 = torch.randn(2, 32, 24, 72, 120, device="cuda")

    def run(read):
        for w in ws:
            w.grad = None
        x = x0
        for w in ws:                                           # a chain: backward launches six weight gradients back to back
            x = ops.conv3d_k3(x, w, 1)
        x.square().mean().backward()
        return read()                                          # NO torch.cuda.synchronize() before the read

    clone = lambda: [w.grad.clone() for w in ws]
    prev = ops.enable_wgrad_overlap(False)
    try:
        ref = run(clone)
        torch.cuda.synchronize()
        ops.enable_wgrad_overlap(True)
        if not ops.WGRAD_OVERLAP:
            pytest.skip("ECM_WGRAD_OVERLAP=0 in the environment")
        ops._SIDE.clear()
        for _ in range(3):
            got = run(clone)
            assert ops._SIDE, "no weight gradient went to the side stream"
            for a, b in zip(got, ref):
                assert torch.equal(a, b)
        # an optimizer step right behind backward (the plain-optimizer path under ECM_WGRAD_OVERLAP=1)
        start = [w.detach().clone() for w in ws]

        def sgd():
            with torch.no_grad():
                out = [w - 0.1 * w.grad for w in ws]
            return out
        after = run(sgd)
        for a, s0, g in zip(after, start, ref):
            assert torch.equal(a, s0 - 0.1 * g)
    finally:
        ops.enable_wgrad_overlap(prev)


def test_create_graph_and_odd_layouts_stay_on_the_main_stream(ecm):
    """backward(create_graph=True) makes AccumulateGrad CLONE the gradient on the main stream (no join against a side-stream
    writer), and so does a weight whose layout is outside its stealing contract: both must compute on the main stream."""
    ops = ecm.ops
    prev = ops.enable_wgrad_overlap(True)
    try:
        if not ops.WGRAD_OVERLAP:
            pytest.skip("ECM_WGRAD_OVERLAP=0 in the environment")
        x = torch.randn(1, 32, 8, 16, 32, device="cuda", requires_grad=True)
        w = (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05).requires_grad_()
        ops._SIDE.clear()
        ops.conv3d_k3(x, w, 1).sum().backward(create_graph=True)
        assert not ops._SIDE
        wt = (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05).permute(1, 0, 2, 3, 4).requires_grad_()     # non-contiguous leaf
        assert wt.is_leaf and not wt.is_contiguous()
        ops.conv3d_k3(x, wt, 1).sum().backward()
        assert not ops._SIDE
        torch.cuda.synchronize()
        ref = torch.autograd.grad(torch.nn.functional.conv3d(x, wt.detach(), padding=1).sum(), x)[0]
        assert ref.shape == x.shape
    finally:
        ops.enable_wgrad_overlap(prev)


def test_free_running_steps_do_not_grow_the_allocator(ecm):
    """A loop that never synchronises: the host enqueues steps several times faster than the device runs them.  The operands of
    the side-stream weight gradients are recorded on that stream, so without the host run-ahead bound (ops.pace_side_streams)
    every step found the previous steps' blocks still pending and went to the driver for new memory (round 4: 320 device
    allocations, 40.7 -> 137.5 GiB over 15 steps of the benchmark).  With it the pool stays where the first steps left it."""
    from importlib import import_module
    D = import_module("explicit-context-mapping-for-stereo-matching_amd.dist")
    ops = ecm.ops
    if not ops._WGRAD_PACE:
        pytest.skip("ECM_WGRAD_PACE=0 in the environment")
    torch.manual_seed(5)
    model = ecm.get_model("cmfsm").cuda().train()
    prev = ops.WGRAD_OVERLAP
    ddp = D.FlatBucketDDP(model, 1)                            # turns the side stream on
    if not ops.WGRAD_OVERLAP:
        pytest.skip("ECM_WGRAD_OVERLAP=0 in the environment")
    opt = torch.optim.Adam(ddp.params, lr=1e-4, fused=True)
    B, H, W = 2, 576, 960                                      # device-bound: the host enqueues a step ~3x faster than it runs
    left, right = torch.randn(B, 3, H, W, device="cuda"), torch.randn(B, 3, H, W, device="cuda")
    gt = torch.rand(B, H, W, device="cuda") * 191

    def step():
        ddp.zero_grad()
        loss, count = D.masked_smooth_l1_x3_with_count(model(left, right), gt, 192)
        ddp.global_mean_loss(loss, count).backward()
        ddp.allreduce_gradients()
        opt.step()

    try:
        for _ in range(3):                                     # free-running from the start: the pool settles in the loop's own pattern
            step()
        torch.cuda.synchronize()
        assert ops._SIDE and any(st[3] is not None for st in ops._SIDE.values()), "no join event behind the side stream's work"
        before = torch.cuda.memory_stats()
        for _ in range(8):                                     # no synchronisation in here
            step()
        torch.cuda.synchronize()
        after = torch.cuda.memory_stats()
        # steady state: un-paced, every step of this loop went to the driver dozens of times and the pool grew by a working set
        # per step (an odd block may still be requested while the pool settles)
        grown = after["num_device_alloc"] - before["num_device_alloc"]
        r0, r1 = before["reserved_bytes.all.current"], after["reserved_bytes.all.current"]
        assert grown <= 4 and r1 <= 1.15 * r0, (grown, r0 / 2**30, r1 / 2**30)     # (un-paced: ~90 allocations, 2x the pool)
        assert ops._lib.query("ecm_async_status", 1) == 0
    finally:
        ops.enable_wgrad_overlap(prev)
        ops._SIDE.clear()
