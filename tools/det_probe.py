import os, sys, torch
sys.path.insert(0, os.getcwd())
import ecm_amd
model = ecm_amd.get_model("cmfsm").cuda().eval()
g = torch.Generator(device="cuda").manual_seed(6)
left = torch.randn(1, 3, 384, 1248, device="cuda", generator=g)
right = torch.randn(1, 3, 384, 1248, device="cuda", generator=g)
with torch.no_grad():
    f1 = model.feature_extraction(left); f2 = model.feature_extraction(left)
    print("encoder deterministic:", [bool(torch.equal(a, b)) for a, b in zip(f1, f2)], [float((a-b).abs().max()) for a,b in zip(f1,f2)])
    lr_l, _, hr_l = f1
    lr_r = model.feature_extraction(right)[0]
    a = model.hot_path(lr_l, hr_l, lr_r); b = model.hot_path(lr_l, hr_l, lr_r)
    print("hot path deterministic:", [bool(torch.equal(p, q)) for p, q in zip(a, b)], [float((p-q).abs().max()) for p,q in zip(a,b)])
    ops = ecm_amd.ops
    x = torch.randn(1, 32, 48, 96, 312, device="cuda"); w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05
    ys = [ops.conv3d_k3(x, w, 1) for _ in range(4)]
    print("conv det:", [bool(torch.equal(ys[0], y)) for y in ys[1:]])
    w9a = model.mapping_matrix.weights(lr_l, hr_l); w9b = model.mapping_matrix.weights(lr_l, hr_l)
    print("ecm weights det:", bool(torch.equal(w9a, w9b)))
    c = ops.cost_volume(lr_l, lr_r, 48)
    g1 = ops.group_norm_act(c[:, :32].contiguous(), torch.ones(32, device="cuda"), torch.zeros(32, device="cuda"), None, True)
    g2 = ops.group_norm_act(c[:, :32].contiguous(), torch.ones(32, device="cuda"), torch.zeros(32, device="cuda"), None, True)
    print("gn det:", bool(torch.equal(g1, g2)))
    xd = torch.randn(1, 64, 12, 24, 78, device="cuda"); wd = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
    d1 = ops.deconv3d_k3s2(xd, wd); d2 = ops.deconv3d_k3s2(xd, wd)
    print("deconv det:", bool(torch.equal(d1, d2)))
    xs = torch.randn(1, 64, 12, 24, 78, device="cuda"); ws = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
    s1 = ops.conv3d_k3(xs, ws, 1); s2 = ops.conv3d_k3(xs, ws, 1)
    print("small conv det:", bool(torch.equal(s1, s2)))
    s1 = ops.conv3d_k3(xs, ws, 2); s2 = ops.conv3d_k3(xs, ws, 2)
    print("small conv s2 det:", bool(torch.equal(s1, s2)))
    wc = torch.randn(1, 32, 3, 3, 3, device="cuda") * 0.05
    c1 = ops.conv3d_k3(x, wc, 1); c2 = ops.conv3d_k3(x, wc, 1)
    print("c1 det:", bool(torch.equal(c1, c2)))
