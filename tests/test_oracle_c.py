"""The plain-C oracle (oracle/ecm_oracle_c.c, naive loops, double accumulation) against the torch oracle and the golden
vectors produced by the reference: an arithmetic restatement that shares no code with ATen."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import ecm_oracle as O
from oracle.weights import seeded


@pytest.fixture(scope="module")
def oc():
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, "-s"])
    return C.CDLL(os.path.join(d, "libecm_oracle_c.so"))


def P(t):
    return t.contiguous().numpy().ctypes.data_as(C.c_void_p)


def test_c_cost_volume(oc):
    L, R = seeded("g1b.L", 2, 8, 4, 24), seeded("g1b.R", 2, 8, 4, 24)
    out = torch.empty(2, 16, 20, 4, 24)
    oc.oc_cost_volume(P(L), P(R), P(out), 2, 8, 4, 24, 20)
    assert torch.equal(out, load_golden("g1b_costvol")["cost"])


def test_c_ecm_weights_eight(oc, cmfsm_sd):
    lr, hr = seeded("g2.lr", 1, 32, 3, 4), seeded("g2.hr", 1, 32, 12, 16)
    Ws = [cmfsm_sd[f"mapping_matrix.similarity1.conv{i}.weight"].contiguous() for i in range(4)]
    out = torch.empty(1, 9, 12, 16)
    oc.oc_ecm_weights_eight(P(lr), P(hr), *[P(w) for w in Ws], P(out), 1, 3, 4, 4)
    torch.testing.assert_close(out, load_golden("g2_ecm_weights")["w9"], rtol=1e-4, atol=1e-6)


def test_c_soft_argmin_and_aggregate(oc):
    cost = seeded("g4.cost", 2, 48, 6, 10, scale=2.0)
    d = torch.empty(2, 6, 10)
    oc.oc_soft_argmin(P(cost), P(d), 2, 48, 60)
    torch.testing.assert_close(d, load_golden("g4_softargmin")["disp"], rtol=1e-5, atol=1e-5)
    w9 = torch.softmax(seeded("c.w9", 2, 9, 24, 40), 1)
    out = torch.empty(2, 24, 40)
    oc.oc_aggregate9(P(d), P(w9), P(out), 2, 6, 10, 4)
    torch.testing.assert_close(out, O.ecm_aggregate_eight(d, w9, 4)[:, 0], rtol=1e-5, atol=1e-5)


def test_c_hourglass_against_reference_fixture(oc, cmfsm_sd):
    """Chain the C conv / deconv / GroupNorm exactly as hourglass.forward does (cmfsm.py:283-303) and compare with the
    reference module's own output (fixture g6, 'none' variant)."""
    g = load_golden("g6_hourglass")
    sd = {k[len("dres3."):]: v for k, v in cmfsm_sd.items() if k.startswith("dres3.")}
    x = seeded("g6.x", 1, 32, 8, 8, 8)

    def conv(t, key, co, st):
        B, ci, D, H, W = t.shape
        y = torch.empty(B, co, (D - 1) // st + 1, (H - 1) // st + 1, (W - 1) // st + 1)
        oc.oc_conv3d_k3(P(t), P(sd[key]), P(y), B, ci, co, D, H, W, st)
        return y

    def deconv(t, key, co):
        B, ci, D, H, W = t.shape
        y = torch.empty(B, co, 2 * D, 2 * H, 2 * W)
        oc.oc_deconv3d_k3s2(P(t), P(sd[key]), P(y), B, ci, co, D, H, W)
        return y

    def gn(t, key, skip=None, relu=False):
        y = torch.empty_like(t)
        S = t[0, 0].numel()
        oc.oc_group_norm(P(t), P(sd[key + ".weight"]), P(sd[key + ".bias"]), P(skip) if skip is not None else None, P(y),
                         t.shape[0], t.shape[1], C.c_long(S), int(relu))
        return y

    out = gn(conv(x, "conv1.0.0.weight", 64, 2), "conv1.0.1", relu=True)
    pre = gn(conv(out, "conv2.0.weight", 64, 1), "conv2.1", relu=True)
    out = gn(conv(pre, "conv3.0.0.weight", 64, 2), "conv3.0.1", relu=True)
    out = gn(conv(out, "conv4.0.0.weight", 64, 1), "conv4.0.1", relu=True)
    post = gn(deconv(out, "conv5.0.weight", 64), "conv5.1", skip=pre, relu=True)
    out = gn(deconv(post, "conv6.0.weight", 32), "conv6.1")
    torch.testing.assert_close(pre, g["none_pre"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(post, g["none_post"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out, g["none_out"], rtol=1e-3, atol=1e-4)
