#!/usr/bin/env python3
"""Fixture for the torch-free C++ host of a whole chain (tests/c_host/chain_host.cpp): a flat binary of
named fp32 tensors -- inputs, reference-layout weights and the outputs of the REFERENCE's own modules run in the build
container: classif1 of cmfsm (Conv3d 32->32 + GroupNorm + ReLU + Conv3d 32->1, cmfsm.py:621-634), softmax +
disparityregression (703-706, 111-123), eight_related_context_mapping (431-593), and the 9-neighbour aggregation
(709-723; restated by the oracle, which fixtures g7a / g7q1 pin against the reference's forward).

Blob: magic "ECMBLOB1", u32 count, then per tensor: u32 name length, name bytes, u32 ndim, u32 dims[ndim], float32 data.
Build container only.  Usage: python -B tests/golden/make_golden_chain.py"""
from __future__ import annotations

import os
import struct
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

for name in ("torchvision", "torchvision.models", "cmf.caffe_pb2"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, "/root/reference")
from cmf.models import get_model  # noqa: E402
from oracle import ecm_oracle as O  # noqa: E402
from oracle.weights import seeded, tensor_for  # noqa: E402

ref = sys.modules["cmf.models.cmfsm"]
torch.set_num_threads(8)
model = get_model("cmfsm")
model.load_state_dict({k: tensor_for(k, v.shape) for k, v in model.state_dict().items()})
model.eval()

D, h, w, s = 12, 7, 13, 4                       # ragged on purpose: odd h and w, D not a multiple of 8
x = seeded("chain.x", 1, 32, D, h, w)
lr, hr = seeded("chain.lr", 1, 32, h, w), seeded("chain.hr", 1, 32, s * h, s * w)
with torch.no_grad():
    hid = F.relu(model.classif1[0](x))                                   # convbn_3d + ReLU (cmfsm.py:621-623)
    logits = model.classif1[2](hid)                                      # Conv3d 32 -> 1
    prob = F.softmax(logits.squeeze(1), dim=1)                           # :703-705
    disp = ref.disparityregression(D)(prob)                              # :706
    planes = model.mapping_matrix(lr, hr, None, None)                    # :664
    w9 = torch.cat(planes, 1)
    pred = O.ecm_aggregate_eight(disp, w9, s)                            # :709-723
sd = model.state_dict()
tensors = {
    "x": x, "lr": lr, "hr": hr,
    "conv_w": sd["classif1.0.0.weight"], "gn_gamma": sd["classif1.0.1.weight"], "gn_beta": sd["classif1.0.1.bias"],
    "c1_w": sd["classif1.2.weight"],
    "W0": sd["mapping_matrix.similarity1.conv0.weight"], "W1": sd["mapping_matrix.similarity1.conv1.weight"],
    "W2": sd["mapping_matrix.similarity1.conv2.weight"], "W3": sd["mapping_matrix.similarity1.conv3.weight"],
    "exp_hidden": hid, "exp_logits": logits, "exp_disp": disp, "exp_w9": w9, "exp_pred": pred,
}
out = os.path.join(ROOT, "tests", "golden", "chain_classif_heads.bin")
with open(out, "wb") as f:
    f.write(b"ECMBLOB1")
    f.write(struct.pack("<I", len(tensors)))
    for k, t in tensors.items():
        a = np.ascontiguousarray(t.detach().numpy().astype(np.float32))
        f.write(struct.pack("<I", len(k)))
        f.write(k.encode())
        f.write(struct.pack("<I", a.ndim))
        f.write(struct.pack("<%dI" % a.ndim, *a.shape))
        f.write(a.tobytes())
print("wrote", out, os.path.getsize(out) // 1024, "KiB;", {k: tuple(v.shape) for k, v in tensors.items()})
