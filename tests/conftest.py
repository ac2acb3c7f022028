import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def cmfsm_shapes():
    with open(os.path.join(GOLDEN, "cmfsm_state_shapes.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def cmfsm_sd(cmfsm_shapes):
    from oracle.weights import make_state_dict
    return make_state_dict(cmfsm_shapes)
