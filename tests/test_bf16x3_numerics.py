"""The numerical claim behind the split-bf16 experiment (tools/experiments/README.md, tools/micro/mfma_bf16x3.hip):
three bf16 terms per operand and six products accumulated in fp32 are as close to the fp64 truth as a plain fp32 dot product
at the reduction lengths of the 3x3x3 convolutions -- i.e. the split costs no accuracy; what decides the experiment is the
measured instruction rate.  CPU emulation (every product of two bf16 values is exact in fp32)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bf16x3_numerics", os.path.join(ROOT, "tools", "experiments", "bf16x3_numerics.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


@pytest.mark.parametrize("K", [288, 864, 1728])
def test_six_product_split_is_fp32_equivalent(K):
    e32, e6, rep = B.errors(K, n=128, seed=K)
    assert rep == 0.0                      # three bf16 terms represent every fp32 value of this range exactly
    assert e6 <= 1.5 * e32 + 1e-8, (K, e32, e6)
    _, e3, _ = B.errors(K, n=128, seed=K, terms=3)
    assert 3 * e32 < e3 < 1e-4             # three products only: ~16 bits, inside the per-stage tolerance but NOT fp32-equivalent
