"""CPU checks of bench.py's bookkeeping helpers (no GPU, no library): the kernel-family table of `step_breakdown_ms` / the launch
table, and the argument parser's defaults the driver relies on (`python bench.py` with no flags = one GPU, a few steps)."""
import importlib.util
import os
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("ecm_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kernel_families():
    b = _bench()
    f = b._family
    assert f("ecm_conv_wino_fwd", (4, 32, 32, 48, 144, 240, 3)) == "winograd_conv_3d"
    assert f("ecm_conv_wino_fwd_add", (8, 64, 64, 1, 144, 240, 1)) == "winograd_conv_2d"
    assert f("ecm_conv_wino_wgrad", (4, 32, 32, 48, 144, 240, 3)) == "winograd_wgrad_3d"       # the scratch size rides in .longs, not here
    assert f("ecm_conv_wino_wgrad", (8, 64, 64, 1, 144, 240, 1)) == "winograd_wgrad_2d"
    assert f("ecm_conv3d_k3_fwd", (4, 32, 64, 48, 144, 240, 2)) == "stride2_conv_deconv_wgrad_3d"
    assert f("ecm_conv3d_k3_wgrad", (4, 32, 32, 48, 144, 240, 1)) == "direct_conv_3d_stride1"
    assert f("ecm_deconv3d_k3s2_fwd", (4, 64, 32, 24, 72, 120, 48, 144, 240)) == "stride2_conv_deconv_wgrad_3d"
    assert f("ecm_gn3d_bwd_p", (4, 32, 1)) == "groupnorm" and f("ecm_gn3d_stats", (4, 32)) == "groupnorm"
    assert f("ecm_conv3d_c1_gn_fwd", (4, 32, 48, 144, 240)) == "classifier_32to1"
    assert f("ecm_weights9_bwd", (4, 144, 240, 4)) == "ecm_weights"
    assert f("ecm_conv2d_fwd_ex", (8, 3, 32)) == "conv2d_direct_family"
    assert f("ecm_conv_wino_pack_weight2", (32, 32, 3)) == "weight_packing"
    assert f("ecm_costvol_conv_assemble_fwd", (4, 32, 48, 144, 240)) == "cost_volume_assemble"
    assert f("ecm_stereo_loss_fwd", ()) == "heads_and_loss" and f("ecm_sum_n", ()) == "gradient_sums"
    assert f("ecm_something_new", (1,)) == "other_hip"


def test_default_arguments_are_the_single_gpu_headline(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.mode == "train" and (a.batch, a.height, a.width, a.maxdisp) == (4, 576, 960, 192)
    assert 1 <= a.steps <= 20 and 1 <= a.warmup <= 5              # finishes within minutes with no flags
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "50", "--warmup", "10"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 50, 10)
