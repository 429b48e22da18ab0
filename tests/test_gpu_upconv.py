"""GPU parity of the composed up-sampling stage (round 6; csrc/kernels_upconv.hip): vp_op_upconv against torch fp64 evaluating the reference's
three-op form (ConvTranspose2d k2 s2 [+ 1x1 skip link] -> Conv2d 3x3 -> GELU; scene_neck.py:29-35, scene_seg_head.py:24-29,35-38) -- the real
stage geometries of the networks (channel counts, skip widths, map sizes incl. the ragged 10x20 / 20x40 / 40x80 ones), both kernel shapes, K slices."""
import numpy as np
import pytest

from test_upconv_emulated import composed_fp64, make_stage, three_op_fp64

pytestmark = pytest.mark.gpu


def _run(lib, seed, cin, cm, cout, cs, h, w, act, cfgs, tol=2e-5, precision="fp16x3"):
    t = make_stage(np.random.default_rng(seed), cin, cm, cout, cs, h, w)
    ref = three_op_fp64(t, act)
    outs = []
    for shape, nsplit in cfgs:
        got = lib.op_upconv(t["x"], t["wt"], t["bt"], t["w3"], t["b3"], skip=t["skip"], ws=t["ws"], bs=t["bs"], act=act, shape=shape, nsplit=nsplit,
                            precision=precision)
        err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
        assert err <= tol, f"shape {shape} nsplit {nsplit}: err {err:.3e}"
        outs.append(got)
    return outs


def test_compose_on_the_device():
    from autoware_vision_pilot_amd import lib

    t = make_stage(np.random.default_rng(7), 70, 90, 50, 24, 5, 7)
    wx, wsk, bias = lib.compose_upconv(t["wt"], t["bt"], t["w3"], t["b3"], ws=t["ws"], bs=t["bs"])
    ref = three_op_fp64(t, act=0)
    assert np.abs(composed_fp64(t, wx, wsk, bias) - ref).max() <= 1e-12 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("name,cin,cm,cout,cs,h,w", [
    ("neck0", 1280, 1280, 768, 80, 10, 20), ("neck1", 768, 768, 512, 40, 20, 40), ("neck2", 512, 512, 512, 24, 40, 80),
    ("head3", 256, 256, 256, 32, 80, 160), ("ego0", 1456, 1456, 768, 80, 10, 20)])
def test_stage_geometries(name, cin, cm, cout, cs, h, w):
    from autoware_vision_pilot_amd import lib

    big = h * w >= 3200
    outs = _run(lib, hash(name) % 1000, cin, cm, cout, cs, h, w, 1, [(6, 1), (7, 1)] + ([] if big else [(6, 4), (7, 3), (-1, 0)]))
    assert np.array_equal(outs[0], outs[1])      # the two shapes walk the same K steps in the same order


def test_stage_head4_no_skip():
    from autoware_vision_pilot_amd import lib

    outs = _run(lib, 44, 128, 128, 128, 0, 160, 320, 1, [(6, 1), (7, 1)])
    assert np.array_equal(outs[0], outs[1])


def test_small_and_ragged():
    from autoware_vision_pilot_amd import lib

    _run(lib, 51, 96, 40, 72, 0, 9, 21, 0, [(6, 1), (7, 1), (7, 2)])
    _run(lib, 52, 64, 64, 200, 40, 10, 20, 1, [(6, 1), (7, 1), (6, 3), (7, 5)])
    _run(lib, 53, 40, 24, 128, 33, 17, 18, 0, [(6, 1), (7, 1), (7, 10)])
    _run(lib, 54, 32, 32, 128, 0, 1, 1, 1, [(6, 1), (7, 1)])


@pytest.mark.parametrize("name,cin,cm,cout,cs,h,w", [
    ("neck0", 1280, 1280, 768, 80, 10, 20), ("neck1", 768, 768, 512, 40, 20, 40), ("neck2", 512, 512, 512, 24, 40, 80),
    ("head3", 256, 256, 256, 32, 80, 160), ("head4", 128, 128, 128, 0, 160, 320)])
def test_stage_geometries_fp16_form(name, cin, cm, cout, cs, h, w):
    """the VP_FP16 engines' form of the kernel (64-channel chunks, one MFMA per product) on the networks' stage geometries: the fp16 engines'
    regression class (operands rounded to fp16: 2^-11 each), not the parity bar; skip widths 96 (one full + one half-dead chunk), 64, 32, 32, none"""
    from autoware_vision_pilot_amd import lib

    big = h * w >= 3200
    outs = _run(lib, hash(name) % 1000, cin, cm, cout, cs, h, w, 1, [(6, 1), (7, 1)] + ([] if big else [(6, 4), (7, 3), (-1, 0)]), tol=8e-3, precision="fp16")
    assert np.array_equal(outs[0], outs[1])      # the two shapes walk the same K steps in the same order
