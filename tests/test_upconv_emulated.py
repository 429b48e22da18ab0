"""The composed up-sampling stage (round 6; csrc/kernels_upconv.hip, engine_upconv.cpp) executed on the CPU through the HIP-on-CPU shim of
tests/emul: ConvTranspose2d(k2, s2) [+ Conv2d 1x1 of a skip tensor] -> Conv2d 3x3 (+ GELU) multiplied out at load and run as ONE launch on the
low-resolution tensor.  The checker is torch in fp64 evaluating the reference's THREE-op form (scene_neck.py:29-35, scene_seg_head.py:24-29,35-38):
  * vp_compose_upconv: the composed weights, applied by torch as a 4x4 / stride-2 / pad-1 transposed convolution + a 3x3 convolution of the skip
    tensor + the border-class bias table, equal the three-op form to fp64 rounding on odd sizes, all four borders, 1-pixel maps;
  * vp_op_upconv: both kernel shapes, one and several input chunks, skip tensors of one and two chunks (every pixel class / tap list), ragged
    patches, K slices (fp32 partials + finish kernel), GELU and none, at the parity mode's operator tolerance."""
import ctypes as ct
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))


@pytest.fixture(scope="module")
def emu_lib():
    import build as emul_build

    from autoware_vision_pilot_amd import lib

    if not os.path.exists(emul_build.CLANG):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    so = ct.CDLL(emul_build.build(), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name, (res, args) in lib._SIGS.items():
        fn = getattr(so, name)
        fn.restype, fn.argtypes = res, args
    saved = lib._lib
    lib._lib = so
    yield lib
    lib._lib = saved


def make_stage(rng, cin, cm, cout, cs, h, w):
    f = np.float32
    t = dict(x=rng.standard_normal((cin, h, w), dtype=f),
             wt=rng.standard_normal((cin, cm, 2, 2), dtype=f) * f(np.sqrt(1.0 / cin)), bt=rng.standard_normal(cm, dtype=f) * f(0.1),
             w3=rng.standard_normal((cout, cm, 3, 3), dtype=f) * f(np.sqrt(2.0 / (9 * cm))), b3=rng.standard_normal(cout, dtype=f) * f(0.1),
             skip=None, ws=None, bs=None)
    if cs:
        t.update(skip=rng.standard_normal((cs, 2 * h, 2 * w), dtype=f), ws=rng.standard_normal((cm, cs), dtype=f) * f(np.sqrt(2.0 / cs)),
                 bs=rng.standard_normal(cm, dtype=f) * f(0.1))
    return t


def three_op_fp64(t, act):
    """the reference's form: upsample (+ skip link) -> decode 3x3 (-> GELU), fp64"""
    d = {k: (torch.from_numpy(v).double() if v is not None else None) for k, v in t.items()}
    u = F.conv_transpose2d(d["x"][None], d["wt"], d["bt"], stride=2)
    if d["skip"] is not None:
        u = u + F.conv2d(d["skip"][None], d["ws"][:, :, None, None], d["bs"])
    y = F.conv2d(u, d["w3"], d["b3"], padding=1)
    return (F.gelu(y) if act else y)[0].numpy()


def composed_fp64(t, wx, wsk, bias):
    """what the composed weights mean, evaluated by torch: per phase a 2x2 convolution of x = one 4x4 / s2 / p1 transposed convolution"""
    x = torch.from_numpy(t["x"]).double()[None]
    cout, cin = wx.shape[2:]
    h, w = x.shape[2:]
    out = torch.zeros(cout, 2 * h, 2 * w, dtype=torch.float64)
    xp = F.pad(x, (1, 1, 1, 1))
    for py in range(2):
        for px in range(2):
            k = torch.from_numpy(wx[py * 2 + px]).reshape(2, 2, cout, cin).permute(2, 3, 0, 1)       # [cout][cin][a][b]
            win = xp[:, :, py:py + h + 1, px:px + w + 1]                                            # rows y + py - 1 + a for a in {0, 1}
            out[:, py::2, px::2] = F.conv2d(win, k)[0]
    if wsk is not None:
        cs = wsk.shape[2]
        k = torch.from_numpy(wsk).reshape(3, 3, cout, cs).permute(2, 3, 0, 1)
        out += F.conv2d(torch.from_numpy(t["skip"]).double()[None], k, padding=1)[0]
    H2, W2 = 2 * h, 2 * w
    rc = np.ones(H2, dtype=int); rc[0] = 0; rc[-1] = 2
    cc = np.ones(W2, dtype=int); cc[0] = 0; cc[-1] = 2
    cls = rc[:, None] * 3 + cc[None, :]
    out += torch.from_numpy(bias[cls]).permute(2, 0, 1)
    return out.numpy()


@pytest.mark.parametrize("cin,cm,cout,cs,h,w", [(5, 7, 6, 3, 3, 5), (4, 4, 3, 0, 1, 1), (3, 5, 4, 2, 2, 1), (9, 8, 5, 4, 1, 4), (6, 6, 6, 0, 5, 3)])
def test_compose_upconv_equals_the_three_op_form_in_fp64(emu_lib, cin, cm, cout, cs, h, w):
    t = make_stage(np.random.default_rng(cin * 100 + h), cin, cm, cout, cs, h, w)
    wx, wsk, bias = emu_lib.compose_upconv(t["wt"], t["bt"], t["w3"], t["b3"], ws=t["ws"], bs=t["bs"])
    ref = three_op_fp64(t, act=0)
    got = composed_fp64(t, wx, wsk, bias)
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, float(np.abs(ref).max()))


def _op_case(lib, seed, cin, cm, cout, cs, h, w, act, cfgs, tol=2e-5, precision="fp16x3"):
    t = make_stage(np.random.default_rng(seed), cin, cm, cout, cs, h, w)
    ref = three_op_fp64(t, act)
    for shape, nsplit in cfgs:
        got = lib.op_upconv(t["x"], t["wt"], t["bt"], t["w3"], t["b3"], skip=t["skip"], ws=t["ws"], bs=t["bs"], act=act, shape=shape, nsplit=nsplit,
                            precision=precision)
        assert got.shape == ref.shape
        err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
        assert err <= tol, f"shape {shape} nsplit {nsplit}: err {err:.3e}"


def test_upconv_kernel_no_skip(emu_lib):
    """upsample_layer_4 + decode_layer_8's form: no skip tensor; one and several chunks; ragged low-resolution maps (rows and columns)"""
    _op_case(emu_lib, 1, 32, 32, 128, 0, 16, 16, 1, [(6, 1), (7, 1)])
    _op_case(emu_lib, 2, 96, 40, 72, 0, 9, 21, 0, [(6, 1), (7, 1), (7, 2)])


def test_upconv_kernel_with_skip(emu_lib):
    """every pixel class of the skip tensor with its tap list (4 / 2 / 2 / 1), one and two 32-channel skip chunks, borders of the map inside and at the
    edge of a patch, the last chunk of a slice being a 1-tap chunk, K slices cutting the x chunks and the skip chunks"""
    _op_case(emu_lib, 3, 32, 48, 128, 24, 8, 16, 1, [(6, 1), (7, 1)])
    _op_case(emu_lib, 4, 64, 64, 200, 40, 10, 20, 1, [(6, 1), (7, 1), (6, 3), (7, 5)])
    _op_case(emu_lib, 5, 40, 24, 128, 33, 17, 18, 0, [(6, 1), (7, 1), (7, 10)])


def test_upconv_kernel_fp16_form(emu_lib):
    """the VP_FP16 engines' form of the kernel (X1: 64-channel chunks, the chunk's halves in the two planes, two MFMAs per fragment pair, one output
    plane): no skip tensor; skip tensors of 32 channels (plane 1 of their chunks is dead), 64 and 128 channels (one and two full chunks per class), 96 channels (one full + one half-dead chunk);
    K slices; both shapes.  Tolerance: fp16 operands (2^-11 each) -- the regression class of the fp16 engines, not the parity bar."""
    _op_case(emu_lib, 11, 64, 32, 128, 0, 9, 17, 1, [(6, 1), (7, 1)], tol=6e-3, precision="fp16")
    _op_case(emu_lib, 12, 128, 48, 128, 24, 8, 16, 1, [(6, 1), (7, 1), (7, 2)], tol=6e-3, precision="fp16")
    _op_case(emu_lib, 13, 64, 64, 200, 40, 10, 20, 0, [(6, 1), (7, 3)], tol=6e-3, precision="fp16")
    _op_case(emu_lib, 14, 120, 40, 128, 112, 5, 9, 1, [(7, 1), (6, 4)], tol=6e-3, precision="fp16")
    _op_case(emu_lib, 16, 64, 40, 128, 80, 7, 11, 1, [(6, 1), (7, 3)], tol=6e-3, precision="fp16")   # 80 -> 96 channels: one full chunk + one half-dead chunk


def test_upconv_fp16_form_refuses_what_it_cannot_tile(emu_lib):
    t = make_stage(np.random.default_rng(15), 32, 32, 128, 0, 8, 16)   # 32 input channels pad to 32: not a multiple of the fp16 form's 64-channel chunk
    with pytest.raises(emu_lib.VpError, match="multiple of 64"):
        emu_lib.op_upconv(t["x"], t["wt"], t["bt"], t["w3"], t["b3"], precision="fp16")
