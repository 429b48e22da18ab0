"""The REAL source of the non-MFMA kernels (csrc/kernels_misc.hip, kernels_backbone.hip, kernels_autodrive.hip) executed on
the CPU through tests/emul (a HIP-on-CPU shim: one host thread per work-item, real barriers / shuffles / atomics) and
checked against the oracle.  Runs in the CPU suite, so an indexing or arithmetic regression in these kernels shows up
without a GPU; the MFMA convolution kernels stay GPU-only (tests/test_gpu_*.py).  Integer / byte work: bit-exact; float
work: the tolerance the GPU parity tests use (the host compiler does not contract a*b+c, the device does)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pre_post

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))


@pytest.fixture(scope="module")
def emu():
    import build as emul_build

    if not os.path.exists(emul_build.CLANG):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    return C.CDLL(emul_build.build())


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def taps_u8(src, dst):
    return np.ascontiguousarray(np.stack(pre_post.linear_taps_u8(src, dst), axis=1).astype(np.int32))


def split16(x):
    """fp32 array -> (hi, lo) fp16 planes of the engine's activation format."""
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return np.ascontiguousarray(hi), np.ascontiguousarray(lo)


def nhwc(x_chw, cpad):
    """CxHxW fp32 -> HxWxCpad fp32 (zero pad channels)."""
    c, h, w = x_chw.shape
    out = np.zeros((h, w, cpad), dtype=np.float32)
    out[..., :c] = x_chw.transpose(1, 2, 0)
    return out


# ---------------------------------------------------------------------------------------------- pre / post-processing
@pytest.mark.parametrize("shape,bgr,planes_rgb", [((45, 80), True, False), ((32, 64), False, True), ((50, 37), True, True)])
def test_preprocess_kernel_bit_exact(emu, shape, bgr, planes_rgb):
    oh, ow = 32, 64
    frame = pre_post.synthetic_frame(shape[0], shape[1], 3, smooth=False)
    want = pre_post.preprocess(frame, input_is_bgr=bgr, planes_rgb=planes_rgb, out_h=oh, out_w=ow)[0]
    xt, yt = taps_u8(shape[1], ow), taps_u8(shape[0], oh)
    src_c, mean, std = np.zeros(3, np.int32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    for c in range(3):                                   # engine.cpp preprocess op: plane c -> colour -> source byte
        colour = c if planes_rgb else 2 - c
        src_c[c] = (2 - colour) if bgr else colour
        mean[c], std[c] = pre_post.MEAN_RGB[colour], pre_post.STD_RGB[colour]
    out = np.empty((3, oh, ow), dtype=np.float32)
    assert emu.emu_preprocess(ptr(frame), frame.strides[0], ptr(xt), ptr(yt), oh, ow, ptr(src_c), ptr(mean), ptr(std), ptr(out)) == 0
    assert np.array_equal(out, want)


def test_decode_kernels_bit_exact(emu):
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((3, 20, 30)).astype(np.float32)
    logits[:, 0, :5] = 0.25                                # ties: first maximum wins
    logits[:, 1, :5] = 0.0                                 # exactly zero is not "> 0"
    out = np.empty((20, 30), dtype=np.uint8)
    for mode, want in ((0, pre_post.seg_mask_u8(logits)), (1, pre_post.egolanes_priority_mask(logits)),
                       (2, pre_post.argmax_classes(logits).astype(np.uint8))):
        assert emu.emu_decode_mask(ptr(logits), 3, 600, mode, ptr(out)) == 0
        assert np.array_equal(out, want), mode
    one = np.ascontiguousarray(logits[:1])
    assert emu.emu_decode_mask(ptr(one), 1, 600, 0, ptr(out)) == 0
    assert np.array_equal(out, pre_post.seg_mask_u8(one))


def test_resize_kernels_bit_exact(emu):
    rng = np.random.default_rng(1)
    mask = rng.integers(0, 256, size=(20, 30), dtype=np.uint8)
    oh, ow = 47, 101
    yt, xt = pre_post.nearest_index(20, oh).astype(np.int32), pre_post.nearest_index(30, ow).astype(np.int32)
    out = np.empty((oh, ow), dtype=np.uint8)
    assert emu.emu_resize_nearest(ptr(mask), 30, ptr(yt), ptr(xt), oh, ow, ptr(out)) == 0
    assert np.array_equal(out, pre_post.resize_nearest_u8(mask, oh, ow))

    plane = rng.standard_normal((20, 30)).astype(np.float32)

    def taps(src, dst):
        s0, s1, a0, a1 = pre_post.linear_taps_f32(src, dst)
        return (np.ascontiguousarray(np.stack([s0, s1], axis=1).astype(np.int32)), np.ascontiguousarray(np.stack([a0, a1], axis=1).astype(np.float32)))

    (yi, yf), (xi, xf) = taps(20, oh), taps(30, ow)
    outf = np.empty((oh, ow), dtype=np.float32)
    assert emu.emu_resize_bilinear_f32(ptr(plane), 30, ptr(yi), ptr(yf), ptr(xi), ptr(xf), oh, ow, ptr(outf)) == 0
    assert np.array_equal(outf, pre_post.resize_bilinear_f32(plane, oh, ow))


def test_visualisation_kernels_bit_exact(emu):
    rng = np.random.default_rng(2)
    frame = pre_post.synthetic_frame(45, 80, 5)
    oh, ow = frame.shape[:2]
    out = np.empty((oh, ow, 3), dtype=np.uint8)
    for viz, labels in ((0, (0, 255)), (1, (0, 255)), (2, (0, 1, 2, 255))):
        mask = rng.choice(np.array(labels, dtype=np.uint8), size=(20, 30))
        yt, xt = pre_post.nearest_index(20, oh).astype(np.int32), pre_post.nearest_index(30, ow).astype(np.int32)
        lut = np.ascontiguousarray(pre_post.viz_lut()[viz])
        assert emu.emu_viz_blend(ptr(mask), 30, ptr(yt), ptr(xt), ptr(frame), frame.strides[0], oh, ow, ptr(lut), 0, ptr(out)) == 0
        assert np.array_equal(out, pre_post.visualize_mask(mask, frame, viz)), viz
    pytest.importorskip("matplotlib")
    depth = (rng.standard_normal((37, 53)) * 3 - 1).astype(np.float32)
    lut = pre_post.viridis_lut_bgr()
    outd = np.empty((37, 53, 3), dtype=np.uint8)
    emu.emu_depth_viz.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    assert emu.emu_depth_viz(ptr(depth), depth.size, ptr(lut), ptr(outd)) == 0
    assert np.array_equal(outd, pre_post.visualize_depth(depth))
    flat = np.full((8, 8), -2.5, dtype=np.float32)           # max == min -> all zeros -> LUT[0]
    outf = np.empty((8, 8, 3), dtype=np.uint8)
    assert emu.emu_depth_viz(ptr(flat), flat.size, ptr(lut), ptr(outf)) == 0
    assert np.array_equal(outf, np.broadcast_to(lut[0], (8, 8, 3)))
