"""SURVEY.md N4, second half, on the CPU: oracle/autospeed.py's own properties, and the device kernels (kernels_detect.hip) + the C ABI
(vp_detect_*, csrc/vp_detect.cpp) executed through the HIP-on-CPU emulation against it, bit for bit."""
import ctypes as ct
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _autospeed_cases as cases  # noqa: E402
from oracle import autospeed, pre_post  # noqa: E402


def test_oracle_letterbox_geometry_and_canvas():
    """onnxruntime_engine.cpp:78-98: 1280x720 -> scale 0.5, 640x360 pasted at y = 140; the canvas is 114 / 255 outside; the pasted part is
    the integer bilinear of the plain preprocess; a frame already 640x640 is copied (scale 1); a portrait frame pads left / right."""
    f = cases.frame(720, 1280, 1)
    t, (scale, px, py) = autospeed.preprocess(f)
    assert (float(scale), px, py) == (0.5, 0, 140) and t.shape == (3, 640, 640) and t.dtype == np.float32
    g = np.float32(114) * np.float32(1.0 / 255.0)
    assert np.all(t[:, :140] == g) and np.all(t[:, 500:] == g)
    want = pre_post.resize_bilinear_u8(f, 360, 640).astype(np.float32) * np.float32(1.0 / 255.0)
    assert np.array_equal(t[:, 140:500], want[:, :, ::-1].transpose(2, 0, 1))
    f2 = cases.frame(640, 640, 2)
    t2, geom2 = autospeed.preprocess(f2)
    assert (float(geom2[0]), geom2[1], geom2[2]) == (1.0, 0, 0)
    assert np.array_equal(t2, (f2.astype(np.float32) * np.float32(1.0 / 255.0))[:, :, ::-1].transpose(2, 0, 1))
    t3, geom3 = autospeed.preprocess(cases.frame(487, 301, 3))
    s3, nw, nh, px3, py3 = autospeed.letterbox_geometry(487, 301)
    assert (nh, py3) == (640, 0) or nh == 639                      # 640 / 487 in fp32, truncated
    assert px3 == (640 - nw) // 2 and np.all(t3[:, :, :px3] == g)


def test_oracle_nms_rules():
    """postProcess / applyNMS by hand: same-class overlap suppressed, other class kept, lower confidence first in the output never, ties in
    box order, clamp to the frame, class -1 when no score is positive, the threshold is `<` (a score equal to it stays)."""
    raw = np.zeros((6, 6), np.float32)                               # 2 classes, 6 boxes, letterbox = image (scale 1, no pad)
    raw[:4, 0] = (100, 100, 50, 50); raw[4, 0] = 0.9                 # A  class 0
    raw[:4, 1] = (104, 100, 50, 50); raw[4, 1] = 0.8                 # overlaps A, class 0 -> suppressed
    raw[:4, 2] = (104, 100, 50, 50); raw[5, 2] = 0.8                 # overlaps A, class 1 -> kept
    raw[:4, 3] = (300, 300, 40, 40); raw[4, 3] = 0.5                 # B  ties with C in confidence
    raw[:4, 4] = (400, 300, 40, 40); raw[4, 4] = 0.5                 # C
    raw[:4, 5] = (630, 10, 60, 60); raw[5, 5] = 0.25                 # partly outside the frame; exactly the threshold
    det = autospeed.postprocess(raw, 0.25, 0.45, np.float32(1.0), 0, 0, 640, 640)
    assert [tuple(d[4:]) for d in det] == [(np.float32(0.9), 0.0), (np.float32(0.8), 1.0), (np.float32(0.5), 0.0), (np.float32(0.5), 0.0), (np.float32(0.25), 1.0)]
    assert det[2][0] == 280 and det[3][0] == 380                     # the tie keeps the box order
    assert tuple(det[4][:4]) == (600.0, 0.0, 640.0, 40.0)            # clamped
    det0 = autospeed.postprocess(raw[:, :1] * 0, 0.0, 0.45, np.float32(1.0), 0, 0, 640, 640)
    assert det0.shape == (1, 6) and det0[0, 5] == -1 and det0[0, 4] == 0   # threshold 0: a box with no positive score comes out as class -1


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


@pytest.mark.parametrize("shape", [(720, 1280), (487, 301), (640, 640), (33, 900)])
def test_letterbox_kernel_bit_exact(emu_lib, shape):
    det = emu_lib.Detector()
    try:
        f = cases.frame(shape[0], shape[1], shape[0])
        got, geom = det.preprocess(f)
        want, wgeom = autospeed.preprocess(f)
        assert (geom[0], geom[1], geom[2]) == (wgeom[0], wgeom[1], wgeom[2])
        assert np.array_equal(got, want)
        view = np.ascontiguousarray(np.pad(f, ((0, 0), (0, 7), (0, 0))))[:, :shape[1]]   # a strided view: rows longer than 3 w
        got2, _ = det.preprocess(view)
        assert np.array_equal(got2, want)
        flat = np.zeros(shape[0] * (shape[1] + 7) * 3 - 21, np.uint8)                  # ... that ENDS with the last pixel of the last row
        tight = np.lib.stride_tricks.as_strided(flat, shape=(shape[0], shape[1], 3), strides=((shape[1] + 7) * 3, 3, 1))
        tight[:] = f
        got3, _ = det.preprocess(tight)
        assert np.array_equal(got3, want)
    finally:
        det.close()


def test_decode_nms_kernel_bit_exact(emu_lib):
    det = emu_lib.Detector(max_boxes=8400, max_attrs=12)
    try:
        f = cases.frame(720, 1280, 5)
        _, geom = det.preprocess(f)
        for seed, (nb, nc, conf, iou) in enumerate([(8400, 4, 0.25, 0.45), (2100, 8, 0.5, 0.3), (700, 1, 0.0, 0.45), (8400, 4, 1.5, 0.45), (37, 3, 0.1, 0.0)]):
            raw = cases.raw_tensor(nb, nc, 100 + seed)
            got, n = det.postprocess(raw, conf, iou)
            kept = cases.check(got, n, raw, conf, iou, geom, 1280, 720)
            assert (kept == 0) == (conf > 1.0)
            if kept > 3:                                               # a caller's buffer smaller than the result: count says so
                got3, n3 = det.postprocess(raw, conf, iou, cap=3)
                cases.check(got3, n3, raw, conf, iou, geom, 1280, 720, cap=3)
        with pytest.raises(emu_lib.VpError):
            det.postprocess(np.zeros((4, 10), np.float32), 0.25, 0.45)   # no class rows
        with pytest.raises(emu_lib.VpError):
            det.postprocess(np.zeros((6, 9000), np.float32), 0.25, 0.45)   # more boxes than the handle was made for
    finally:
        det.close()


def test_postprocess_needs_geometry(emu_lib):
    det = emu_lib.Detector(max_boxes=64, max_attrs=6)
    try:
        with pytest.raises(emu_lib.VpError):
            det.postprocess(np.zeros((6, 8), np.float32), 0.25, 0.45)
        det.set_letterbox(np.float32(0.5), 0, 140, 1280, 720)          # the geometry of a frame preprocessed elsewhere
        raw = cases.raw_tensor(64, 2, 9, clusters=6, p_obj=0.5)
        got, n = det.postprocess(raw, 0.25, 0.45)
        cases.check(got, n, raw, 0.25, 0.45, (np.float32(0.5), 0, 140), 1280, 720)
    finally:
        det.close()


def test_decode_nms_tie_free_cases_emulated(emu_lib):
    """As tests/test_gpu_autospeed.py::test_decode_nms_tie_free_cases, through the CPU emulation."""
    det = emu_lib.Detector(max_boxes=2100, max_attrs=12)
    try:
        for (nb, nc, seed, useed), (conf, iou, scale, px, py, ow, oh) in cases.TIE_FREE_CASES:
            raw = cases.untie(cases.raw_tensor(nb, nc, seed), useed)
            det.set_letterbox(np.float32(scale), int(px), int(py), int(ow), int(oh))
            got, n = det.postprocess(raw, conf, iou)
            cases.check(got, n, raw, np.float32(conf), np.float32(iou), (np.float32(scale), int(px), int(py)), int(ow), int(oh))
    finally:
        det.close()
