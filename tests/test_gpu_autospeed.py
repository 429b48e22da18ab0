"""SURVEY.md N4, second half, on the MI355X: the AutoSpeed detector's letterbox preprocess and decode + NMS (kernels_detect.hip) through the
C ABI (vp_detect_*) against oracle/autospeed.py -- bit for bit: the tensors, the kept set, its order and every coordinate."""
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _autospeed_cases as cases  # noqa: E402
from oracle import autospeed  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    from autoware_vision_pilot_amd import lib

    d = lib.Detector(max_boxes=16384, max_attrs=84)
    yield d
    d.close()


@pytest.mark.parametrize("shape", [(720, 1280), (1080, 1920), (487, 301), (640, 640), (33, 900), (2160, 3840)])
def test_letterbox_bit_exact(det, shape):
    f = cases.frame(shape[0], shape[1], shape[1])
    got, geom = det.preprocess(f)
    want, wgeom = autospeed.preprocess(f)
    assert tuple(geom) == tuple(wgeom)
    assert np.array_equal(got, want)


def test_decode_nms_bit_exact(det):
    f = cases.frame(720, 1280, 5)
    _, geom = det.preprocess(f)
    for seed, (nb, nc, conf, iou) in enumerate([(8400, 4, 0.25, 0.45), (8400, 80, 0.25, 0.45), (2100, 8, 0.5, 0.3), (700, 1, 0.0, 0.45), (8400, 4, 1.5, 0.45),
                                                (37, 3, 0.1, 0.0), (16384, 4, 0.25, 0.6), (8400, 4, 0.0, 0.45)]):
        raw = cases.raw_tensor(nb, nc, 100 + seed)
        got, n = det.postprocess(raw, conf, iou)
        kept = cases.check(got, n, raw, conf, iou, geom, 1280, 720)
        if kept > 3:
            got3, n3 = det.postprocess(raw, conf, iou, cap=3)
            cases.check(got3, n3, raw, conf, iou, geom, 1280, 720, cap=3)


def test_dense_scene_every_box_a_candidate(det):
    """Worst case of the stage: every one of 8400 boxes above the threshold (the whole sort, thousands of kept boxes)."""
    f = cases.frame(1080, 1920, 6)
    _, geom = det.preprocess(f)
    raw = cases.raw_tensor(8400, 4, 77, clusters=3000, p_obj=1.0)
    got, n = det.postprocess(raw, 0.1, 0.45)
    kept = cases.check(got, n, raw, 0.1, 0.45, geom, 1920, 1080)
    assert kept > 1000


def test_stage_times(det, capsys):
    f = cases.frame(720, 1280, 8)
    raw = cases.raw_tensor(8400, 4, 1)
    det.preprocess(f)
    det.postprocess(raw, 0.25, 0.45)
    t0 = time.perf_counter()
    for _ in range(20):
        det.preprocess(f)
    t1 = time.perf_counter()
    for _ in range(20):
        det.postprocess(raw, 0.25, 0.45)
    t2 = time.perf_counter()
    with capsys.disabled():
        print(f"\n[autospeed stages, host-to-host incl. copies] letterbox 1280x720 -> 640x640: {(t1 - t0) / 20 * 1e3:.3f} ms; decode + NMS of 8400 boxes: {(t2 - t1) / 20 * 1e3:.3f} ms")


def test_decode_nms_tie_free_cases(det):
    """Tensors without equal confidences (the reference's std::sort is unstable: only there is the order of the detections fully specified by its
    code): the device stage returns the restatement's detections bit for bit -- kept set, order, coordinates, confidence, class."""
    for (nb, nc, seed, useed), (conf, iou, scale, px, py, ow, oh) in cases.TIE_FREE_CASES:
        raw = cases.untie(cases.raw_tensor(nb, nc, seed), useed)
        det.set_letterbox(np.float32(scale), int(px), int(py), int(ow), int(oh))
        got, n = det.postprocess(raw, conf, iou)
        cases.check(got, n, raw, np.float32(conf), np.float32(iou), (np.float32(scale), int(px), int(py)), int(ow), int(oh))
