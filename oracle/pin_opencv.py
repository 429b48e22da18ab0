"""Pin the OpenCV stages of the hot path against OpenCV ITSELF, wherever ``cv2`` imports (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The C++ front-ends resize the camera frame with ``cv::resize(..., INTER_LINEAR)`` (common/backends/onnx_runtime_backend.cpp:44,
production_release/src/inference/onnxruntime_engine.cpp:78), the ROS2 node up-sizes the depth map with ``cv::resize INTER_LINEAR`` on
CV_32F (run_model_node.cpp:104) and the class map with ``INTER_NEAREST`` (run_model_node.cpp:177).  OpenCV is neither in the reference
tree nor in this image (``import cv2`` fails; no network), so oracle/pre_post.py RESTATES OpenCV's published schemes -- 11-bit fixed-point
taps for the u8 bilinear resize, fp32 taps for the float one, ``floor(dst * scale)`` for nearest -- and rows a1 / a2 / a14 / a16 of
SURVEY.md section 8 are bit-exact against that restatement: "parity unpinned" at the OpenCV boundary.

This script turns "unpinned" into "pinned on first contact": on any machine where ``cv2`` imports it
  1. runs cv2.resize on seeded frames at the sizes the tests use (1280x720, 1920x1080, 640x360, odd 651x487, and a 2x3 corner case),
  2. compares bit for bit with the restatement (u8 bilinear, nearest) and to <= 1 ulp (fp32 bilinear),
  3. with --write stores sampled fixtures in tests/golden/opencv_pin.npz, which tests/test_oracle_golden.py::test_opencv_pin_fixture
     re-checks on every later run WITHOUT cv2.
Here (no cv2) it prints why it cannot run and exits 0 -- nothing is claimed.

usage: python -m oracle.pin_opencv [--write]
"""
import os
import sys

import numpy as np

from . import pre_post

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "opencv_pin.npz")
CASES = [(720, 1280, 11), (1080, 1920, 12), (360, 640, 13), (487, 651, 14), (2, 3, 15), (320, 640, 16)]


def main(argv):
    try:
        import cv2
    except Exception as ex:  # noqa: BLE001
        print(f"pin_opencv: cv2 does not import here ({ex!r}); the OpenCV stages stay pinned to the restatement only (SURVEY.md 8c: parity unpinned)")
        return 0
    worst_f32 = 0.0
    fixtures = {"cv2_version": np.array(cv2.__version__)}
    for h, w, seed in CASES:
        frame = pre_post.synthetic_frame(h, w, seed, smooth=(seed % 2 == 1))
        # a1 / a2: u8 frame -> 640x320, INTER_LINEAR
        want = cv2.resize(frame, (pre_post.NET_W, pre_post.NET_H), interpolation=cv2.INTER_LINEAR)
        got = pre_post.resize_bilinear_u8(frame)
        if not np.array_equal(got, want):
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            print(f"pin_opencv: u8 INTER_LINEAR {w}x{h} -> 640x320 differs from cv2 {cv2.__version__}: {int((d > 0).sum())} bytes, max |diff| {int(d.max())}")
            return 1
        # a16: class map 320x640 -> frame size, INTER_NEAREST
        rng = np.random.default_rng(seed)
        mask = (rng.integers(0, 2, size=(pre_post.NET_H, pre_post.NET_W), dtype=np.uint8) * 255).astype(np.uint8)
        want_m = cv2.resize(mask, (w, h), interpolation=cv2.INTER_NEAREST)
        if not np.array_equal(pre_post.resize_nearest_u8(mask, h, w), want_m):
            print(f"pin_opencv: INTER_NEAREST 640x320 -> {w}x{h} differs from cv2 {cv2.__version__}")
            return 1
        # a14: depth map fp32 320x640 -> frame size, INTER_LINEAR on CV_32F
        depth = rng.standard_normal((pre_post.NET_H, pre_post.NET_W)).astype(np.float32) * np.float32(7.0)
        want_d = cv2.resize(depth, (w, h), interpolation=cv2.INTER_LINEAR)
        got_d = pre_post.resize_bilinear_f32(depth, h, w)
        ulp = np.abs(got_d - want_d) / np.maximum(np.spacing(np.abs(want_d)), np.float32(1e-30))
        worst_f32 = max(worst_f32, float(ulp.max()))
        if h * w <= 651 * 487:   # small cases go into the fixture whole; the large ones as strided samples
            fixtures[f"u8_{h}x{w}"] = want
            fixtures[f"nearest_{h}x{w}"] = want_m
            fixtures[f"f32_{h}x{w}"] = want_d
        else:
            fixtures[f"u8_{h}x{w}"] = want[::7, ::11]
            fixtures[f"nearest_{h}x{w}"] = want_m[::13, ::17]
            fixtures[f"f32_{h}x{w}"] = want_d[::13, ::17]
    print(f"pin_opencv: cv2 {cv2.__version__}: u8 INTER_LINEAR and INTER_NEAREST bit-exact on {len(CASES)} sizes; fp32 INTER_LINEAR within {worst_f32:.2f} ulp")
    if worst_f32 > 1.0:
        return 1
    if "--write" in argv:
        np.savez_compressed(GOLDEN, **fixtures)
        print(f"pin_opencv: wrote {GOLDEN}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
