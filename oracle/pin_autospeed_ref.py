#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: pin oracle/autospeed.py against the REFERENCE'S OWN code and write tests/golden/autospeed_ref.npz.

    python -m oracle.pin_autospeed_ref        (build container only: needs /root/reference and g++)

`make -C oracle` compiles VisionPilot/middleware_recipes/common/backends/autospeed/onnxruntime_engine.cpp where it lies (never copied) on the
stand-in headers of oracle/ref_stubs into oracle/_ref/autospeed_ref; this script feeds it
  * detector tensors -> its postProcess + computeIoU + applyNMS (:170-290): the restatement must return the same detections bit for bit --
    kept set, order, coordinates, confidence, class -- on tensors WITHOUT equal confidences (with ties the reference's unstable std::sort
    decides; one such case is run and reported, not pinned);
  * frames -> its preprocessAutoSpeed (:71-113): scale_, pad_x_, pad_y_ and the [3][640][640] tensor must match the restatement given
    the same cv::resize placeholder (nearest neighbour, oracle/ref_stubs/opencv2/opencv.hpp) -- geometry, canvas of 114s, /255, plane order.
The fixtures (inputs + the reference binary's outputs) are what tests/test_oracle_golden.py checks the oracle against on any machine."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _autospeed_cases as cases  # noqa: E402
from oracle import autospeed  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "autospeed_ref")


def nearest(img, new_h, new_w):
    """oracle/ref_stubs/opencv2/opencv.hpp cv::resize placeholder."""
    h, w = img.shape[:2]
    ys = np.minimum(h - 1, np.arange(new_h, dtype=np.int64) * h // new_h)
    xs = np.minimum(w - 1, np.arange(new_w, dtype=np.int64) * w // new_w)
    return img[ys][:, xs]


def ref_post(raw, conf, iou, scale, pad_x, pad_y, ow, oh, tmp):
    rawf, outf = os.path.join(tmp, "raw.bin"), os.path.join(tmp, "det.bin")
    np.ascontiguousarray(raw, np.float32).tofile(rawf)
    subprocess.run([BIN, "post", rawf, str(raw.shape[0]), str(raw.shape[1]), repr(float(conf)), repr(float(iou)), repr(float(scale)), str(pad_x), str(pad_y), str(ow), str(oh), outf],
                   check=True, capture_output=True)
    blob = np.fromfile(outf, dtype=np.uint8)
    n = int(blob[:4].view(np.int32)[0])
    rec = blob[4:4 + 24 * n]
    det = rec.view(np.float32).reshape(n, 6).copy()
    det[:, 5] = rec.view(np.int32).reshape(n, 6)[:, 5].astype(np.float32)
    return det


def ref_pre(frame, tmp):
    ff, outf = os.path.join(tmp, "frame.bin"), os.path.join(tmp, "pre.bin")
    np.ascontiguousarray(frame).tofile(ff)
    subprocess.run([BIN, "pre", ff, str(frame.shape[0]), str(frame.shape[1]), outf], check=True, capture_output=True)
    blob = np.fromfile(outf, dtype=np.uint8)
    scale = blob[:4].view(np.float32)[0]
    px, py = (int(v) for v in blob[4:12].view(np.int32))
    return blob[12:].view(np.float32).reshape(3, 640, 640).copy(), (scale, px, py)


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- postProcess + NMS
        geom = [(np.float32(0.5), 0, 140, 1280, 720), (np.float32(640.0 / 1920.0), 0, 140, 1920, 1080), (np.float32(1.0), 0, 0, 640, 640), (autospeed.letterbox_geometry(487, 301)[0], 122, 0, 301, 487)]
        specs = [(2100, 4, 0.25, 0.45, 0), (2100, 8, 0.5, 0.3, 1), (700, 1, 0.05, 0.45, 2), (2100, 4, 1.5, 0.45, 0), (37, 3, 0.1, 0.0, 3), (2100, 4, 0.2, 0.6, 1)]
        for i, (nb, nc, conf, iou, g) in enumerate(specs):
            raw = cases.untie(cases.raw_tensor(nb, nc, 300 + i), i)
            scale, px, py, ow, oh = geom[g]
            det_ref = ref_post(raw, conf, iou, scale, px, py, ow, oh, tmp)
            det_orc = autospeed.postprocess(raw, conf, iou, scale, px, py, ow, oh)
            same = det_ref.shape == det_orc.shape and np.array_equal(det_ref.view(np.uint32), det_orc.view(np.uint32))
            print(f"postProcess case {i}: {nb} boxes x {nc} classes, conf {conf} iou {iou}: reference keeps {len(det_ref)}, restatement {'IDENTICAL (bit for bit, same order)' if same else 'DIFFERS'}")
            assert same
            out[f"post{i}_spec"] = np.array([nb, nc, 300 + i, i], np.int64)           # the tensor is cases.untie(cases.raw_tensor(nb, nc, seed), i)
            out[f"post{i}_sum"] = np.array([float(raw.astype(np.float64).sum())])       # ... and this is its checksum
            out[f"post{i}_args"] = np.array([conf, iou, scale, px, py, ow, oh], np.float64)
            out[f"post{i}_det"] = det_ref
        # with ties: reported, not pinned
        raw = np.nan_to_num(cases.raw_tensor(2100, 4, 400))
        det_ref = ref_post(raw, 0.25, 0.45, np.float32(0.5), 0, 140, 1280, 720, tmp)
        det_orc = autospeed.postprocess(raw, 0.25, 0.45, np.float32(0.5), 0, 140, 1280, 720)
        key = lambda d: sorted(map(tuple, d.view(np.uint32).tolist()))
        print(f"postProcess with EQUAL confidences (two decimals): reference keeps {len(det_ref)}, restatement {len(det_orc)}; same order: "
              f"{det_ref.shape == det_orc.shape and np.array_equal(det_ref.view(np.uint32), det_orc.view(np.uint32))}; same set: {key(det_ref) == key(det_orc)} "
              f"(the reference's std::sort is not stable: ties are its choice)")
        # ---- preprocessAutoSpeed: geometry, canvas, /255, plane order (resize = the placeholder on both sides)
        for i, (h, w) in enumerate([(720, 1280), (1080, 1920), (487, 301), (640, 640), (33, 900), (641, 1283)]):
            f = cases.frame(h, w, 50 + i)
            t_ref, g_ref = ref_pre(f, tmp)
            t_orc, g_orc = autospeed.preprocess(f, resize_fn=nearest)
            same = g_ref[0] == g_orc[0] and g_ref[1:] == tuple(g_orc[1:]) and np.array_equal(t_ref, t_orc)
            print(f"preprocessAutoSpeed {h}x{w}: scale {g_ref[0]:.6f} pad ({g_ref[1]}, {g_ref[2]}): restatement {'IDENTICAL' if same else 'DIFFERS'}")
            assert same
            out[f"pre{i}_hw"] = np.array([h, w], np.int32)
            out[f"pre{i}_geom"] = np.array([g_ref[0], g_ref[1], g_ref[2]], np.float64)
            out[f"pre{i}_checksum"] = np.array([float(t_ref.astype(np.float64).sum()), float(t_ref[0, 320, 320]), float(t_ref[2, 0, 0])], np.float64)
    path = os.path.join(ROOT, "tests", "golden", "autospeed_ref.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
