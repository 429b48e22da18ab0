#!/usr/bin/env python3
"""Developer tool: host-to-host time of the AutoSpeed detector's device stages (vp_detect_preprocess: letterbox of a 1920x1080 BGR frame to
640x640 planes; vp_detect_postprocess: decode + per-class NMS of a [4 + classes][8400] tensor), p50 over 200 calls each.

python tools/bench_detect.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401
from autoware_vision_pilot_amd import lib
import _autospeed_cases as cases

det = lib.Detector(640, 640, 8400, 84)
frame = cases.frame(1080, 1920, 3)
out = {}
for _ in range(20):
    det.preprocess(frame)
t = []
for _ in range(200):
    t0 = time.perf_counter(); det.preprocess(frame); t.append((time.perf_counter() - t0) * 1e6)
out["preprocess_1920x1080_to_640x640_p50_us"] = round(float(np.percentile(t, 50)), 1)
for name, nc, p_obj in (("4_classes_8400_boxes", 4, 0.12), ("80_classes_8400_boxes", 80, 0.12), ("4_classes_dense", 4, 0.6)):
    raw = cases.raw_tensor(8400, nc, 11, p_obj=p_obj)
    for _ in range(20):
        d, n = det.postprocess(raw, 0.25, 0.45)
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); d, n = det.postprocess(raw, 0.25, 0.45); t.append((time.perf_counter() - t0) * 1e6)
    out[f"postprocess_{name}_p50_us"] = round(float(np.percentile(t, 50)), 1)
    out[f"postprocess_{name}_candidates_kept"] = [int((np.nan_to_num(raw[4:]).max(axis=0) > 0.25).sum()), int(n)]
out["library"] = lib.version()
print(json.dumps(out))
