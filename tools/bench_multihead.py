#!/usr/bin/env python3
"""BASELINE configs[2]: SceneSeg + Scene3D + EgoLanes on ONE 1280x720 camera frame, shared encoder, 1 MI355X.

python tools/bench_multihead.py [--precision fp16] [--steps 300] [--warmup 30] [--streams 3] [--no-share]
One step = preprocess + SceneSeg (base engine) + Scene3D + EgoLanes heads on the same frame; with sharing the two
extra heads start from the base engine's backbone taps (vp_create_shared).  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from autoware_vision_pilot_amd import lib, weights as vw
lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)
from autoware_vision_pilot_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--warmup", type=int, default=30)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--no-share", action="store_true")
a = ap.parse_args()

sd_seg = synthetic.make_state_dict("sceneseg", 0)
sd_3d = synthetic.share_backbone(synthetic.make_state_dict("scene3d", 1), "scene3d", sd_seg, "sceneseg")
sd_ego = synthetic.share_backbone(synthetic.make_state_dict("egolanes", 2), "egolanes", sd_seg, "sceneseg")
blobs = {"sceneseg": vw.pack_state_dict(sd_seg), "scene3d": vw.pack_state_dict(sd_3d), "egolanes": vw.pack_state_dict(sd_ego)}
frame = synthetic.synthetic_frame(720, 1280, 1)
groups = []
for _ in range(a.streams):
    base = lib.Engine("sceneseg", blobs["sceneseg"], precision=a.precision)
    if a.no_share:
        heads = [lib.Engine(k, blobs[k], precision=a.precision) for k in ("scene3d", "egolanes")]
        for h in heads:
            h.upload_frame(frame)
    else:
        heads = [lib.Engine(k, blobs[k], precision=a.precision, base=base) for k in ("scene3d", "egolanes")]
    base.upload_frame(frame)
    groups.append((base, heads))

def step(i):
    base, heads = groups[i % len(groups)]
    base.enqueue()
    for h in heads:
        h.enqueue()

def sync_all():
    for base, heads in groups:
        base.sync()
        for h in heads:
            h.sync()

for i in range(a.warmup):
    step(i)
sync_all()
t0 = time.perf_counter()
for i in range(a.steps):
    step(i)
sync_all()
dt = time.perf_counter() - t0
fps = a.steps / dt
gflop = 367.0 + 397.0 + 196.7 - (0.0 if a.no_share else 2 * 3.136)  # BASELINE.md section 2; shared backbones run once
print(json.dumps({"metric": "frames/sec, SceneSeg+Scene3D+EgoLanes on one 1280x720 camera (BASELINE configs[2])", "value": round(fps, 2),
                  "unit": "frames/s", "ms_per_frame": round(1e3 / fps, 4), "precision": a.precision, "frames_in_flight": a.streams,
                  "shared_encoder": not a.no_share, "gflop_per_frame": round(gflop, 1),
                  "achieved_tflops": round(gflop * fps / 1e3, 1), "frac_of_2.5PF": round(gflop * fps / 2.5e6, 4)}))
