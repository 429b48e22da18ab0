#!/usr/bin/env python3
"""BASELINE configs[4]: AutoDrive on 1920x1080 frames, fp8 (e4m3) weights, streaming (one backbone pass + head per frame).

python tools/bench_autodrive.py [--precision fp16] [--steps 500] [--warmup 50] [--streams 3] [--no-fp8]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from autoware_vision_pilot_amd import lib, synthetic, weights as vw
lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16")
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--warmup", type=int, default=50)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--no-fp8", action="store_true")
a = ap.parse_args()
blob = vw.pack_state_dict(synthetic.make_autodrive_state_dict(5))
frame = synthetic.synthetic_frame(1080, 1920, 21)
engs = [lib.Engine("autodrive", blob, precision=a.precision, weights_fp8=not a.no_fp8) for _ in range(a.streams)]
for e in engs:
    e.upload_frame(frame)
for i in range(a.warmup):
    engs[i % len(engs)].enqueue()
for e in engs:
    e.sync()
t0 = time.perf_counter()
for i in range(a.steps):
    engs[i % len(engs)].enqueue()
for e in engs:
    e.sync()
dt = time.perf_counter() - t0
lat = []
for _ in range(50):
    t1 = time.perf_counter(); engs[0].enqueue(); engs[0].sync(); lat.append((time.perf_counter() - t1) * 1e3)
fps = a.steps / dt
print(json.dumps({"metric": "frames/sec, AutoDrive 1920x1080 streaming (BASELINE configs[4])", "value": round(fps, 1), "unit": "frames/s",
                  "precision": a.precision,
                  "weights": "fp32 checkpoint -> fp16 planes" if a.no_fp8 else "fp8: one OCP e4m3 byte per weight + per-row fp32 scale in HBM, converted in the weight-staging path (round 4)",
                  "weight_bytes": engs[0].weight_bytes(), "library": lib.version(), "plan_hash": f"{engs[0].plan_hash():016x}",
                  "frames_in_flight": a.streams, "p50_ms": round(float(np.percentile(lat, 50)), 3),
                  "gflop_per_frame": 8.1, "launches_per_frame": len(engs[0].layers())}))
