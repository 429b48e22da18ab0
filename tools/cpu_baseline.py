#!/usr/bin/env python3
"""The cpu_baseline leg of bench.py on its own (no GPU needed): the CPU oracle timed on this host.
    python tools/cpu_baseline.py --threads 1        # per-core figure
    python tools/cpu_baseline.py                    # all host cores"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from autoware_vision_pilot_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="sceneseg")
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--seconds", type=float, default=20.0)
a = ap.parse_args()
sd = synthetic.make_state_dict(a.kind, 1)
frame = synthetic.synthetic_frame(720, 1280, 3)
print(json.dumps(bench.cpu_baseline(a.kind, sd, frame, a.seconds, a.threads)))
