#!/usr/bin/env python3
"""Developer tool: does the batched encoder (vp_create_batched + one shared-prefix head per camera) beat independent
engines in flight?  Times frames/s of SceneSeg for
    A) `streams` independent single-frame engines, round-robin                       (bench.py's configuration)
    B) `groups` x [one batched encoder over `batch` cameras + `batch` heads], round-robin
on the same GPU, same precision.  One JSON line per configuration.
usage: python tools/bench_batched.py [precision] [seconds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from autoware_vision_pilot_amd import lib, synthetic, weights as vw  # noqa: E402
lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
kind = "sceneseg"
blob = vw.pack_state_dict(synthetic.make_state_dict(kind, 0))
frame = synthetic.synthetic_frame(720, 1280, 10)


class Single:
    frames = 1

    def __init__(self):
        self.e = lib.Engine(kind, blob, precision=prec)
        self.e.upload_frame(frame)

    def enqueue(self):
        self.e.enqueue()

    def sync(self):
        self.e.sync()

    def close(self):
        self.e.close()


class Group:
    def __init__(self, batch):
        self.frames = batch
        self.enc = lib.Engine(kind, blob, precision=prec, frames=batch)
        self.heads = [lib.Engine(kind, blob, precision=prec, base=self.enc, frame_index=f) for f in range(batch)]
        for f in range(batch):
            self.enc.upload_frame(synthetic.synthetic_frame(720, 1280, 10 + 100 * f), index=f)

    def enqueue(self):
        self.enc.enqueue()
        for h in self.heads:
            h.enqueue()

    def sync(self):
        self.enc.sync()

    def close(self):
        for h in self.heads:
            h.close()
        self.enc.close()


def run(make, n, label):
    slots = [make() for _ in range(n)]
    for s in slots:
        s.enqueue()
        s.enqueue()
        s.sync()
    per = slots[0].frames

    def timed(steps):
        for s in slots:
            s.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            slots[i % n].enqueue()
        for s in slots:
            s.sync()
        return time.perf_counter() - t0

    timed(10)
    probe = timed(20) / 20
    k = max(30, int(1.1 * seconds / probe))
    el = timed(k)
    lat = []
    for _ in range(30):
        t1 = time.perf_counter()
        slots[0].enqueue()
        slots[0].sync()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat.sort()
    print(json.dumps({"config": label, "precision": prec, "frames_per_s": round(k * per / el, 1), "passes": k, "frames_per_pass": per,
                      "p50_ms_per_pass": round(lat[len(lat) // 2], 3)}), flush=True)
    for s in slots:
        s.close()


for n in (1, 3):
    run(Single, n, f"{n} independent engine(s) in flight")
for batch, groups in ((3, 1), (3, 2), (2, 2), (2, 3), (4, 1)):
    run(lambda b=batch: Group(b), groups, f"{groups} x batched encoder over {batch} cameras")
