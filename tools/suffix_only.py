import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from autoware_vision_pilot_amd import lib, weights as vw
lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)
from autoware_vision_pilot_amd import synthetic
sd_seg = synthetic.make_state_dict("sceneseg", 0)
sd_3d = synthetic.share_backbone(synthetic.make_state_dict("scene3d", 1), "scene3d", sd_seg, "sceneseg")
b_seg, b_3d = vw.pack_state_dict(sd_seg), vw.pack_state_dict(sd_3d)
frame = synthetic.synthetic_frame(720, 1280, 1)
groups = []
for _ in range(3):
    base = lib.Engine("sceneseg", b_seg, precision="fp16")
    head = lib.Engine("scene3d", b_3d, precision="fp16", base=base)
    base.infer(frame)
    head.enqueue(); head.enqueue(); head.sync()
    groups.append((base, head))
def run(n):
    for i in range(n):
        groups[i % 3][1].enqueue()
    for b, h in groups: h.sync()
run(60)
t0 = time.perf_counter(); run(600); dt = time.perf_counter() - t0
print("scene3d context+neck+head only, 3 in flight: %.1f frames/s = %.1f us per frame" % (600 / dt, dt / 600 * 1e6))
full = [lib.Engine("scene3d", b_3d, precision="fp16") for _ in range(3)]
for e in full:
    e.upload_frame(frame); e.enqueue(); e.enqueue(); e.sync()
def runf(n):
    for i in range(n): full[i % 3].enqueue()
    for e in full: e.sync()
runf(60)
t0 = time.perf_counter(); runf(600); dt = time.perf_counter() - t0
print("scene3d full network, 3 in flight: %.1f frames/s = %.1f us per frame" % (600 / dt, dt / 600 * 1e6))
