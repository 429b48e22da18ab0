#!/usr/bin/env python3
"""Developer tool: a few eager frames so rocprofv3 --pmc can attribute counters per kernel dispatch, preceded by a
CALIBRATION dispatch of known traffic (torch elementwise add over 1 GiB in, 1 GiB out: wide coalesced streaming, larger
than the 256 MiB Infinity Cache) from which tools/pmc_summarize.py derives the FETCH_SIZE / WRITE_SIZE -> bytes factors
(MI355X_MICROARCH.md, HBM: FETCH_SIZE under-reports wide reads 2x on gfx950, WRITE_SIZE is uncalibrated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from autoware_vision_pilot_amd import lib, weights as vw
lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)
from autoware_vision_pilot_amd import synthetic
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
a = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    torch.add(a, 1.0, out=b)
torch.cuda.synchronize()
# the metric configuration's whole plan: SceneSeg + Scene3D on the shared encoder (bench.py's workload), eager launches
sd = synthetic.make_state_dict("sceneseg", 0)
eng = lib.Engine("sceneseg", vw.pack_state_dict(sd), precision=prec)
sd3 = synthetic.share_backbone(synthetic.make_state_dict("scene3d", 1), "scene3d", sd, "sceneseg")
eng3 = lib.Engine("scene3d", vw.pack_state_dict(sd3), precision=prec, base=eng)
eng.use_graph(False)
eng3.use_graph(False)
eng.upload_frame(synthetic.synthetic_frame(720, 1280, 1))
for _ in range(4):
    eng.enqueue()
    eng3.enqueue()
eng.sync()
