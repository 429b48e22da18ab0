#!/usr/bin/env python3
"""Developer tool: run a few eager frames so rocprofv3 --pmc can attribute counters per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from autoware_vision_pilot_amd import lib, weights as vw
from oracle import pre_post, weights
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
eng = lib.Engine("sceneseg", vw.pack_state_dict(weights.make_state_dict("sceneseg", 0)), precision=prec)
eng.use_graph(False)
eng.upload_frame(pre_post.synthetic_frame(720, 1280, 1))
for _ in range(3):
    eng.enqueue()
eng.sync()
