#!/usr/bin/env python3
"""Developer tool: full per-layer timing table (eager, HIP events) as TSV.  python tools/layer_profile.py [kind] [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from autoware_vision_pilot_amd import lib, weights as vw
lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)
from autoware_vision_pilot_amd import synthetic

kind = sys.argv[1] if len(sys.argv) > 1 else "sceneseg"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16"
import time
if kind == "autodrive":   # BASELINE configs[4]: 1920x1080 frames, fp8-stored weights
    blob, kw, frame = vw.pack_state_dict(synthetic.make_autodrive_state_dict(5)), {"weights_fp8": True}, synthetic.synthetic_frame(1080, 1920, 21)
else:
    seed = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}[kind]
    blob, kw, frame = vw.pack_state_dict(synthetic.make_state_dict(kind, seed)), {}, synthetic.synthetic_frame(720, 1280, 1)
t_create = time.perf_counter()
eng = lib.Engine(kind, blob, precision=prec, **kw)   # vp_create: BN folding, kernel-specific weight layouts, upload, plan
t_create = time.perf_counter() - t_create
eng.upload_frame(frame)
ms = eng.profile_layers(20)
print(f"# {kind} {prec}: eager sum {ms.sum()*1e3:.1f} us over {len(ms)} launches; vp_create (fold, lay out and upload the weights, build the plan) {t_create:.2f} s")
fam = {}
for (n, fl, by), k, t in zip(eng.layers(), eng.layer_kernels(), ms):
    print(f"{n}\t{k}\t{t*1e3:.1f}\t{fl/1e9:.3f}\t{(fl/(t*1e-3)/1e12 if t>0 else 0):.1f}\t{by/1e6:.2f}\t{(by/(t*1e-3)/1e9 if t>0 else 0):.0f}")
    f = fam.setdefault(k, [0.0, 0.0, 0]); f[0] += t; f[1] += fl; f[2] += 1
print("# per kernel family: name, launches, total us, TFLOP/s")
for k, (t, fl, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"# {k}\t{n}\t{t*1e3:.1f}\t{fl/(t*1e-3)/1e12:.1f}")
eng.upload_frame(frame)
for _ in range(10): eng.enqueue()
eng.sync(); eng.timer_begin()
for _ in range(50): eng.enqueue()
print(f"# graph replay {eng.timer_end()/50:.3f} ms/frame")
