#!/usr/bin/env python3
"""Developer tool: per-stage comparison of the AutoDrive engine against the oracle (head stages)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from autoware_vision_pilot_amd import lib, weights as vw
from oracle import autodrive, pre_post
frames = [pre_post.synthetic_frame(1080, 1920, s) for s in (20, 21)]
sd = autodrive.make_state_dict(5)
sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
xs = [torch.from_numpy(pre_post.preprocess(f, True, True, 512, 1024, resize="pil_bilinear")) for f in frames]
with torch.no_grad():
    fp, fc = autodrive.backbone(sdt, xs[0]), autodrive.backbone(sdt, xs[1])
    x = torch.cat([fp, fc], 1)
    st = {"head.cat": x}
    for i in (1, 2, 3):
        x = F.silu(F.conv2d(x, sdt[f"head.conv_{i}.weight"], sdt[f"head.conv_{i}.bias"], padding=1))
        st[f"head.conv_{i}"] = x
eng = lib.Engine("autodrive", vw.pack_state_dict(sd), precision="fp16x3")
eng.infer_pair(frames[0], frames[1])
names = {n: i for i, (n, c, h, w) in enumerate(eng.tensors())}
for k, r in st.items():
    t = eng.tensor_read(names[k]); r = r[0].numpy()
    print(k, t.shape, r.shape, "err", float((np.abs(t - r) / np.maximum(1, np.abs(r))).max()), "halves", float(np.abs(t[:256]-r[:256]).max()) if k=="head.cat" else "")
print("engine", eng.logits().reshape(3), "oracle", [float(v) for v in autodrive.forward(sdt, xs[0], xs[1])])
