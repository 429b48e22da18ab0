#!/usr/bin/env python3
"""Developer tool: the same network built twice in two processes that differ in ONE environment knob (e.g. VP_MBCONV_BACK=0 / 1), every
activation tensor both engines have in common compared: which layer a numerical difference enters at.
    python tools/cmp_env_tensors.py scene3d VP_MBCONV_BACK"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def child(kind, out):
    from autoware_vision_pilot_amd import lib, weights as vw
    lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)
    from oracle import pre_post, weights
    seeds = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
    blob = vw.pack_state_dict(weights.make_state_dict(kind, seeds[kind]))
    eng = lib.Engine(kind, blob, precision="fp16x3")
    eng.infer(pre_post.synthetic_frame(720, 1280, 1))
    d = {}
    for i, (n, c, h, w) in enumerate(eng.tensors()):
        if "encoder" in n:
            d[n] = eng.tensor_read(i)
    np.savez(out, **d)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    kind, knob = sys.argv[1], sys.argv[2]
    outs = []
    for v in ("0", "1"):
        o = f"/tmp/cmp_{knob}_{v}.npz"
        subprocess.check_call([sys.executable, __file__, "--child", kind, o], env=dict(os.environ, **{knob: v}))
        outs.append(np.load(o))
    a, b = outs
    for n in a.files:
        if n in b.files:
            x, y = a[n], b[n]
            print(f"{n:60s} max|a| {np.abs(x).max():9.3e}  max|a-b| {np.abs(x - y).max():9.3e}  rel {np.abs(x - y).max() / max(1e-30, np.abs(x).max()):9.3e}")
