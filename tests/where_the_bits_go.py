#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a study script for the GPU box, not a test; its table is committed as profiles/r06_where_the_bits_go.tsv).

VERDICT round 5 item 5b: for the sweep rows where the parity-mode engine is farther from an fp64 evaluation than the fp32 reference is, WHERE along the
network does that happen?  For each such (network, weight seed, frame) the engine's tensors at the oracle's tap points -- the five encoder taps, the
context block's output, the neck's output, the logits -- are compared with an fp64 evaluation, beside the fp32 reference's distance from the same fp64
tensors: max |a - fp64| / max |fp64| per tap.  A stage that loses bits shows as a jump of the engine's column against the reference's.

    python tests/where_the_bits_go.py > gpurun_out/r06_where_the_bits_go.tsv      (on an MI355X; ~1 min of fp64 forwards per row on the host cores)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from autoware_vision_pilot_amd import lib, weights as vw  # noqa: E402
from oracle import nets, pre_post, weights  # noqa: E402

BASE_SEED = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
CASES = [("domainseg", 20, 104, 360, 640, False), ("domainseg", 20, 106, 1080, 1920, False), ("scene3d", 20, 106, 1080, 1920, False),
         ("domainseg", 10, 108, 487, 651, False), ("sceneseg", 0, 101, 720, 1280, True)]
BB = {"sceneseg": "Backbone.encoder.", "scene3d": "PreTrainedBackbone.pretrainedBackBone.encoder.", "domainseg": "DomainSegUpstream.pretrainedBackBone.encoder."}
NECK_LAST = {"sceneseg": "SceneNeck.decode_layer_5", "scene3d": "DepthNeck.decode_layer_5", "domainseg": "DomainSegUpstream.pretrainedNeck.decode_layer_5"}


def backbone_blocks(sd, prefix, image):
    """oracle/nets.py backbone() block by block: {engine tensor name of each MBConv block's output: tensor}"""
    out = {}
    x = nets._cna(sd, prefix + "0", image, stride=2)
    for si, (e, k, st, cin, cout, n) in enumerate(weights.B0_STAGES, start=1):
        for bi in range(n):
            x = nets._mbconv(sd, f"{prefix}{si}.{bi}.block.", x, e, st if bi == 0 else 1, cin if bi == 0 else cout, cout)
            out[f"{prefix}{si}.{bi}.block.{3 if e != 1 else 2}"] = x
    return out


def main():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    print("# engine (fp16x3) and fp32 reference against an fp64 evaluation, per tap: max |a - fp64| / max |fp64|   (tests/where_the_bits_go.py)")
    print("# network\tweight_seed\tframe_seed\ttap\tfp32_reference\tengine\tengine/reference")
    for kind, wseed, fseed, h, w, smooth in CASES:
        sd = weights.make_state_dict(kind, BASE_SEED[kind] + wseed)
        sdt = nets.to_torch(sd)
        frame = pre_post.synthetic_frame(h, w, fseed, smooth=smooth)
        x = torch.from_numpy(pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=False))
        with torch.no_grad():
            o32, i32 = nets.forward(kind, sdt, x, return_intermediates=True)
            o64, i64 = nets.forward(kind, {k: v.double() for k, v in sdt.items()}, x.double(), return_intermediates=True)
        eng = lib.Engine(kind, vw.pack_state_dict(sd), precision="fp16x3")
        try:
            eng.infer(frame)
            names = {n: i for i, (n, _, _, _) in enumerate(eng.tensors())}
            P = BB[kind]
            taps = [("f0", P + "0", i32["feats"][0], i64["feats"][0]), ("f1", P + "2.1.block.3", i32["feats"][1], i64["feats"][1]),
                    ("f2", P + "3.1.block.3", i32["feats"][2], i64["feats"][2]), ("f3", P + "4.2.block.3", i32["feats"][3], i64["feats"][3]),
                    ("f4", P + "8", i32["feats"][4], i64["feats"][4]), ("neck", NECK_LAST[kind], i32["neck"], i64["neck"])]
            rows = []
            if os.environ.get("BITS_BLOCKS", "1") != "0":      # every MBConv block's output between the taps
                with torch.no_grad():
                    b32 = backbone_blocks(sdt, P, x)
                    b64 = backbone_blocks({k: v.double() for k, v in sdt.items()}, P, x.double())
                taps = taps[:1] + [(n[len(P):], n, b32[n], b64[n]) for n in b32] + taps[4:]
            for tag, name, a32, a64 in taps:
                if name not in names:
                    continue
                got = eng.tensor_read(names[name]).astype(np.float64)
                r64 = a64[0].numpy()
                scale = float(np.abs(r64).max())
                rows.append((tag, float(np.abs(a32[0].numpy().astype(np.float64) - r64).max()) / scale, float(np.abs(got - r64).max()) / scale))
            r64 = o64[0].numpy()
            scale = float(np.abs(r64).max())
            rows.append(("logits", float(np.abs(o32[0].numpy().astype(np.float64) - r64).max()) / scale, float(np.abs(eng.logits().astype(np.float64) - r64).max()) / scale))
            for tag, e_ref, e_got in rows:
                print(f"{kind}\t{BASE_SEED[kind] + wseed}\t{fseed}\t{tag}\t{e_ref:.3e}\t{e_got:.3e}\t{e_got / max(e_ref, 1e-30):.2f}", flush=True)
        finally:
            eng.close()


if __name__ == "__main__":
    main()
