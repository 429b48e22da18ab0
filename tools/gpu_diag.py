#!/usr/bin/env python3
"""GPU bring-up diagnostic (developer tool, not part of the product or the test-suite):
per-checkpoint error of the HIP path vs the CPU oracle, and per-layer timing.  Usage on a GPU box:
    python tools/gpu_diag.py [--kinds sceneseg,...] [--precisions fp16,fp16x3] [--profile]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from autoware_vision_pilot_amd import lib, weights as vw  # noqa: E402
from oracle import nets, pre_post, weights  # noqa: E402

SEEDS = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
PREFIX_BB = {k: weights.PREFIX[k]["backbone"] for k in SEEDS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="sceneseg")
    ap.add_argument("--precisions", default="fp16x3,fp16")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    frame = pre_post.synthetic_frame(720, 1280, 1)
    report = {}
    for kind in args.kinds.split(","):
        sd_np = weights.make_state_dict(kind, SEEDS[kind])
        sd = nets.to_torch(sd_np)
        x = torch.from_numpy(pre_post.preprocess(frame))
        t0 = time.time()
        ref, inter = nets.forward(kind, sd, x, return_intermediates=True)
        print(f"[{kind}] oracle forward {time.time() - t0:.2f}s  threads={torch.get_num_threads()}", flush=True)
        ref = ref[0].numpy()
        blob = vw.pack_state_dict(sd_np)
        for prec in args.precisions.split(","):
            t0 = time.time()
            eng = lib.Engine(kind, blob, precision=prec)
            print(f"[{kind}/{prec}] engine built in {time.time() - t0:.2f}s, {len(eng.layers())} launches/frame", flush=True)
            eng.infer(frame)
            pre_ok = np.array_equal(eng.input_tensor(), x.numpy())
            print(f"  preprocess bit-exact: {pre_ok}")
            P = PREFIX_BB[kind]
            names = {P + "0": inter["feats"][0], P + "2.1.block.3": inter["feats"][1], P + "3.1.block.3": inter["feats"][2],
                     P + "4.2.block.3": inter["feats"][3], P + "8": inter["feats"][4]}
            tl = eng.tensors()
            rows = []
            for i, (n, c, h, w) in enumerate(tl):
                r = None
                if n in names:
                    r = names[n][0].numpy()
                elif n.endswith("context_layer_6"):
                    r = inter["ctx"][0].numpy()
                elif n.endswith("decode_layer_5"):
                    r = inter["neck"][0].numpy()
                elif n == "BackboneFeatureFusion":
                    r = inter["deep"][0].numpy()
                if r is None:
                    continue
                g = eng.tensor_read(i)
                e = np.abs(g - r)
                rows.append((n, float(e.max()), float(np.abs(r).max()), float((e / np.maximum(1, np.abs(r))).max())))
            got = eng.logits()
            e = np.abs(got - ref)
            rows.append(("logits", float(e.max()), float(np.abs(ref).max()), float((e / np.maximum(1, np.abs(ref))).max())))
            for n, emax, rmax, rel in rows:
                print(f"  {n:70s} max|err| {emax:.3e}  max|ref| {rmax:.3e}  rel {rel:.3e}")
            if ref.shape[0] > 1:
                rc, gc = pre_post.argmax_classes(ref), pre_post.argmax_classes(got)
                srt = np.sort(ref, axis=0)
                m = (srt[-1] - srt[-2])[rc != gc]
                print(f"  class flips: {(rc != gc).sum()} / {rc.size}; max oracle margin at a flip: {m.max() if m.size else 0:.3e}")
            report[f"{kind}/{prec}"] = rows
            # timing: graph replay, frame resident
            eng.upload_frame(frame)
            for _ in range(5):
                eng.enqueue()
            eng.sync()
            eng.timer_begin()
            for _ in range(args.iters):
                eng.enqueue()
            ms = eng.timer_end() / args.iters
            print(f"  graph replay: {ms:.3f} ms/frame  ({1000.0 / ms:.1f} FPS)")
            if args.profile:
                t = eng.profile_layers(10)
                lay = eng.layers()
                order = np.argsort(-t)
                tot = float(t.sum())
                print(f"  eager per-layer total {tot:.3f} ms; top layers:")
                for i in order[:25]:
                    n, fl, by = lay[i]
                    tf = fl / (t[i] * 1e-3) / 1e12 if t[i] > 0 else 0
                    print(f"    {n:60s} {t[i] * 1000:9.1f} us  {fl / 1e9:8.2f} GFLOP  {tf:8.1f} TFLOP/s  {by / 1e6:8.2f} MB")
            eng.close()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
