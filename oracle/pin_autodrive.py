#!/usr/bin/env python3
"""Pin oracle/autodrive.py against the reference's OWN AutoDrive nn.Module and (re)generate tests/golden/autodrive.npz.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Runs only where /root/reference exists (the build container); the GPU box
uses the committed fixture.  Usage:  python oracle/pin_autodrive.py

Executed from the reference (imported, never copied): Models/model_components/autodrive/{autodrive_network,
autodrive_backbone,autodrive_head}.py and common_layers.py.  The whole network is reference code here (no third-party
backbone), so AutoDrive is fully PINNED: backbone P5 map, the three outputs, with fp32 weights and with the
fp8(e4m3)-dequantised weights of BASELINE configs[4]."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import autodrive, pre_post  # noqa: E402

SEED = 5
FRAME_SEEDS = (20, 21)
GOLDEN = os.path.join(ROOT, "tests", "golden", "autodrive.npz")


def reference_preprocess(frame_bgr):
    """Models/visualizations/AutoDrive/video_visualization.py:29-33 with the SAME library calls where they exist here: BGR -> RGB,
    PIL Image.resize((1024, 512), Image.BILINEAR) (Pillow itself), torchvision's to_tensor (u8 -> fp32 / 255, HWC -> CHW) and
    normalize ((t - mean) / std, fp32) spelled out in torch (torchvision and cv2 are not installed; both steps are one-liners
    in torchvision/transforms/_functional_tensor.py)."""
    from PIL import Image

    rgb = np.ascontiguousarray(frame_bgr[..., ::-1])
    pil = Image.fromarray(rgb).resize((autodrive.NET_W, autodrive.NET_H), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(3, 1, 1)
    return t.sub(mean).div(std)[None].numpy()


def frames():
    """Two synthetic 1080p BGR frames -> network inputs 1x3x512x1024 by the reference's own frame path (PIL's antialiased BILINEAR);
    the oracle's restatement of that path (pre_post.preprocess(resize="pil_bilinear")) must reproduce it bit for bit."""
    out = []
    for s in FRAME_SEEDS:
        f = pre_post.synthetic_frame(1080, 1920, s)
        x = reference_preprocess(f)
        mine = pre_post.preprocess(f, input_is_bgr=True, planes_rgb=True, out_h=autodrive.NET_H, out_w=autodrive.NET_W, resize="pil_bilinear")
        assert np.array_equal(x, mine), "oracle frame path != PIL + to_tensor + normalize"
        out.append(x)
    return out


def main():
    from Models.model_components.autodrive.autodrive_network import AutoDrive

    torch.manual_seed(0)
    xp, xc = (torch.from_numpy(v) for v in frames())
    fix = {}
    for tag, quant in (("fp32", False), ("fp8", True)):
        sd_np = autodrive.make_state_dict(SEED)
        if quant:
            sd_np = autodrive.quantize_fp8_e4m3(sd_np)
        sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
        m = AutoDrive().eval()
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
        with torch.no_grad():
            ref_p5 = m.backbone(xc)
            ref = m(xp, xc)
            ora_p5 = autodrive.backbone(sd, xc)
            ora = autodrive.forward(sd, xp, xc)
        e_p5 = float((ora_p5 - ref_p5).abs().max())
        e_out = max(float((a - b).abs().max()) for a, b in zip(ora, ref))
        print(tag, "max |oracle - reference|: P5", f"{e_p5:.3e}", "outputs", f"{e_out:.3e}", "| P5 std", float(ref_p5.std()),
              "outputs", [float(v) for v in ref])
        assert e_p5 <= 1e-4 * max(1.0, float(ref_p5.abs().max())) and e_out <= 1e-5, (e_p5, e_out)
        idx = np.random.default_rng(99).choice(ref_p5.numel(), 4096, replace=False)
        fix[f"{tag}_p5_idx"] = idx.astype(np.int64)
        fix[f"{tag}_p5"] = ref_p5.numpy().ravel()[idx].astype(np.float32)
        fix[f"{tag}_out"] = np.array([float(v) for v in ref], dtype=np.float32)
    fix["weight_seed"] = np.int64(SEED)
    fix["frame_seeds"] = np.array(FRAME_SEEDS, dtype=np.int64)
    np.savez_compressed(GOLDEN, **fix)
    print("written", GOLDEN)


if __name__ == "__main__":
    main()
