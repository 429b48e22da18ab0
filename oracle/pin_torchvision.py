"""Pin the EfficientNet-B0 restatement (oracle/nets.py backbone) against torchvision ITSELF, wherever ``torchvision`` imports
(TEST INFRASTRUCTURE -- see oracle/__init__.py).

The reference's encoder is ``torchvision.models.efficientnet_b0(...).features`` (Models/model_components/backbone.py:9,13-21) and its Python
front-end is ``transforms.ToTensor`` + ``Normalize`` (Models/inference/scene_seg_infer.py:15-20).  torchvision is neither in the reference
tree nor in this image, so row a4 of SURVEY.md section 8 is pinned to a THIRD-PARTY implementation of the same published network (HF
``transformers``: oracle/pin_backbone_hf.py) and a3 to the formula -- "parity unpinned" at the torchvision boundary.

On a machine where torchvision imports this script
  1. builds ``efficientnet_b0(weights=None).features.eval()`` and load_state_dict()s the oracle's seeded state-dict into it UNCHANGED (the
     oracle keeps torchvision's key layout so that real checkpoints load: SURVEY.md 8c),
  2. compares the five taps the reference consumes (features[0], [2], [3], [4], [8]: backbone.py:13-21) with oracle.nets.backbone on a seeded
     input -- same framework, same fp32 kernels, so the bar is 1e-5 relative (summation order inside conv kernels only),
  3. compares ToTensor + Normalize with the oracle's normalize_planes bit for bit,
  4. with --write stores sampled taps in tests/golden/torchvision_pin.npz for tests/test_oracle_golden.py::test_torchvision_pin_fixture.
Here (no torchvision) it prints why it cannot run and exits 0 -- nothing is claimed.

usage: python -m oracle.pin_torchvision [--write]
"""
import os
import sys

import numpy as np
import torch

from . import nets, pre_post
from .weights import PREFIX, make_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "torchvision_pin.npz")
SEED, H, W = 0, 96, 160


def main(argv):
    try:
        import torchvision
        from torchvision import models, transforms
    except Exception as ex:  # noqa: BLE001
        print(f"pin_torchvision: torchvision does not import here ({ex!r}); the encoder stays pinned to HF transformers only (oracle/pin_backbone_hf.py)")
        return 0
    sd = make_state_dict("sceneseg", SEED)
    prefix = PREFIX["sceneseg"]["backbone"]
    feats = models.efficientnet_b0(weights=None).features.eval()
    own = {k[len(prefix):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = feats.load_state_dict(own, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if missing or unexpected:
        print(f"pin_torchvision: key layout differs from torchvision {torchvision.__version__}: missing {missing[:5]}, unexpected {list(unexpected)[:5]}")
        return 1
    x = torch.from_numpy(np.random.default_rng(SEED).standard_normal((1, 3, H, W)).astype(np.float32))
    taps_ref, t = [], x
    with torch.no_grad():
        for i, m in enumerate(feats):
            t = m(t)
            if i in (0, 2, 3, 4, 8):
                taps_ref.append(t)
        taps = nets.backbone(nets.to_torch(sd), prefix, x)
    worst = 0.0
    for a, b in zip(taps, taps_ref):
        worst = max(worst, float((torch.abs(a - b) / torch.clamp(torch.abs(b), min=1.0)).max()))
    img = pre_post.synthetic_frame(320, 640, 3)
    from PIL import Image
    tv = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])(Image.fromarray(img)).numpy()
    ours = pre_post.normalize_planes(img, input_is_bgr=False, planes_rgb=True)
    same_norm = bool(np.array_equal(tv, ours.reshape(tv.shape)))
    print(f"pin_torchvision: torchvision {torchvision.__version__}: five taps within {worst:.2e} (relative); ToTensor + Normalize bit-exact: {same_norm}")
    if worst > 1e-5 or not same_norm:
        return 1
    if "--write" in argv:
        np.savez_compressed(GOLDEN, torchvision_version=np.array(torchvision.__version__), x=x.numpy(),
                            **{f"tap{i}": b.numpy()[0, ::3, ::5, ::7] for i, b in enumerate(taps_ref)})
        print(f"pin_torchvision: wrote {GOLDEN}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
