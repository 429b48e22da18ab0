#!/usr/bin/env python3
"""Generate tests/golden/tiny_export.{onnx,npz}: a few-KB model exported with the reference's exporter settings
(Models/exports/convert_pytorch_to_onnx.py:144-154: opset 18, export_params, do_constant_folding) whose module tree has
the shapes that matter to the ONNX reader (autoware_vision_pilot_amd/weights.py load_onnx_state_dict):
  * torchvision-style nesting  encoder.<i>.<j>.block.<k>.{0 conv, 1 BatchNorm}  -> exporter-folded anonymous Conv weights;
  * a Conv+`norm` pair named like the reference's common_layers.Conv (`conv`, `norm`);
  * a plain biased Conv, a ConvTranspose, a biased Linear (names survive), a bias-free Linear (anonymous transposed MatMul);
  * the whole trunk invoked twice (two frames, as AutoDrive does) -> "_1"-suffixed duplicate nodes.
The .npz holds the torch state_dict the file was exported from.  Usage: python tests/golden/make_tiny_onnx.py
(the `onnx` package is absent here; the exporter's onnxscript post-processing hook is stubbed, see
oracle/pin_autodrive_onnx.py)."""
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


class ConvNorm(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.norm = nn.BatchNorm2d(cout, eps=1e-3)

    def forward(self, x):
        return nn.functional.silu(self.norm(self.conv(x)))


class Block(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.block = nn.Sequential(
            nn.Sequential(nn.Conv2d(c, 2 * c, 1, bias=False), nn.BatchNorm2d(2 * c), nn.SiLU()),
            nn.Sequential(nn.Conv2d(2 * c, 2 * c, 3, padding=1, groups=2 * c, bias=False), nn.BatchNorm2d(2 * c), nn.SiLU()),
            nn.Sequential(nn.Conv2d(2 * c, c, 1, bias=False), nn.BatchNorm2d(c)))

    def forward(self, x):
        return x + self.block(x)


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = nn.Sequential(
            nn.Sequential(nn.Conv2d(3, 8, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(8), nn.SiLU()),
            nn.Sequential(Block(8), Block(8)))
        self.stage = ConvNorm(8, 8)
        self.plain = nn.Conv2d(8, 4, 1)
        self.up = nn.ConvTranspose2d(4, 4, 2, 2)
        self.proj = nn.Linear(4, 3, bias=False)
        self.fc = nn.Linear(3, 2)

    def trunk(self, x):
        return self.up(self.plain(self.stage(self.encoder(x))))

    def forward(self, a, b):
        f = self.trunk(a) + self.trunk(b)
        return self.fc(self.proj(f.mean(dim=(2, 3))))


def main():
    torch.manual_seed(7)
    m = Tiny().eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 2.0)
    from oracle.pin_autodrive_onnx import export_like_reference

    x = torch.randn(1, 3, 8, 8)
    export_like_reference(m, (x, x.flip(3)), os.path.join(HERE, "tiny_export.onnx"), ["a", "b"], ["output"])
    sd = {k: v.numpy() for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
    np.savez_compressed(os.path.join(HERE, "tiny_export.npz"), **sd)
    print("written", os.path.getsize(os.path.join(HERE, "tiny_export.onnx")), "bytes,", len(sd), "tensors")


if __name__ == "__main__":
    main()
