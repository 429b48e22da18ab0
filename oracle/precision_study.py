"""Per-layer precision sensitivity of the SceneSeg-family networks (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Models the engine's fp16 matrix pipe on the CPU oracle: an fp16 x fp16 product is exact in fp32 and the accumulator is
fp32, so a precision mode is fully described by how each operand of each contraction is ROUNDED:

    'f32'  a,        w          (the oracle itself)
    'h'    fp16(a),  fp16(w)    1 MFMA per product   (engine VP_FP16)
    'a2'   hi+lo(a), fp16(w)    2 MFMAs per product  (activation pair x single weight plane)
    'w2'   fp16(a),  hi+lo(w)   2 MFMAs per product
    'x3'   hi+lo(a), hi+lo(w)   3 MFMAs per product  (engine VP_FP16X3; lo*lo dropped, < 2^-22)

The study rounds ONE layer at a time (all the others exact) and reports that layer's contribution to the logits error
under the parity bar |d| <= 1e-3 * max(1, |ref|), then evaluates whole-network assignments.  Output feeds the
engine's mixed-precision plan; results are committed under profiles/.

usage: python -m oracle.precision_study [kind] [--assign FILE]
"""
import json
import sys

import numpy as np
import torch
import torch.nn.functional as TF

from . import nets, pre_post
from .weights import make_state_dict
from autoware_vision_pilot_amd import synthetic


def r16(x):
    return x.to(torch.float16).to(torch.float32)


def r22(x):
    hi = r16(x)
    return hi + r16(x - hi)


ROUND = {"f32": (lambda a: a, lambda w: w), "h": (r16, r16), "a2": (r22, r16), "w2": (r16, r22), "x3": (r22, r22)}


class Proxy:
    """Stands in for torch.nn.functional inside oracle.nets: numbers the contractions in call order."""

    def __init__(self):
        self.modes = {}
        self.default = "f32"
        self.names = []
        self.flops = []
        self.i = 0
        self.record = True

    def __getattr__(self, k):
        return getattr(TF, k)

    def _mode(self, name, flop):
        i = self.i
        self.i += 1
        if self.record:
            self.names.append(name)
            self.flops.append(flop)
        return ROUND[self.modes.get(i, self.default)]

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        ho = (x.shape[2] + 2 * (padding if isinstance(padding, int) else padding[0]) - w.shape[2]) // (
            stride if isinstance(stride, int) else stride[0]) + 1
        wo = ho * x.shape[3] // x.shape[2]
        ra, rw = self._mode(f"conv{w.shape[2]}x{w.shape[3]} {w.shape[1] * groups}->{w.shape[0]} g{groups} @{ho}x{wo}",
                            2.0 * w.numel() * ho * wo)
        return TF.conv2d(ra(x), rw(w), b, stride, padding, dilation, groups)

    def conv_transpose2d(self, x, w, b=None, stride=1):
        ra, rw = self._mode(f"convT {w.shape[0]}->{w.shape[1]} @{x.shape[2]}x{x.shape[3]}",
                            2.0 * w.numel() * x.shape[2] * x.shape[3])
        return TF.conv_transpose2d(ra(x), rw(w), b, stride=stride)

    def linear(self, x, w, b=None):
        ra, rw = self._mode(f"linear {w.shape[1]}->{w.shape[0]}", 2.0 * w.numel())
        return TF.linear(ra(x), rw(w), b)


def run(kind, sd, image, proxy, modes=None, default="f32"):
    proxy.modes = modes or {}
    proxy.default = default
    proxy.i = 0
    out = nets.forward(kind, sd, image)
    proxy.record = False
    return out


def bar(out, ref):
    d = (out - ref).abs()
    return float((d / ref.abs().clamp(min=1.0)).max())


def flips(kind, out, ref):
    if kind in ("sceneseg", "egolanes"):
        return int((out.argmax(1) != ref.argmax(1)).sum())
    return int(((out > 0) != (ref > 0)).sum())


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "sceneseg"
    torch.set_num_threads(8)
    sd = nets.to_torch(make_state_dict(kind, seed=0))
    frame = synthetic.synthetic_frame(720, 1280, seed=1)
    image = torch.from_numpy(pre_post.preprocess(frame))
    proxy = Proxy()
    nets.F = proxy
    try:
        ref = run(kind, sd, image, proxy)
        n = len(proxy.names)
        total = sum(proxy.flops)
        print(f"# {kind}: {n} contractions, {total / 1e9:.1f} GFLOP, logits |max| {float(ref.abs().max()):.2f}")
        rows = []
        for mode in ("h", "a2", "w2"):
            errs = []
            for i in range(n):
                out = run(kind, sd, image, proxy, {i: mode})
                errs.append(bar(out, ref))
            rows.append(errs)
        print("# idx\tname\tGFLOP\terr[h]\terr[a2]\terr[w2]")
        for i in range(n):
            print(f"{i}\t{proxy.names[i]}\t{proxy.flops[i] / 1e9:.3f}\t{rows[0][i]:.2e}\t{rows[1][i]:.2e}\t{rows[2][i]:.2e}")
        for mode in ("h", "a2", "w2", "x3"):
            out = run(kind, sd, image, proxy, {}, default=mode)
            print(f"# all-{mode}: bar {bar(out, ref):.3e}  flips {flips(kind, out, ref)}")
        json.dump(dict(kind=kind, names=proxy.names, gflop=[f / 1e9 for f in proxy.flops], h=rows[0], a2=rows[1],
                       w2=rows[2]), open(f"/tmp/precision_{kind}.json", "w"))
    finally:
        nets.F = TF


if __name__ == "__main__":
    main()
