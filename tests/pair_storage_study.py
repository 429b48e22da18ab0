#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a study script, not a test: ~6 min of fp64 forwards on 8 cores; its table is committed as profiles/r06_pair_storage_study.tsv).

VERDICT round 5 item 5b: in the parity sweep a few passes on ill-scaled weight seeds are FARTHER from an fp64 evaluation than the fp32 reference is
(worst ratio 1.87: DomainSeg seed 23, profiles/r06_parity_sweep.tsv).  Where do those bits go?  The engine computes every product exactly enough
(three fp16 MFMAs per product, fp32 accumulation like the reference) -- what differs from fp32 is what a TENSOR CAN HOLD: a (hi, lo) fp16 pair carries
22 significand bits (unit round-off 2^-23 relative) against fp32's 24 (2^-25), in every activation tensor and every weight.

This script isolates exactly that: the network is evaluated in fp64 (exact arithmetic for this purpose) with ONE change at a time --
    f32-storage : every layer output and every weight rounded to fp32                      (what an ideal fp32 engine could at best deliver)
    pair-storage: every layer output and every weight rounded to a (hi, lo) fp16 pair      (what an ideal fp16x3 engine could at best deliver)
and the distance of each from the unmodified fp64 evaluation is printed beside the distances the sweep measured for the real fp32 reference and the
real engine.  If pair-storage alone is about as far from fp64 as the engine is, the ratio is a property of the number format, not of a kernel.

    python tests/pair_storage_study.py [kind wseed fseed h w smooth] > profiles/r06_pair_storage_study.tsv
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import nets, pre_post, weights  # noqa: E402

BASE_SEED = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
CASES = [("domainseg", 20, 104, 360, 640, False), ("domainseg", 20, 106, 1080, 1920, False), ("scene3d", 20, 106, 1080, 1920, False),
         ("domainseg", 10, 108, 487, 651, False)]


def round_f32(t):
    return t.float().double()


def round_pair(t):
    hi = t.half().double()
    lo = (t - hi).half().double()
    return hi + lo


def round_weight_rows(v, rnd):
    """the engine's weight format: per OUTPUT ROW a power-of-two prescale that puts the row maximum into [2^13, 2^14), then `rnd`, then the exact inverse
    (engine_internal.hpp prescale_exp / row_prescale) -- a ConvTranspose2d weight [cin][cout][2][2] has its output channels on axis 1"""
    if rnd is round_f32 or v.dim() < 2:
        return rnd(v)
    is_convt = v.dim() == 4 and tuple(v.shape[2:]) == (2, 2)
    w = v.transpose(0, 1) if is_convt else v
    flat = w.reshape(w.shape[0], -1)
    amax = flat.abs().amax(dim=1).clamp_min(1e-300)
    sexp = torch.floor(13.0 - torch.log2(amax))
    sc = torch.pow(torch.tensor(2.0, dtype=torch.float64), sexp).reshape(-1, *([1] * (w.dim() - 1)))
    out = rnd(w * sc) / sc
    return out.transpose(0, 1) if is_convt else out


def gelu_engine(x64):
    """the parity mode's GELU as the kernels evaluate it (csrc/common.hpp gelu_exact: Abramowitz-Stegun 7.1.26 in fp32, |erf error| <= 1.5e-7), on fp32
    inputs, returned in fp64 -- torch's fp32 exp / reciprocal stand in for v_exp_f32 / v_rcp_f32 (1 ulp class)"""
    x = x64.float()
    u = x.abs() * np.float32(0.70710678118654752440)
    t = 1.0 / (np.float32(0.3275911) * u + 1.0)
    poly = t * np.float32(1.061405429) - np.float32(1.453152027)
    poly = t * poly + np.float32(1.421413741)
    poly = t * poly - np.float32(0.284496736)
    poly = t * poly + np.float32(0.254829592)
    P = poly * t * torch.exp(-u * u)
    hx = 0.5 * x
    return torch.where(x >= 0, x - hx * P, hx * P).double()


def forward_with_storage(kind, sd64, x64, rnd, gelu=None):
    """nets.forward in fp64 with every convolution / linear / transposed-convolution OUTPUT passed through `rnd` and every parameter stored the way the
    engine stores it (weights: row prescale + `rnd`; biases and BatchNorm statistics stay fp32 in the engine)"""
    sd = {k: (round_weight_rows(v, rnd) if v.is_floating_point() and v.dim() >= 2 else (round_f32(v) if v.is_floating_point() else v)) for k, v in sd64.items()}
    real = {n: getattr(F, n) for n in ("conv2d", "conv_transpose2d", "linear", "gelu", "silu")}
    try:
        for n, fn in real.items():
            setattr(F, n, (lambda f: (lambda *a, **k: rnd(f(*a, **k))))(fn))
        if gelu is not None:
            F.gelu = lambda t, *a, **k: rnd(gelu(t))
        return nets.forward(kind, sd, rnd(x64))[0].numpy()
    finally:
        for n, fn in real.items():
            setattr(F, n, fn)


def main():
    cases = CASES
    if len(sys.argv) >= 7:
        cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6] not in ("0", "False"))]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    print("# what the STORAGE FORMAT alone costs: fp64 evaluation with every tensor (activations after each linear operator / activation function, and the "
          "weights) rounded to fp32 or to a (hi, lo) fp16 pair; max |a - fp64| / max(1, |fp64|) over the logits (tests/pair_storage_study.py)")
    print("# network\tweight_seed\tframe\tframe_seed\tfp32_reference_vs_fp64\tf32_storage_vs_fp64\tpair_storage_vs_fp64\tf32_storage+engine_GELU\tpair_storage+engine_GELU\t(pair+GELU)/fp32_reference")
    for kind, wseed, fseed, h, w, smooth in cases:
        sd = weights.make_state_dict(kind, BASE_SEED[kind] + wseed)
        sdt = nets.to_torch(sd)
        frame = pre_post.synthetic_frame(h, w, fseed, smooth=smooth)
        x = torch.from_numpy(pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=(kind == "egolanes")))
        with torch.no_grad():
            ref32 = nets.forward(kind, sdt, x)[0].numpy().astype(np.float64)
            sd64 = {k: v.double() for k, v in sdt.items()}
            r64 = nets.forward(kind, sd64, x.double())[0].numpy()
            s32 = forward_with_storage(kind, sd64, x.double(), round_f32)
            spair = forward_with_storage(kind, sd64, x.double(), round_pair)
            s32g = forward_with_storage(kind, sd64, x.double(), round_f32, gelu=gelu_engine)
            spairg = forward_with_storage(kind, sd64, x.double(), round_pair, gelu=gelu_engine)
        rel = lambda a: float((np.abs(a - r64) / np.maximum(1.0, np.abs(r64))).max())
        e_ref, e32, epair, e32g, epairg = rel(ref32), rel(s32), rel(spair), rel(s32g), rel(spairg)
        print(f"{kind}\t{BASE_SEED[kind] + wseed}\t{h}x{w}\t{fseed}\t{e_ref:.3e}\t{e32:.3e}\t{epair:.3e}\t{e32g:.3e}\t{epairg:.3e}\t{epairg / max(e_ref, 1e-30):.2f}", flush=True)


if __name__ == "__main__":
    main()
