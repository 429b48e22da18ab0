#!/usr/bin/env python3
"""Pin the ONNX reader (autoware_vision_pilot_amd/weights.py load_onnx_state_dict) against a file made
by the reference's OWN module and the reference's OWN exporter settings.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Runs only where /root/reference exists (the build container).
Usage:  python oracle/pin_autodrive_onnx.py

Steps: build the reference `AutoDrive` nn.Module (Models/model_components/autodrive/autodrive_network.py) with the seeded
weights of oracle/autodrive.py, export it as Models/exports/convert_pytorch_to_onnx.py:102-126 does (opset 18,
export_params, do_constant_folding, the same input/output names), read the file back with the package's reader and check
  1. the key set is the state_dict's, minus every `*.norm.*` tensor, plus one `*.conv.bias` per folded Conv;
  2. un-folded tensors are bit-identical, folded ones equal w*g/sqrt(v+eps), beta - mean*g/sqrt(v+eps) to 1e-6;
  3. the oracle run on the converted dict reproduces the reference module's three outputs to 1e-5;
  4. the library's native reader (csrc/onnx_reader.cpp via vp_convert_onnx) returns the same tensors, bit for bit.
The `onnx` Python package is absent in this image; the TorchScript exporter only needs it for a post-processing hook
that does not apply here (no onnxscript functions), so that hook is stubbed for the export call."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from autoware_vision_pilot_amd import weights  # noqa: E402
from oracle import autodrive, pin_autodrive  # noqa: E402


def export_like_reference(module, inputs, path, input_names, output_names):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    keep = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    try:
        torch.onnx.export(module, inputs, path, export_params=True, opset_version=18, do_constant_folding=True,
                          input_names=input_names, output_names=output_names, dynamo=False)
    finally:
        onnx_proto_utils._add_onnxscript_fn = keep


def main():
    from Models.model_components.autodrive.autodrive_network import AutoDrive

    ref_sd = autodrive.make_state_dict(pin_autodrive.SEED)
    m = AutoDrive().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ref_sd.items()}, strict=False)
    xp, xc = (torch.from_numpy(v) for v in pin_autodrive.frames())
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "autodrive.onnx")
        export_like_reference(m, (xp, xc), path, ["image_prev", "image_curr"], ["distance", "curvature", "flag_logit"])
        print("exported", os.path.getsize(path), "bytes")
        got = weights.load_onnx_state_dict(path)
        from autoware_vision_pilot_amd import lib  # the library's native reader (csrc/onnx_reader.cpp), host only

        native = weights.unpack_blob(open(lib.convert_onnx(path, os.path.join(d, "autodrive.vpw")), "rb").read())
        assert set(native) == set(got) and all(np.array_equal(native[k], got[k]) for k in got), "native reader != python reader"

    convs = [k[:-len(".conv.weight")] for k in ref_sd if k.endswith(".conv.weight")]
    want = {k for k in ref_sd if ".norm." not in k} | {p + ".conv.bias" for p in convs}
    assert set(got) == want, (sorted(set(got) - want)[:5], sorted(want - set(got))[:5])
    worst = 0.0
    for p in convs:
        g, b, mu, v = (ref_sd[p + ".norm." + t] for t in ("weight", "bias", "running_mean", "running_var"))
        s = g / np.sqrt(v + np.float32(autodrive.BN_EPS))
        w, bb = ref_sd[p + ".conv.weight"] * s[:, None, None, None], b - mu * s
        worst = max(worst, float(np.abs(got[p + ".conv.weight"] - w).max() / np.abs(w).max()),
                    float(np.abs(got[p + ".conv.bias"] - bb).max() / max(1e-9, float(np.abs(bb).max()))))
    assert worst <= 1e-6, worst
    for k in want:
        if not any(k.startswith(p + ".conv.") for p in convs):
            assert np.array_equal(got[k], ref_sd[k]), k
    with torch.no_grad():
        ref = m(xp, xc)
        ora = autodrive.forward({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in got.items()}, xp, xc)
    err = max(float((a - b).abs().max()) for a, b in zip(ora, ref))
    print(f"{len(got)} tensors ({len(convs)} folded convs, worst fold error {worst:.2e}); "
          f"oracle on the converted weights vs reference module: {err:.3e}")
    assert err <= 1e-5, err


if __name__ == "__main__":
    main()
