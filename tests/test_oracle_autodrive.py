"""AutoDrive oracle vs the committed reference fixture (tests/golden/autodrive.npz, written by oracle/pin_autodrive.py from
the reference's own nn.Module): CPU-only, keeps the restatement pinned on boxes without /root/reference."""
import os

import numpy as np
import pytest
import torch

from oracle import autodrive, pre_post

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "autodrive.npz")


@pytest.fixture(scope="module")
def inputs():
    g = np.load(GOLDEN)
    xs = [torch.from_numpy(pre_post.preprocess(pre_post.synthetic_frame(1080, 1920, int(s)), input_is_bgr=True, planes_rgb=True,
                                               out_h=autodrive.NET_H, out_w=autodrive.NET_W, resize="pil_bilinear")) for s in g["frame_seeds"]]
    return g, xs


def test_spec_matches_reference_inventory():
    assert autodrive.param_count() == 10_581_557  # reference state_dict, num_batches_tracked excluded
    keys = [k for k, _, _ in autodrive.model_spec()]
    assert len(keys) == len(set(keys)) == 118  # 132 state_dict entries - 14 num_batches_tracked


@pytest.mark.parametrize("tag", ["fp32", "fp8"])
def test_oracle_reproduces_reference_fixture(inputs, tag):
    g, (xp, xc) = inputs
    sd = autodrive.make_state_dict(int(g["weight_seed"]))
    if tag == "fp8":
        sd = autodrive.quantize_fp8_e4m3(sd)
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        p5 = autodrive.backbone(sd, xc).numpy().ravel()[g[f"{tag}_p5_idx"]]
        out = np.array([float(v) for v in autodrive.forward(sd, xp, xc)], dtype=np.float32)
    assert np.abs(p5 - g[f"{tag}_p5"]).max() <= 1e-4
    assert np.abs(out - g[f"{tag}_out"]).max() <= 1e-5
    assert out[0] > 0.05 and abs(out[1]) < 0.95  # the fixture is not saturated (ReLU / tanh), so parity is not vacuous


def test_fp8_quantisation_is_e4m3():
    rng = np.random.default_rng(0)
    w = {"x.weight": rng.standard_normal((4, 64)).astype(np.float32)}
    q = autodrive.quantize_fp8_e4m3(w)["x.weight"]
    for r in range(4):
        scale = np.abs(w["x.weight"][r]).max() / 448.0
        v = np.abs(q[r] / scale)
        v = v[v > 0]
        m, _ = np.frexp(v)                       # v = m * 2^e, m in [0.5, 1): e4m3 normals carry 4 significant bits
        assert np.allclose(m * 16, np.round(m * 16), atol=1e-4) or (v < 2.0 ** -6).any()
        assert v.max() <= 448.0 + 1e-3 and np.abs(q[r] - w["x.weight"][r]).max() <= np.abs(w["x.weight"][r]).max() / 14
