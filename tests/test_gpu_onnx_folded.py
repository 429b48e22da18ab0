"""blobs in the form the ONNX reader emits for exporter-folded files (`<conv>.weight` + `<conv>.bias`,
no norm tensors; autoware_vision_pilot_amd/weights.py load_onnx_state_dict) must drive the engine to the same result as
the state_dict blob.  The reader itself is pinned on CPU (tests/test_host_cpu.py, oracle/pin_autodrive_onnx.py); here the
folded form is built with the same arithmetic the exporter's constant folding uses, for both naming schemes the engine
knows: torchvision `X.0` conv + `X.1` BatchNorm (eps 1e-5, SceneSeg family) and the reference's `X.conv` + `X.norm`
(eps 1e-3, AutoDrive)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fold_like_exporter(sd):
    out = dict(sd)
    for k in [k for k in sd if k.endswith(".running_var")]:
        norm = k[:-len(".running_var")]
        head, leaf = norm.rsplit(".", 1)
        conv, eps = (head + ".conv", 1e-3) if leaf == "norm" else (head + ".0", 1e-5)
        assert leaf in ("norm", "1") and conv + ".weight" in sd and conv + ".bias" not in sd, norm
        s = (sd[norm + ".weight"] / np.sqrt(sd[k] + np.float32(eps))).astype(np.float32)
        out[conv + ".weight"] = (sd[conv + ".weight"] * s[:, None, None, None]).astype(np.float32)
        out[conv + ".bias"] = (sd[norm + ".bias"] - sd[norm + ".running_mean"] * s).astype(np.float32)
        for t in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out.pop(norm + "." + t, None)
    assert len(out) < len(sd)
    return out


def test_sceneseg_from_folded_blob(state_dicts, engines, frame720):
    from autoware_vision_pilot_amd import lib, weights as vw

    ref = engines("sceneseg", "fp16x3")
    ref.infer(frame720)
    want = ref.logits().copy()
    eng = lib.Engine("sceneseg", vw.pack_state_dict(fold_like_exporter(state_dicts("sceneseg"))), precision="fp16x3")
    try:
        eng.infer(frame720)
        got = eng.logits()
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
        assert np.array_equal(got.argmax(0), want.argmax(0))
    finally:
        eng.close()


def test_autodrive_from_folded_blob():
    from autoware_vision_pilot_amd import lib, weights as vw
    from oracle import autodrive, pre_post

    sd = autodrive.make_state_dict(5)
    frames = [pre_post.synthetic_frame(1080, 1920, s) for s in (20, 21)]
    outs = []
    for blob in (vw.pack_state_dict(sd), vw.pack_state_dict(fold_like_exporter(sd))):
        eng = lib.Engine("autodrive", blob, precision="fp16x3")
        try:
            eng.infer_pair(frames[0], frames[1])
            outs.append(eng.logits().reshape(3).copy())
        finally:
            eng.close()
    assert np.abs(outs[0] - outs[1]).max() <= 1e-5, outs


def test_engine_from_onnx_path(tmp_path):
    """vp_create on a `*.onnx` model_path (what the reference's backends are given): the library's native reader
    (csrc/onnx_reader.cpp) feeds the engine directly, no Python conversion step."""
    from pbwriter import onnx_model

    from autoware_vision_pilot_amd import lib, weights as vw
    from oracle import autodrive, pre_post

    folded = fold_like_exporter(autodrive.make_state_dict(5))
    path = tmp_path / "AutoDrive.onnx"
    path.write_bytes(onnx_model(folded))
    frames = [pre_post.synthetic_frame(1080, 1920, s) for s in (20, 21)]
    outs = []
    for w in (vw.pack_state_dict(folded), str(path)):
        eng = lib.Engine("autodrive", w, precision="fp16x3")
        try:
            eng.infer_pair(frames[0], frames[1])
            outs.append(eng.logits().reshape(3).copy())
        finally:
            eng.close()
    assert np.array_equal(outs[0], outs[1]), outs
    with pytest.raises(lib.VpError, match="cannot open"):
        lib.Engine("autodrive", str(tmp_path / "missing.onnx"))


def test_incomplete_norm_is_refused(state_dicts):
    """Neither form: norm tensors partly missing and no conv bias -> engine creation fails loudly, no guessing."""
    from autoware_vision_pilot_amd import lib, weights as vw

    sd = dict(state_dicts("sceneseg"))
    k = next(k for k in sd if k.endswith(".1.weight") and k[:-len(".1.weight")] + ".1.running_var" in sd)
    sd.pop(k)
    with pytest.raises(Exception):
        lib.Engine("sceneseg", vw.pack_state_dict(sd), precision="fp16")
