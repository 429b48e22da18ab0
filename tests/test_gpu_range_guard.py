"""Range guard on the GPU (include/vp_hip.h "RANGE GUARD"): VP_FP16X3 has fp32-class significand but fp16 exponent range, so an
activation beyond 65504 must be reported, not returned as garbage.  SceneSeg with its last encoder stage scaled up by 1e4: f4 passes
65504, the decoder turns inf into NaN, the probe on the logits raises VP_ERR_RANGE from the synchronous call -- per frame, through every
entry point that synchronises (vp_infer, vp_infer_multi for the head that overflows, vp_enqueue + vp_fetch_outputs) -- and a healthy engine
next to it is untouched.  A folded weight beyond the fp16 range fails at load."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hot(sd, factor):
    out = dict(sd)
    k = "Backbone.encoder.8.1.weight"          # BatchNorm scale of features[8] (320 -> 1280): scales f4, the context and the whole decoder
    assert k in sd
    out[k] = sd[k] * np.float32(factor)
    return out


def test_activation_overflow_is_loud(state_dicts, frame720, engines):
    from autoware_vision_pilot_amd import lib, weights as vw

    sd = state_dicts("sceneseg")
    eng = lib.Engine("sceneseg", vw.pack_state_dict(_hot(sd, 1e4)), precision="fp16x3")
    try:
        for _ in range(2):                      # once per offending frame, not once per engine
            with pytest.raises(lib.VpRangeError, match="non-finite"):
                eng.infer(frame720)
        eng.upload_frame(frame720)
        eng.enqueue()
        with pytest.raises(lib.VpRangeError):
            eng.fetch_outputs()
        eng.set_finite_check(False)             # without the probe: silent garbage
        eng.infer(frame720)
        assert not np.isfinite(eng.logits()).all()
    finally:
        eng.close()
    good = engines("sceneseg", "fp16x3")
    good.infer(frame720)
    assert np.isfinite(good.logits()).all()


def test_shared_head_overflow_is_reported_by_infer_multi(state_dicts, frame720):
    from autoware_vision_pilot_amd import lib, synthetic, weights as vw

    sd_seg = state_dicts("sceneseg")
    sd3 = synthetic.share_backbone(dict(state_dicts("scene3d")), "scene3d", sd_seg, "sceneseg")
    k = next(k for k in sd3 if k.endswith("decode_layer_4.weight"))
    sd3[k] = sd3[k] * np.float32(3e4)           # Scene3D's own neck overflows; SceneSeg on the same encoder pass is fine
    base = lib.Engine("sceneseg", vw.pack_state_dict(sd_seg), precision="fp16x3")
    head = lib.Engine("scene3d", vw.pack_state_dict(sd3), precision="fp16x3", base=base)
    try:
        with pytest.raises(lib.VpRangeError):
            base.infer_multi([head], frame720)
        assert np.isfinite(base.logits()).all()
    finally:
        head.close()
        base.close()


def test_weight_beyond_fp16_fails_at_load(state_dicts):
    from autoware_vision_pilot_amd import lib, weights as vw

    sd = dict(state_dicts("sceneseg"))
    k = next(k for k in sd if k.endswith("decode_layer_8.weight"))
    w = sd[k].copy()
    w.flat[5] = np.float32(1e6)
    sd[k] = w
    for prec in ("fp16x3", "fp16"):
        with pytest.raises(lib.VpRangeError, match="fp16 range"):
            lib.Engine("sceneseg", vw.pack_state_dict(sd), precision=prec)


def test_bad_frame_then_good_frame_on_one_replaying_engine(engines, frame720):
    """the probe's verdict is per pass -- a NaN tensor, then a finite frame, on ONE warmed engine whose pass is a replayed
    hipGraph.  The second call must succeed (the probe overwrites every word of its verdict each pass: no clear node to lose in a capture)."""
    from autoware_vision_pilot_amd import lib

    eng = engines("sceneseg", "fp16x3")
    for _ in range(3):                      # eager pass, capture, replay
        eng.infer(frame720)
    good = eng.logits().copy()
    x = eng.input_tensor().copy()
    bad = x.copy()
    bad[0, 0, 7, 9] = np.nan
    for _ in range(2):
        with pytest.raises(lib.VpRangeError, match="non-finite"):
            eng.infer_tensor(bad)
        eng.infer_tensor(x)                 # a finite tensor right behind it: no stale verdict
        assert np.array_equal(eng.logits(), good)
        eng.infer(frame720)
        assert np.array_equal(eng.logits(), good)
