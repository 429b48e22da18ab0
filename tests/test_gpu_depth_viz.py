"""SURVEY.md 8f N4, depth half: vp_visualize_depth_bgr8 against the oracle's restatement of
DepthVisualizationEngine::visualize (depth_visualization_engine.cpp:9-26), bit-exact, on the Scene3D engine."""
import numpy as np
import pytest

from oracle import pre_post

pytestmark = pytest.mark.gpu


def test_depth_visualisation_bit_exact(engines, frame720):
    pytest.importorskip("matplotlib")
    eng = engines("scene3d", "fp16x3")
    eng.infer(frame720)
    for h, w in ((720, 1280), (320, 640), (97, 131)):
        depth = eng.depth_resized(h, w)
        got = eng.visualize_depth(h, w)
        want = pre_post.visualize_depth(depth)
        assert got.shape == (h, w, 3) and got.dtype == np.uint8
        assert np.array_equal(got, want), f"{h}x{w}: {np.count_nonzero((got != want).any(axis=2))} pixels differ"
        assert len(np.unique(got.reshape(-1, 3), axis=0)) > 32          # a real colour ramp, not a constant image
    assert np.array_equal(eng.depth_resized(720, 1280), eng.depth_resized(720, 1280))  # the refactored resize path is stable


def test_depth_visualisation_errors():
    from autoware_vision_pilot_amd import lib, synthetic, weights as vw

    eng = lib.Engine("scene3d", vw.pack_state_dict(synthetic.make_state_dict("scene3d", 1)), precision="fp16")
    try:
        with pytest.raises(lib.VpError, match="has not been run"):
            eng.visualize_depth(720, 1280)
    finally:
        eng.close()
