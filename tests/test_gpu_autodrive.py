"""AutoDrive (BASELINE configs[4]) on the engine vs the pinned oracle.

Same seeded weights and frames as tests/golden/autodrive.npz (oracle/pin_autodrive.py ran the reference's own
nn.Module on them).  Parity bar: the three outputs within 1e-3 of the fp32 oracle in the fp16x3 mode -- with fp32
weights and with the fp8(e4m3)-dequantised weights of configs[4]; intermediate feature maps within 1e-3 * max(1,|ref|)."""
import os

import numpy as np
import pytest
import torch

from oracle import autodrive, pre_post

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "autodrive.npz")


@pytest.fixture(scope="module")
def setup():
    from autoware_vision_pilot_amd import weights as vw

    g = np.load(GOLDEN)
    frames = [pre_post.synthetic_frame(1080, 1920, int(s)) for s in g["frame_seeds"]]
    sd = autodrive.make_state_dict(int(g["weight_seed"]))
    return g, frames, sd, vw.pack_state_dict(sd)


def _tensor(eng, name):
    for i, (n, c, h, w) in enumerate(eng.tensors()):
        if n == name:
            return eng.tensor_read(i)
    raise KeyError(name)


def test_preprocess_1024x512_bit_exact(setup):
    """The AutoDrive frame path is the reference's (video_visualization.py:29-33): PIL's antialiased BILINEAR to 1024x512, to_tensor,
    normalize -- bit-exact against the oracle's restatement, which the CPU suite pins against Pillow itself; 1920x1080 (a 1.875x /
    2.11x down-scale: 5 / 7 taps per pass), two odd sizes (up-scaling along one axis; a pass that is skipped), and the other two
    resize modes behind vp_set_resize_mode."""
    from autoware_vision_pilot_amd import lib, synthetic

    _, frames, _, blob = setup
    eng = lib.Engine("autodrive", blob, precision="fp16")
    try:
        assert eng.input_hw() == (512, 1024)
        assert eng.resize_mode() == lib.VP_RESIZE_PIL_BILINEAR
        for f in (frames[1], synthetic.synthetic_frame(487, 651, 3), synthetic.synthetic_frame(512, 1300, 4), synthetic.synthetic_frame(2160, 3840, 5)):
            eng.infer(f)
            want = pre_post.preprocess(f, input_is_bgr=True, planes_rgb=True, out_h=512, out_w=1024, resize="pil_bilinear")
            assert np.array_equal(eng.input_tensor(), want), f.shape
        for mode, name in ((lib.VP_RESIZE_PIL_BICUBIC, "pil_bicubic"), (lib.VP_RESIZE_CV_LINEAR, "cv"), (lib.VP_RESIZE_PIL_BILINEAR, "pil_bilinear")):
            eng.set_resize_mode(mode)
            eng.infer(frames[0])
            want = pre_post.preprocess(frames[0], input_is_bgr=True, planes_rgb=True, out_h=512, out_w=1024, resize=name)
            assert np.array_equal(eng.input_tensor(), want), name
    finally:
        eng.close()


@pytest.mark.parametrize("fp8", [False, True])
def test_autodrive_parity_fp16x3(setup, fp8):
    from autoware_vision_pilot_amd import lib

    g, frames, sd, blob = setup
    tag = "fp8" if fp8 else "fp32"
    sdt = {k: torch.from_numpy(v) for k, v in (autodrive.quantize_fp8_e4m3(sd) if fp8 else sd).items()}
    xs = [torch.from_numpy(pre_post.preprocess(f, input_is_bgr=True, planes_rgb=True, out_h=512, out_w=1024, resize="pil_bilinear")) for f in frames]
    with torch.no_grad():
        p5, inter = autodrive.backbone(sdt, xs[1], return_intermediates=True)
        ref = np.array([float(v) for v in autodrive.forward(sdt, xs[0], xs[1])], dtype=np.float32)
    assert np.abs(ref - g[f"{tag}_out"]).max() <= 1e-5  # the oracle itself is pinned to the reference module
    eng = lib.Engine("autodrive", blob, precision="fp16x3", weights_fp8=fp8)
    try:
        eng.infer_pair(frames[0], frames[1])
        got = eng.logits().reshape(3)
        for name, r in (("backbone.p1", inter["p1"]), ("backbone.p2.1.ctx2", inter["p2_ctx"]), ("backbone.p3.1.ctx2", inter["p3_ctx"]),
                        ("backbone.p4.1.ctx2", inter["p4_ctx"]), ("backbone.p5.1.ctx2", inter["p5_ctx"]), ("backbone.p5.2.cv2", inter["sppf"]),
                        ("backbone.p5.3.cv2", p5)):
            t = _tensor(eng, name)
            r = r[0].numpy()
            err = float((np.abs(t - r) / np.maximum(1.0, np.abs(r))).max())
            assert err <= 1e-3, f"{tag} {name}: err {err:.3e}"
        assert np.abs(got - ref).max() <= 1e-3, f"{tag}: got {got}, oracle {ref}"
        # streaming form: frame 0 then frame 1 through plain infer() pairs (f0,f0) then (f0,f1)
        eng2 = lib.Engine("autodrive", blob, precision="fp16x3", weights_fp8=fp8)
        try:
            eng2.infer(frames[0])
            eng2.infer(frames[1])
            assert np.array_equal(eng2.logits().reshape(3), got)
        finally:
            eng2.close()
    finally:
        eng.close()


def test_autodrive_stream_of_four_frames(setup):
    """the shift -> cv2 (alias store into the 512-channel head input) -> head ordering over a stream LONGER than infer_pair's: four
    different frames through plain infer(); every frame's outputs against the oracle on (frame n-1, frame n) -- frame 0 pairs with itself."""
    from autoware_vision_pilot_amd import lib

    g, frames, sd, blob = setup
    stream = frames + [pre_post.synthetic_frame(1080, 1920, 77), pre_post.synthetic_frame(1080, 1920, 78)]
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xs = [torch.from_numpy(pre_post.preprocess(f, input_is_bgr=True, planes_rgb=True, out_h=512, out_w=1024, resize="pil_bilinear")) for f in stream]
    eng = lib.Engine("autodrive", blob, precision="fp16x3")
    try:
        for n, f in enumerate(stream):
            eng.infer(f)
            got = eng.logits().reshape(3).copy()
            with torch.no_grad():
                ref = np.array([float(v) for v in autodrive.forward(sdt, xs[max(n - 1, 0)], xs[n])], dtype=np.float32)
            assert np.abs(got - ref).max() <= 1e-3, f"frame {n}: got {got}, oracle {ref}"
    finally:
        eng.close()


def test_autodrive_fp16_close(setup):
    from autoware_vision_pilot_amd import lib

    g, frames, _, blob = setup
    eng = lib.Engine("autodrive", blob, precision="fp16")
    try:
        eng.infer_pair(frames[0], frames[1])
        got = eng.logits().reshape(3)
        assert np.abs(got - g["fp32_out"]).max() <= 3e-2, (got, g["fp32_out"])
        eng.infer_pair(frames[0], frames[1])
        assert np.array_equal(got, eng.logits().reshape(3))  # graph replay is deterministic
    finally:
        eng.close()


def test_autodrive_ctx_fused_form_matches(setup, vp_opts):
    """Round 5: the CTX expansion matvec + the 1 -> C/2 convolution as ONE launch per stage (kernels_misc.hip ctx_exp_conv1_kernel, 16x16 and 8x8 patches) --
    measured slower than the two launches on this network and therefore opt-in (VP_CTX_FUSE=1), but a shipped kernel: same three scalars
    (the matvec's summation order differs: a few ulp), four launches fewer, a different plan hash."""
    from autoware_vision_pilot_amd import lib

    g, frames, sd, blob = setup
    base = lib.Engine("autodrive", blob, precision="fp16x3")
    try:
        base.infer_pair(frames[0], frames[1])
        want, n0, h0 = base.logits().reshape(3).copy(), len(base.layers()), base.plan_hash()
    finally:
        base.close()
    vp_opts.setenv("VP_CTX_FUSE", "1")
    eng = lib.Engine("autodrive", blob, precision="fp16x3")
    try:
        assert len(eng.layers()) == n0 - 4 and eng.plan_hash() != h0
        assert sum(1 for k in eng.layer_kernels() if k.startswith("ctx_exp_conv1<t")) == 4
        for _ in range(2):
            eng.infer_pair(frames[0], frames[1])
            got = eng.logits().reshape(3)
            assert np.abs(got - want).max() <= 1e-5, (got, want)
            assert np.abs(got - g["fp32_out"]).max() <= 1e-3
    finally:
        eng.close()
