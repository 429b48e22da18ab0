"""Parity of the full HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star: "class-index maps bit-exact, float tensors within 1e-3"):
  * fp16x3 (parity mode): |a-b| <= 1e-3 * max(1,|b|) on every checked tensor; class maps must agree on every
    pixel whose oracle top-2 margin is >= 2e-3 (pixels inside the float tolerance band can legitimately flip and
    are counted and bounded).
  * fp16 (the reference's "fp16" configuration, fp16 tensors in HBM) is a labelled NON-PARITY option: it does not meet the
    1e-3 bar on these weights (measured 2.2e-2, 154 class flips of 204 800) and no parity claim rests on it -- bench.py reports
    it beside the parity-mode value, never as it.  What is checked here is a REGRESSION bound on what plain fp16 delivers:
    max |a-b| <= FP16_TOL * max|b| per tensor; class-map agreement >= 99.5 %, every flip inside the fp16 error band.
Integer/byte stages (preprocess, decode, nearest resize) are compared bit-exactly.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["sceneseg", "scene3d", "domainseg", "egolanes"]
PREFIX_BB = {"sceneseg": "Backbone.encoder.", "scene3d": "PreTrainedBackbone.pretrainedBackBone.encoder.",
             "domainseg": "DomainSegUpstream.pretrainedBackBone.encoder.", "egolanes": "BEVBackbone.encoder."}
FP16_TOL = 3e-2


def _tensor_by_name(eng, name):
    for i, (n, c, h, w) in enumerate(eng.tensors()):
        if n == name:
            return eng.tensor_read(i)
    raise KeyError(name)


def _rel(a, b):
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())


@pytest.mark.parametrize("size", [(720, 1280), (320, 640), (487, 651), (1080, 1920), (200, 300), (2160, 3840), (16, 16), (2, 3)])
def test_preprocess_bit_exact(engines, size):
    from autoware_vision_pilot_amd import lib
    from oracle import pre_post

    eng = engines("sceneseg", "fp16")
    frame = pre_post.synthetic_frame(size[0], size[1], 5, smooth=False)
    for pix, planes in [(lib.VP_BGR8, lib.VP_PLANES_BGR), (lib.VP_BGR8, lib.VP_PLANES_RGB), (lib.VP_RGB8, lib.VP_PLANES_RGB)]:
        eng.set_input_format(pix, planes)
        for form, name in ((lib.VP_NORM_TORCHVISION, "torchvision"), (lib.VP_NORM_OPENCV, "opencv")):   # q / 255 (Python API) and q * fl(1/255) (C++ front-ends)
            eng.set_norm_form(form)
            eng.infer(frame)
            got = eng.input_tensor()
            ref = pre_post.preprocess(frame, input_is_bgr=(pix == lib.VP_BGR8), planes_rgb=(planes == lib.VP_PLANES_RGB), norm_form=name)
            assert np.array_equal(got, ref), f"{size} fmt={pix} planes={planes} {name}: {np.abs(got - ref).max()}"
    eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_BGR)
    eng.set_norm_form(lib.VP_NORM_TORCHVISION)


def test_preprocess_strided_rows(engines):
    from oracle import pre_post

    eng = engines("sceneseg", "fp16")
    big = pre_post.synthetic_frame(360, 700, 9, smooth=False)
    view = big[:, :640]  # row stride 2100 bytes > 3*640
    assert not view.flags["C_CONTIGUOUS"]
    import ctypes as C
    lib = eng._lib
    rc = lib.vp_infer(eng._h, view.ctypes.data_as(C.c_void_p), 360, 640, view.strides[0])
    assert rc == 0
    assert np.array_equal(eng.input_tensor(), pre_post.preprocess(np.ascontiguousarray(view)))


@pytest.mark.parametrize("kind", KINDS)
def test_network_parity_fp16x3(engines, oracle_runs, frame720, kind):
    eng = engines(kind, "fp16x3")
    ref_out, inter, x = oracle_runs(kind)
    eng.infer(frame720)
    assert np.array_equal(eng.input_tensor(), x)
    P = PREFIX_BB[kind]
    bb = {P + "0": 0, P + "2.1.block.3": 1, P + "3.1.block.3": 2, P + "4.2.block.3": 3, P + "8": 4}
    for name, fi in bb.items():
        got = _tensor_by_name(eng, name)
        assert _rel(got, inter["feats"][fi][0].numpy()) <= 1e-3, f"{kind} {name}"
    got = eng.logits()
    assert got.shape == ref_out.shape
    assert _rel(got, ref_out) <= 1e-3, f"{kind} logits rel err {_rel(got, ref_out):.3e}"


@pytest.mark.parametrize("kind", KINDS)
def test_network_parity_fp16(engines, oracle_runs, frame720, kind):
    eng = engines(kind, "fp16")
    ref_out, _, _ = oracle_runs(kind)
    eng.infer(frame720)
    got = eng.logits()
    err = np.abs(got - ref_out).max() / np.abs(ref_out).max()
    assert err <= FP16_TOL, f"{kind}: fp16 max err / max|ref| = {err:.3e}"


def test_class_map_parity(engines, oracle_runs, frame720):
    from autoware_vision_pilot_amd import lib
    from oracle import pre_post

    ref_out, _, _ = oracle_runs("sceneseg")
    ref_cls = pre_post.argmax_classes(ref_out)
    srt = np.sort(ref_out, axis=0)
    margin = srt[-1] - srt[-2]
    for precision, band, min_agree in (("fp16x3", None, 1.0), ("fp16", 0.25, 0.995)):
        eng = engines("sceneseg", precision)
        eng.set_decode_mode(lib.VP_DECODE_CLASS_INDEX)
        eng.infer(frame720)
        cls = eng.mask().astype(np.int64)
        # decode itself is integer work: bit-exact against the oracle decode of the SAME logits
        assert np.array_equal(cls, pre_post.argmax_classes(eng.logits()))
        flips = cls != ref_cls
        if band is None:
            # parity mode, the sweep's strict rule (tests/test_gpu_parity_sweep.py): ZERO flips except at pixels whose oracle decision margin is at
            # most twice the MEASURED maximum logit error of this pass -- a tie inside the float tolerance (measured on this frame: 0 flips)
            band = 2.0 * float(np.abs(eng.logits() - ref_out).max()) + 1e-12
            assert band <= 2e-3
        else:
            assert flips.mean() <= 1 - min_agree, f"{precision}: {flips.sum()} flips"
        assert (margin[flips] <= band).all(), f"{precision}: {flips.sum()} flips, one outside the tolerance band {band:.3e}: max margin {margin[flips].max():.3e}"
        eng.set_decode_mode(lib.VP_DECODE_SEG_MASK)
        eng.infer(frame720)
        assert np.array_equal(eng.mask(), pre_post.seg_mask_u8(eng.logits()))


@pytest.mark.parametrize("kind", KINDS)
def test_golden_fixture(engines, frame720, kind):
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"full_{kind}.npz"))
    eng = engines(kind, "fp16x3")
    eng.infer(frame720)
    got = eng.logits().ravel()[g["samples_idx"]]
    assert _rel(got, g["samples"]) <= 1e-3


def test_decode_variants_bit_exact(engines, frame720):
    from autoware_vision_pilot_amd import lib
    from oracle import pre_post

    eng = engines("egolanes", "fp16x3")
    eng.infer(frame720)
    lg = eng.logits()
    assert lg.shape == (3, 80, 160)
    assert np.array_equal(eng.mask(), pre_post.egolanes_priority_mask(lg))
    eng = engines("domainseg", "fp16x3")
    eng.infer(frame720)
    assert np.array_equal(eng.mask(), pre_post.seg_mask_u8(eng.logits()))


def test_resized_outputs(engines, frame720):
    from oracle import pre_post

    eng = engines("sceneseg", "fp16x3")
    eng.infer(frame720)
    for (h, w) in [(720, 1280), (360, 640), (1080, 1920), (333, 777)]:
        assert np.array_equal(eng.mask_resized(h, w), pre_post.resize_nearest_u8(eng.mask(), h, w))
    eng = engines("scene3d", "fp16x3")
    eng.infer(frame720)
    d = eng.logits()[0]
    for (h, w) in [(720, 1280), (333, 777)]:
        assert np.array_equal(eng.depth_resized(h, w), pre_post.resize_bilinear_f32(d, h, w))


def test_graph_replay_is_deterministic(engines, frame720):
    eng = engines("sceneseg", "fp16")
    eng.use_graph(False)
    eng.infer(frame720)
    a = eng.logits()
    eng.use_graph(True)
    eng.infer(frame720)
    b = eng.logits()
    eng.infer(frame720)
    c = eng.logits()
    assert np.array_equal(a, b) and np.array_equal(b, c)


def test_two_engines_interleaved(engines, frame720):
    e1, e2 = engines("sceneseg", "fp16"), engines("egolanes", "fp16")
    e1.infer(frame720)
    r1 = e1.logits()
    e1.upload_frame(frame720)
    e2.upload_frame(frame720)
    for _ in range(3):
        e1.enqueue()
        e2.enqueue()
    e1.fetch_outputs()
    e2.fetch_outputs()
    assert np.array_equal(e1.logits(), r1)
    assert e2.logits().shape == (3, 80, 160)


def test_python_operator_api(state_dicts, oracle_runs):
    """B3: same names / call convention / return types / errors as Models/inference/*_infer.py."""
    from PIL import Image

    from autoware_vision_pilot_amd import infer, weights as vw
    from oracle import nets, pre_post
    import torch

    img = Image.fromarray(pre_post.synthetic_frame(320, 640, 3))  # RGB 640x320
    x = torch.from_numpy(infer.image_loader(img))
    with pytest.raises(ValueError):
        infer.SceneSegNetworkInfer("")
    for cls, kind in ((infer.SceneSegNetworkInfer, "sceneseg"), (infer.Scene3DNetworkInfer, "scene3d"),
                      (infer.DomainSegNetworkInfer, "domainseg"), (infer.EgoLanesNetworkInfer, "egolanes")):
        net = cls(vw.pack_state_dict(state_dicts(kind)), precision="fp16x3")
        out = net.inference(img)
        ref = nets.forward(kind, nets.to_torch(state_dicts(kind)), x)[0].numpy()
        if kind == "sceneseg":
            assert out.dtype == np.int64 and out.shape == (320, 640)
            ref_cls = pre_post.argmax_classes(ref)
            srt = np.sort(ref, axis=0)
            flips = out != ref_cls
            assert flips.mean() < 5e-4 and ((srt[-1] - srt[-2])[flips] < 2e-3).all()
            with pytest.raises(ValueError):
                net.inference(img.resize((320, 160)))
            # inference_resized = the scripts' `image.resize((640, 320))` + `inference(image)` in one device pass; the Pillow mode it needs is
            # BORROWED: it stays set between consecutive calls (no per-frame table rebuild / recapture) and the caller's own mode is back the
            # moment the caller looks
            from autoware_vision_pilot_amd import lib
            big = Image.fromarray(pre_post.synthetic_frame(720, 1280, 5))
            r1 = net.inference_resized(big)
            h1 = net.model.plan_hash()
            r2 = net.inference_resized(big)
            assert np.array_equal(r1, r2) and net.model.plan_hash() == h1
            via_pil = net.inference(big.resize((640, 320)))
            assert (r1 != via_pil).mean() < 5e-4
            assert net.model.resize_mode() == lib.VP_RESIZE_CV_LINEAR
            r3 = net.inference_resized(big)
            assert np.array_equal(r1, r3)
        elif kind == "scene3d":
            assert out.dtype == np.float32 and out.shape == (320, 640, 1)
            assert _rel(out[..., 0], ref[0]) <= 1e-3
        elif kind == "domainseg":
            assert out.shape == (320, 640, 1) and set(np.unique(out)) <= {0.0, 1.0}
            flips = out[..., 0] != pre_post.binary_float(ref)[..., 0]
            assert flips.mean() < 5e-4 and (np.abs(ref[0])[flips] < 2e-3).all()
        else:
            assert out.shape == (3, 80, 160) and _rel(out, ref) <= 1e-3
        net.model.close()


def test_errors_are_loud(engines):
    from autoware_vision_pilot_amd import lib

    with pytest.raises(lib.VpError):
        lib.Engine("sceneseg", b"not a blob at all")
    with pytest.raises((lib.VpError, ValueError)):
        lib.Engine("sceneseg", "/nonexistent/weights.vpw")
    eng = engines("sceneseg", "fp16")
    with pytest.raises(ValueError):
        eng.infer(np.zeros((10, 10), dtype=np.uint8))


def test_no_state_leaks_between_frames(engines, frame720):
    """Per-frame accumulators (SE pool sums, split-K slabs) must be re-initialised by every replay: A, B, A -> A."""
    from oracle import pre_post

    other = pre_post.synthetic_frame(720, 1280, 77)
    for kind, prec in (("sceneseg", "fp16x3"), ("egolanes", "fp16")):
        eng = engines(kind, prec)
        eng.infer(frame720)
        a1 = eng.logits()
        eng.infer(other)
        b = eng.logits()
        for _ in range(3):
            eng.infer(frame720)
        a2 = eng.logits()
        assert np.array_equal(a1, a2), f"{kind}/{prec}: state leaked between frames"
        assert not np.array_equal(a1, b)


def test_visualize_mask_bit_exact(engines, frame720):
    """colour LUT + nearest resize + 50/50 blend (masks_visualization_engine.cpp:11-58), bit-exact."""
    from autoware_vision_pilot_amd import lib
    from oracle import pre_post

    eng = engines("sceneseg", "fp16x3")
    eng.set_decode_mode(lib.VP_DECODE_SEG_MASK)
    eng.infer(frame720)
    got = eng.visualize_mask(0, frame720.shape[:2])
    want = pre_post.visualize_mask(eng.mask(), frame720, 0)
    assert got.shape == frame720.shape and np.array_equal(got, want)
    assert (got != frame720 // 2 + 0).any()  # some pixels carry the red overlay
    ego = engines("egolanes", "fp16x3")
    ego.set_decode_mode(lib.VP_DECODE_LANE_LABEL)
    ego.infer(frame720)
    assert np.array_equal(ego.visualize_mask(2, frame720.shape[:2]), pre_post.visualize_mask(ego.mask(), frame720, 2))
    dom = engines("domainseg", "fp16x3")
    dom.infer(frame720)
    assert np.array_equal(dom.visualize_mask(1, frame720.shape[:2]), pre_post.visualize_mask(dom.mask(), frame720, 1))


def test_config1_640x360_rgb_class_map(engines, state_dicts):
    """BASELINE configs[0] / SURVEY.md 8(d) row 1: SceneSeg on a 640x360 RGB frame (rng(0) integers), the reference's own
    CPU-runnable case: logits within 1e-3 and the int64 class map equal to the oracle's except inside the tolerance band."""
    from autoware_vision_pilot_amd import lib
    from oracle import nets, pre_post

    frame = np.random.default_rng(0).integers(0, 256, size=(360, 640, 3), dtype=np.uint8)
    x = torch.from_numpy(pre_post.preprocess(frame, input_is_bgr=False, planes_rgb=True))
    ref = nets.forward("sceneseg", nets.to_torch(state_dicts("sceneseg")), x)[0].numpy()
    eng = engines("sceneseg", "fp16x3")
    eng.set_input_format(lib.VP_RGB8, lib.VP_PLANES_RGB)
    eng.set_decode_mode(lib.VP_DECODE_CLASS_INDEX)
    try:
        eng.infer(frame)
        assert np.array_equal(eng.input_tensor(), x.numpy())
        got = eng.logits()[0] if eng.logits().ndim == 4 else eng.logits()
        assert _rel(got, ref) <= 1e-3
        cls, ref_cls = eng.mask().astype(np.int64), pre_post.argmax_classes(ref)
        srt = np.sort(ref, axis=0)
        flips = cls != ref_cls
        assert flips.mean() <= 5e-4 and ((srt[-1] - srt[-2])[flips] < 2e-3).all()
    finally:
        eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_BGR)
        eng.set_decode_mode(lib.VP_DECODE_SEG_MASK)


def test_egolanes_autosteer_handover_ring(engines, frame720):
    """vp_set_lane_ring (production_release main.cpp:472-534): the raw EgoLanes logits of the last two frames, [t-1 | t] = fp32
    {1, 6, 80, 160}, kept on the device behind every pass -- eager first pass, graph capture, graph replays -- bit for bit the engine's own logits
    of those frames; one valid frame after the first pass (the reference skips AutoSteer then), two from the second on; switching the ring off and on
    again starts over; other model kinds refuse."""
    from autoware_vision_pilot_amd import lib
    from oracle import pre_post

    eng = engines("egolanes", "fp16x3")
    eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_RGB)
    eng.set_lane_ring(True)
    try:
        frames = [frame720] + [pre_post.synthetic_frame(360, 640, 40 + i) for i in range(4)]
        prev = None
        for i, f in enumerate(frames):
            eng.infer(f)
            cur = eng.logits().copy()
            ring, nvalid = eng.lane_ring()
            assert nvalid == min(i + 1, 2)
            assert np.array_equal(ring[3:], cur), i
            assert np.array_equal(ring[:3], prev) if prev is not None else not ring[:3].any()
            prev = cur
        eng.set_lane_ring(False)
        with pytest.raises(lib.VpError):
            eng.lane_ring()
        eng.set_lane_ring(True)
        eng.infer(frames[1])
        ring, nvalid = eng.lane_ring()
        assert nvalid == 1 and not ring[:3].any() and np.array_equal(ring[3:], eng.logits())
        with pytest.raises(ValueError):
            engines("sceneseg", "fp16x3").set_lane_ring(True)
    finally:
        eng.set_lane_ring(False)
        eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_BGR)
