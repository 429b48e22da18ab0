"""CPU suite, part 1: the oracle against the committed golden vectors.

tests/golden/*.npz were produced by oracle/pin_against_reference.py, which executes the REFERENCE's own nn.Modules
(context / neck / heads / feature fusion imported from /root/reference/Models/model_components) on the seeded
state-dicts.  /root/reference does not exist on the GPU box, so here we only read the fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import nets, pre_post, weights

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SEEDS = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
KINDS = list(SEEDS)


def _close(a, b, tol=1e-4):
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()) <= tol


def test_param_counts():
    # SceneSeg 48.36 M (4.008 M backbone + 44.355 M decoder), EgoLanes 51.9 M
    assert weights.param_count("sceneseg") == 48362503
    assert abs(weights.param_count("egolanes") - 51.9e6) < 0.1e6
    bb = sum(int(np.prod(s)) for _, s, k in weights.backbone_spec("x.") if k not in ("bn_mean", "bn_var"))
    assert abs(bb - 4.008e6) < 0.01e6


@pytest.mark.parametrize("kind", KINDS)
def test_decoder_small_against_reference_modules(kind):
    g = np.load(os.path.join(GOLDEN, f"decoder_small_{kind}.npz"))
    seed = int(g["weight_seed"])
    sd = nets.to_torch(weights.make_state_dict(kind, seed))
    rng = np.random.default_rng(10_000 + seed)
    c = weights.context_channels(kind)
    shapes = [(32, 32, 48), (24, 16, 24), (40, 8, 12), (80, 4, 6), (c, 2, 3)]
    fs = [torch.from_numpy(rng.standard_normal((1,) + s, dtype=np.float32)) for s in shapes]
    p = weights.PREFIX[kind]
    with torch.no_grad():
        nk = nets.neck(sd, p["neck"], fs[4], fs)
        out = nets.head_egolanes(sd, p["head"], nk) if kind == "egolanes" else nets.head_full_res(sd, p["head"], nk, fs)
    assert _close(nk[0, ::8].numpy(), g["neck_ds"])
    assert _close(out[0].numpy(), g["out"])


@pytest.mark.parametrize("kind", KINDS)
def test_context_against_reference_modules(kind):
    g = np.load(os.path.join(GOLDEN, f"context_{kind}.npz"))
    seed = int(g["weight_seed"])
    sd = nets.to_torch(weights.make_state_dict(kind, seed))
    rng = np.random.default_rng(20_000 + seed)
    f = torch.from_numpy(np.abs(rng.standard_normal((1, weights.context_channels(kind), 10, 20), dtype=np.float32)))
    with torch.no_grad():
        got = nets.context(sd, weights.PREFIX[kind]["context"], f)[0].numpy()
    assert _close(got.ravel()[g["samples_idx"]], g["samples"])


@pytest.mark.parametrize("kind", KINDS)
def test_full_network_against_golden(kind):
    g = np.load(os.path.join(GOLDEN, f"full_{kind}.npz"))
    sd = nets.to_torch(weights.make_state_dict(kind, int(g["weight_seed"])))
    frame = pre_post.synthetic_frame(720, 1280, int(g["frame_seed"]))
    x = torch.from_numpy(pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=False))
    out = nets.forward(kind, sd, x)[0].numpy()
    assert tuple(out.shape) == tuple(g["shape"])
    assert _close(out.ravel()[g["samples_idx"]], g["samples"], tol=2e-4)
    if kind == "sceneseg":
        cls = pre_post.argmax_classes(out)
        assert np.array_equal(np.bincount(cls.ravel(), minlength=3), g["hist"]) or (cls != g["classes"]).sum() <= int(g["margin_lt_1e-3"])
        srt = np.sort(out, axis=0)
        flips = cls != g["classes"]
        assert ((srt[-1] - srt[-2])[flips] < 1e-3).all()
        assert np.array_equal(cls, torch.max(torch.from_numpy(out).permute(1, 2, 0), dim=2)[1].numpy())  # scene_seg_infer.py:54
    elif kind == "egolanes":
        assert (pre_post.egolanes_priority_mask(out) != g["mask"]).mean() < 1e-3
    elif kind == "domainseg":
        assert (np.packbits(out[0] > 0) != g["mask_packed"]).mean() < 1e-3
    else:
        assert _close(out[0, ::8, ::8], g["depth_ds8"], tol=2e-4)


def test_preprocess_fixture_and_properties():
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    assert np.array_equal(pre_post.resize_bilinear_u8(g["frame"], 40, 80), g["resized"])
    assert np.array_equal(pre_post.resize_bilinear_u8(g["frame"][:20, :30], 47, 61), g["up"])
    # identity at scale 1, constant images stay constant, plane-order bookkeeping
    img = pre_post.synthetic_frame(320, 640, 3, smooth=False)
    assert np.array_equal(pre_post.resize_bilinear_u8(img), img)
    const = np.full((100, 200, 3), 77, dtype=np.uint8)
    assert (pre_post.resize_bilinear_u8(const) == 77).all()
    a = pre_post.preprocess(img, input_is_bgr=True, planes_rgb=False)
    b = pre_post.preprocess(img[..., ::-1], input_is_bgr=False, planes_rgb=True)
    assert np.array_equal(a[0, ::-1], b[0])
    # torchvision ToTensor+Normalize semantics (scene_seg_infer.py:15-20) on an RGB image
    t = torch.from_numpy(img).permute(2, 0, 1).float().div(255)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    assert np.array_equal(pre_post.preprocess(img, input_is_bgr=False, planes_rgb=True)[0], t.sub(mean).div(std).numpy())


def test_decode_definitions():
    rng = np.random.default_rng(0)
    lg = rng.standard_normal((3, 17, 23), dtype=np.float32)
    lg[:, 0, 0] = 1.5  # exact three-way tie -> class 0
    lg[1:, 0, 1] = 9.0  # tie between 1 and 2 -> class 1
    cls = pre_post.argmax_classes(lg)
    assert cls[0, 0] == 0 and cls[0, 1] == 1
    assert np.array_equal(cls, torch.max(torch.from_numpy(lg).permute(1, 2, 0), dim=2)[1].numpy())
    m = pre_post.seg_mask_u8(lg)
    assert set(np.unique(m)) <= {0, 255} and np.array_equal(m == 255, cls == 1)
    one = lg[:1].copy()
    one[0, 0, 0] = 0.0  # run_model_node.cpp:168 '> 0' is strict
    assert pre_post.seg_mask_u8(one)[0, 0] == 0
    lab = pre_post.egolanes_priority_mask(lg)
    ref = np.full(lg.shape[1:], 255, np.uint8)
    ref[lg[0] > 0] = 0
    ref[lg[1] > 0] = 1
    ref[lg[2] > 0] = 2
    assert np.array_equal(lab, ref)
    assert np.array_equal(pre_post.egolanes_planes(lg, 0.0), (lg > 0).astype(np.float32))
    # nearest / bilinear resize: identity at scale 1, integer up-scale replicates
    assert np.array_equal(pre_post.resize_nearest_u8(m, 17, 23), m)
    assert np.array_equal(pre_post.resize_nearest_u8(m, 34, 46), np.repeat(np.repeat(m, 2, 0), 2, 1))
    assert np.array_equal(pre_post.resize_bilinear_f32(lg[0], 17, 23), lg[0])


def test_backbone_against_independent_implementation_fixture():
    """a4 pin: the EfficientNet-B0 restatement vs values sampled from HuggingFace transformers' EfficientNet (an independent
    implementation of the published network) carrying the SAME seeded weights, stride-2 pads made symmetric as torchvision's
    (oracle/pin_backbone_hf.py wrote the fixture; measured difference 0.0 on all 16 MBConv blocks, stem and top conv)."""
    from oracle import pin_backbone_hf as pin

    g = np.load(pin.GOLDEN)
    h, w = (int(v) for v in g["hw"])
    prefix = weights.PREFIX["sceneseg"]["backbone"]
    sd_t = nets.to_torch(weights.make_state_dict("sceneseg", int(g["seed"])))
    image = torch.from_numpy(np.random.default_rng(int(g["image_seed"])).standard_normal((1, 3, h, w)).astype(np.float32))
    outs = pin.oracle_blocks(sd_t, prefix, image)
    assert len(outs) == 18
    for i, y in enumerate(outs):
        got = y.reshape(-1)[torch.from_numpy(g[f"idx{i}"])].numpy()
        want = g[f"val{i}"]
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, float(np.abs(want).max())), i


def test_backbone_against_transformers_live():
    """The same comparison executed live where transformers is importable (this image): every block, bit for bit."""
    pytest.importorskip("transformers")
    from oracle import pin_backbone_hf as pin

    prefix = weights.PREFIX["sceneseg"]["backbone"]
    sd = weights.make_state_dict("sceneseg", 7)
    m = pin.build_hf()
    pin.load_oracle_weights(m, sd, prefix)
    image = torch.from_numpy(np.random.default_rng(5).standard_normal((1, 3, 64, 96)).astype(np.float32))
    a, b = pin.oracle_blocks(nets.to_torch(sd), prefix, image), pin.hf_blocks(m, image)
    for i, (x, y) in enumerate(zip(a, b)):
        assert float(((x - y).abs() / y.abs().clamp(min=1.0)).max()) <= 1e-5, i


def test_exact_2x_downscale_is_the_box_average():
    """cv::resize(INTER_LINEAR) switches to INTER_AREA when both scale factors are exactly 2 (e.g. a 1280x640 frame for the
    640x320 network input; imgproc resize.cpp).  With half-pixel centres an exact 2x bilinear samples midway between two
    source pixels per axis, i.e. it IS the 2x2 box average, and the integer bilinear's rounding ((sum + 2) >> 2) equals
    INTER_AREA's rounding of that average: no special case is needed -- checked here against a direct 2x2 mean."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(640, 1280, 3), dtype=np.uint8)
    got = pre_post.resize_bilinear_u8(img, 320, 640)
    s = img.astype(np.int32)
    box = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(got, box.astype(np.uint8))


def test_pil_resample_is_pillows():
    """oracle/pre_post.py resize_pil_u8 (the AutoDrive frame path, Models/visualizations/AutoDrive/video_visualization.py:29-33, and
    the scene visualisations' default-filter Image.resize) against PIL ITSELF, bit for bit: the headline 1920x1080 -> 1024x512,
    odd sizes, up-scaling along one or both axes, a pass that is skipped, both filters -- and the fp32 planes against PIL +
    to_tensor + normalize spelled out in torch."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(12)
    cases = [((1080, 1920), (512, 1024)), ((487, 651), (512, 1024)), ((720, 1280), (320, 640)), ((33, 47), (64, 90)), ((512, 1024), (512, 1024)),
             ((320, 700), (320, 640)), ((1081, 641), (320, 640)), ((2, 3), (320, 640))]
    for (h, w), (oh, ow) in cases:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for which, pilf in ((pre_post.PIL_BILINEAR, Image.BILINEAR), (pre_post.PIL_BICUBIC, Image.BICUBIC)):
            ref = np.asarray(Image.fromarray(img).resize((ow, oh), pilf))
            assert np.array_equal(pre_post.resize_pil_u8(img, oh, ow, which), ref), ((h, w), (oh, ow), which)
    frame = pre_post.synthetic_frame(1080, 1920, 20)           # BGR
    pil = Image.fromarray(np.ascontiguousarray(frame[..., ::-1])).resize((1024, 512), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    t = t.sub(torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)).div(torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1))
    mine = pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=True, out_h=512, out_w=1024, resize="pil_bilinear")
    assert np.array_equal(mine[0], t.numpy())


def test_metric_configuration_against_golden():
    """BASELINE.json's metric configuration -- SceneSeg + Scene3D on one frame, Scene3D on SceneSeg's encoder (scene_3d_network.py:9-13)
    -- against the fixture the REFERENCE's DepthContext / Scene3DNeck / Scene3DHead modules produced on SceneSeg's taps
    (oracle/pin_against_reference.py section 5).  SceneSeg's half is full_sceneseg.npz (test_full_network_against_golden)."""
    from autoware_vision_pilot_amd import synthetic

    g = np.load(os.path.join(GOLDEN, "metric_scene3d_on_sceneseg.npz"))
    seed_seg, seed_3d = (int(v) for v in g["weight_seeds"])
    sd3 = synthetic.share_backbone(weights.make_state_dict("scene3d", seed_3d), "scene3d", weights.make_state_dict("sceneseg", seed_seg), "sceneseg")
    frame = pre_post.synthetic_frame(720, 1280, int(g["frame_seed"]))
    x = torch.from_numpy(pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=False))
    out = nets.forward("scene3d", nets.to_torch(sd3), x)[0].numpy()
    assert tuple(out.shape) == tuple(g["shape"])
    assert _close(out.ravel()[g["samples_idx"]], g["samples"], tol=2e-4)
    assert _close(out[0, ::8, ::8], g["depth_ds8"], tol=2e-4)


def test_autospeed_letterbox_geometry_known_answers():
    """oracle/autospeed.py letterbox_geometry against values worked out from the reference's expressions (onnxruntime_engine.cpp:76-92: scale =
    min(W/w, H/h) in fp32, new size truncated, pads by integer division) for six frame shapes incl. portrait, square, extreme aspect and odd sizes.
    (The detector stages are PARITY UNPINNED -- oracle/autospeed.py header: the reference's C++ needs ONNX Runtime + OpenCV, absent here.)"""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _autospeed_cases as cases
    from oracle import autospeed

    for (h, w), want in cases.LETTERBOX_GEOMETRY:
        scale, new_w, new_h, px, py = autospeed.letterbox_geometry(h, w)
        assert (float(scale), px, py) == (float(np.float32(want[0])), want[1], want[2]), (h, w)
        assert new_w == int(np.float32(w) * scale) and new_h == int(np.float32(h) * scale)


def test_normalisation_forms_against_c_float_arithmetic(tmp_path):
    """oracle/pre_post.py normalize_planes in both spellings against the C++ front-ends' arithmetic compiled by gcc: the reference's
    expressions are float multiplications / subtractions / divisions in C (cv::Mat::convertTo(CV_32FC3, 1.0 / 255.0) = `q * (float)(1.0 /
    255.0)`, onnx_runtime_backend.cpp:45; `(x - MEAN[c]) / STD[c]`, onnxruntime_engine.cpp:98), so a ten-line C program evaluating them
    with contraction off IS that arithmetic; torchvision's to_tensor is `q / 255.0f`.  All 256 x 3 values, bit for bit."""
    import shutil
    import subprocess

    from oracle import pre_post

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not found")
    src = tmp_path / "norm.c"
    src.write_text(r"""
#include <stdio.h>
int main(void) {
  static const float MEAN[3] = {0.485f, 0.456f, 0.406f}, STD[3] = {0.229f, 0.224f, 0.225f};
  const float inv = (float)(1.0 / 255.0);
  for (int c = 0; c < 3; ++c)
    for (int q = 0; q < 256; ++q) {
      volatile float a = (float)q * inv;   /* convertTo(CV_32FC3, 1.0 / 255.0) */
      volatile float b = (float)q / 255.0f; /* to_tensor */
      volatile float ya = (a - MEAN[c]) / STD[c], yb = (b - MEAN[c]) / STD[c];
      printf("%a %a\n", ya, yb);
    }
  return 0;
}
""")
    exe = tmp_path / "norm"
    subprocess.check_call([gcc, "-O0", "-ffp-contract=off", "-o", str(exe), str(src)])
    vals = np.array([[float.fromhex(t) for t in line.split()] for line in subprocess.check_output([str(exe)], text=True).splitlines()], dtype=np.float32)
    img = np.zeros((16, 16, 3), np.uint8)
    img[...] = np.arange(256, dtype=np.uint8).reshape(16, 16, 1)
    for col, form in ((0, "opencv"), (1, "torchvision")):
        got = pre_post.normalize_planes(img, input_is_bgr=False, planes_rgb=True, norm_form=form).reshape(3, 256)
        assert np.array_equal(got, vals[:, col].reshape(3, 256)), form
    assert int((vals[:, 0] != vals[:, 1]).sum()) == 322


# ---- third-party boundaries: "pinned on first contact" (round 5).  oracle/pin_opencv.py / pin_torchvision.py run wherever cv2 / torchvision
# import and write a fixture; from then on these tests hold the restatement to it on every box.  Here neither imports: the scripts say so and
# exit 0, the fixture tests skip -- nothing is claimed (parity unpinned at these boundaries).
def test_pin_scripts_skip_cleanly_or_pass():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mod in ("oracle.pin_opencv", "oracle.pin_torchvision"):
        r = subprocess.run([sys.executable, "-m", mod], cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (mod, r.stdout[-500:], r.stderr[-500:])
        assert mod.split(".")[1] in r.stdout


def test_opencv_pin_fixture():
    path = os.path.join(os.path.dirname(__file__), "golden", "opencv_pin.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/opencv_pin.npz not generated yet (oracle/pin_opencv.py --write needs cv2, absent from this image)")
    from oracle import pin_opencv, pre_post

    g = np.load(path)
    for h, w, seed in pin_opencv.CASES:
        frame = pre_post.synthetic_frame(h, w, seed, smooth=(seed % 2 == 1))
        small = h * w <= 651 * 487
        got = pre_post.resize_bilinear_u8(frame)
        assert np.array_equal(got if small else got[::7, ::11], g[f"u8_{h}x{w}"])
        rng = np.random.default_rng(seed)
        mask = (rng.integers(0, 2, size=(pre_post.NET_H, pre_post.NET_W), dtype=np.uint8) * 255).astype(np.uint8)
        got_m = pre_post.resize_nearest_u8(mask, h, w)
        assert np.array_equal(got_m if small else got_m[::13, ::17], g[f"nearest_{h}x{w}"])
        depth = rng.standard_normal((pre_post.NET_H, pre_post.NET_W)).astype(np.float32) * np.float32(7.0)
        got_d = pre_post.resize_bilinear_f32(depth, h, w)
        want_d = g[f"f32_{h}x{w}"]
        got_d = got_d if small else got_d[::13, ::17]
        assert (np.abs(got_d - want_d) <= np.spacing(np.abs(want_d)) + 1e-30).all()


def test_torchvision_pin_fixture():
    path = os.path.join(os.path.dirname(__file__), "golden", "torchvision_pin.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/torchvision_pin.npz not generated yet (oracle/pin_torchvision.py --write needs torchvision, absent from this image)")
    import torch

    from oracle import nets, pin_torchvision
    from oracle.weights import PREFIX, make_state_dict

    g = np.load(path)
    sd = make_state_dict("sceneseg", pin_torchvision.SEED)
    with torch.no_grad():
        taps = nets.backbone(nets.to_torch(sd), PREFIX["sceneseg"]["backbone"], torch.from_numpy(g["x"]))
    for i, t in enumerate(taps):
        a, b = t.numpy()[0, ::3, ::5, ::7], g[f"tap{i}"]
        assert float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()) <= 1e-5
