"""Parity sweep, round 5: the evidence widened to WHAT IS SHIPPED.  tests/test_gpu_parity_sweep.py sweeps the four
scene networks as stand-alone engines; this file sweeps, against the same CPU oracle and with the same rules,
  (a) the METRIC configuration -- Scene3D on SceneSeg's encoder through vp_enqueue_multi, forked and not (scene_3d_network.py:25-30) -- and
      the three-head configuration (+ EgoLanes on its own backbone and stream) over the same 8 frames;
  (b) AutoDrive: 6 frame pairs x 3 weight seeds x {fp32, fp8-stored} weights, the three scalars within 1e-3 (autodrive_network.py:32-36);
  (c) degenerate frames -- all 0, all 255, a two-colour checkerboard -- for every network: a constant frame makes every interior pixel's
      logits THE SAME vector, so a near-tie is a tie everywhere at once, and saturated inputs sit at the ends of the normalised range
      (onnx_runtime_backend.cpp:41-60);
  (d) a "trained-like" weight family (synthetic.make_trained_like_state_dict: BN scale spread over 10^3 per layer, variances down to
      eps, every second decoder tensor 3-10x smaller than in the Kaiming family) beside the three Kaiming seeds.
Rules (as in the stand-alone sweep): floats within 1e-3 x max(1, |ref|) of the fp32 oracle, or -- for a pass that misses it -- the engine
no further from an fp64 evaluation than 3x the fp32 reference is; class maps: the engine's decode bit-identical to the oracle decode of
the engine's own logits, and ZERO flips against the oracle's map outside pixels whose oracle margin is within 2x the pass's measured
logit error (each counted in the table).  The table goes to gpurun_out/parity_sweep_r5.tsv (copied to profiles/r05_parity_sweep.tsv)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FRAMES = [((720, 1280), 101, True), ((720, 1280), 102, False), ((360, 640), 103, True), ((360, 640), 104, False),
          ((1080, 1920), 105, True), ((1080, 1920), 106, False), ((720, 1280), 107, True), ((487, 651), 108, False)]
ROWS = []
K64 = 2.5   # round 6: 3.0 -> 2.5 (tests/test_gpu_parity_sweep.py has the reasoning and the by-layer record)


def _decode(kind, logits):
    from oracle import pre_post

    if kind == "sceneseg":
        srt = np.sort(logits, axis=0)
        return pre_post.argmax_classes(logits), srt[-1] - srt[-2]
    if kind == "egolanes":
        return (logits > 0).astype(np.int64), np.abs(logits)
    return (logits[0] > 0).astype(np.int64), np.abs(logits[0])


def _judge(section, kind, tag, sdt, x, ref, got, fails):
    """One pass against the oracle: appends a table row, appends to `fails` what breaks the rules."""
    from oracle import nets

    assert np.isfinite(got).all(), f"{section} {kind} {tag}: non-finite output"
    err_abs = float(np.abs(got - ref).max())
    err_rel = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
    ref_cls, margin = _decode(kind, ref)
    got_cls, _ = _decode(kind, got)
    flips = got_cls != ref_cls
    nflip = int(flips.sum())
    worst = float(margin[flips].max()) if nflip else 0.0
    note = ""
    if err_rel > 1e-3:
        sd64 = {k: v.double() for k, v in sdt.items()}
        r64 = nets.forward(kind, sd64, torch.from_numpy(x).double())[0].numpy()
        rel64 = lambda a: float((np.abs(a - r64) / np.maximum(1.0, np.abs(r64))).max())  # noqa: E731
        e_ref, e_got = rel64(ref.astype(np.float64)), rel64(got.astype(np.float64))
        note = f"fp64-judged: fp32 reference {e_ref:.3e} / engine {e_got:.3e} from fp64 = {e_got / max(e_ref, 1e-30):.2f}x (|logits| up to {float(np.abs(ref).max()):.0f})"
        if not e_got <= K64 * e_ref:
            fails.append(f"{section} {kind} {tag}: rel err {err_rel:.3e}; vs fp64: engine {e_got:.3e}, reference {e_ref:.3e}")
    if kind != "scene3d" and not (nflip == 0 or worst <= 2.0 * err_abs):
        fails.append(f"{section} {kind} {tag}: {nflip} flips, largest oracle margin {worst:.3e} vs max logit error {err_abs:.3e}")
    ROWS.append((section, kind, tag, err_abs, err_rel, int((margin < 1e-3).sum()), nflip, worst, note))
    return got_cls


@pytest.fixture(scope="module")
def metric_setup(state_dicts):
    """BASELINE.json's metric configuration: SceneSeg + Scene3D on SceneSeg's encoder (base seeds), EgoLanes beside them."""
    from autoware_vision_pilot_amd import lib, synthetic, weights as vw
    from oracle import nets

    sd_seg = state_dicts("sceneseg")
    sd_3d = synthetic.share_backbone(dict(state_dicts("scene3d")), "scene3d", sd_seg, "sceneseg")
    sd_ego = state_dicts("egolanes")
    base = lib.Engine("sceneseg", vw.pack_state_dict(sd_seg), precision="fp16x3")
    base.set_decode_mode(lib.VP_DECODE_CLASS_INDEX)
    head = lib.Engine("scene3d", vw.pack_state_dict(sd_3d), precision="fp16x3", base=base)
    ego = lib.Engine("egolanes", vw.pack_state_dict(sd_ego), precision="fp16x3")
    ego.set_input_format(lib.VP_BGR8, lib.VP_PLANES_RGB)
    yield base, head, ego, nets.to_torch(sd_seg), nets.to_torch(sd_3d)
    head.close()
    base.close()
    ego.close()


def test_metric_and_three_head_configurations(metric_setup):
    """(a): the engines bench.py's `value` and `three_heads_*` are measured on, over the sweep's 8 frames."""
    from oracle import nets, pre_post

    base, head, ego, sdt_seg, sdt_3d = metric_setup
    fails = []
    for (h, w), fseed, smooth in FRAMES:
        frame = pre_post.synthetic_frame(h, w, fseed, smooth=smooth)
        x = pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=False)
        ref_seg = nets.forward("sceneseg", sdt_seg, torch.from_numpy(x))[0].numpy()
        ref_3d = nets.forward("scene3d", sdt_3d, torch.from_numpy(x))[0].numpy()
        # EgoLanes alone (its own stream idle otherwise): the reference point of the concurrent run below
        ego.infer(frame)
        ego_alone = ego.logits().copy()
        outs = {}
        for fork in (True, False):
            base.set_multi_fork(fork)
            for rep in range(2):                       # the second pass replays the captured graph of this mode
                base.upload_frame(frame)
                ego.upload_frame(frame)
                base.enqueue_multi([head])             # SceneSeg + Scene3D: one graph launch on the base engine's stream ...
                ego.enqueue()                          # ... EgoLanes on its own stream beside it (the three-head configuration)
                base.fetch_outputs()
                head.fetch_outputs()
                ego.fetch_outputs()
            assert np.array_equal(base.input_tensor(), x)
            outs[fork] = (base.logits().copy(), head.logits().copy(), base.mask().copy(), ego.logits().copy())
        assert all(np.array_equal(a, b) for a, b in zip(outs[True], outs[False])), f"frame {fseed}: forked and unforked graphs disagree"
        assert np.array_equal(outs[True][3], ego_alone), f"frame {fseed}: EgoLanes beside the other two differs from EgoLanes alone"
        base.infer(frame)                              # the separate synchronous calls: vp_infer + vp_infer_shared
        head.infer_shared()
        assert np.array_equal(base.logits(), outs[True][0]) and np.array_equal(head.logits(), outs[True][1])
        tag = f"{h}x{w} seed {fseed} smooth {int(smooth)}"
        got_cls = _judge("metric-config(multi)", "sceneseg", tag, sdt_seg, x, ref_seg, outs[True][0], fails)
        assert np.array_equal(outs[True][2].astype(np.int64), got_cls)      # the fused decode = the oracle decode of the engine's own logits
        _judge("metric-config(multi)", "scene3d", tag, sdt_3d, x, ref_3d, outs[True][1], fails)
        ROWS.append(("three-head(concurrent)", "egolanes", tag, 0.0, 0.0, 0, 0, 0.0, "bit-identical to the stand-alone engine of the base sweep (same seed, same frame)"))
    base.set_multi_fork(True)
    assert not fails, fails


def _degenerate_frames():
    z = np.zeros((720, 1280, 3), dtype=np.uint8)
    s = np.full((720, 1280, 3), 255, dtype=np.uint8)
    yy, xx = np.mgrid[0:720, 0:1280]
    cb = np.where((((yy // 45) + (xx // 80)) % 2 == 0)[..., None], np.array([255, 0, 64], dtype=np.uint8), np.array([0, 255, 192], dtype=np.uint8)).astype(np.uint8)
    return [("all-0", z), ("all-255", s), ("checkerboard", cb)]


@pytest.mark.parametrize("kind", ["sceneseg", "scene3d", "domainseg", "egolanes"])
def test_degenerate_frames(kind, state_dicts, engines):
    """(c): constant and saturated frames through the stand-alone engines of the base seeds."""
    from autoware_vision_pilot_amd import lib
    from oracle import nets, pre_post

    eng = engines(kind, "fp16x3")
    sdt = nets.to_torch(state_dicts(kind))
    rgb = kind == "egolanes"
    eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_RGB if rgb else lib.VP_PLANES_BGR)
    if kind == "sceneseg":
        eng.set_decode_mode(lib.VP_DECODE_CLASS_INDEX)
    fails = []
    try:
        for name, frame in _degenerate_frames():
            x = pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=rgb)
            ref = nets.forward(kind, sdt, torch.from_numpy(x))[0].numpy()
            eng.infer(frame)
            assert np.array_equal(eng.input_tensor(), x)
            got = eng.logits()
            got_cls = _judge("degenerate", kind, name, sdt, x, ref, got, fails)
            if kind == "sceneseg":
                assert np.array_equal(eng.mask().astype(np.int64), got_cls)
            elif kind == "domainseg":
                assert np.array_equal(eng.mask() > 0, got_cls > 0)
    finally:
        eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_BGR)
        if kind == "sceneseg":
            eng.set_decode_mode(lib.VP_DECODE_SEG_MASK)
    assert not fails, fails


@pytest.mark.parametrize("kind", ["sceneseg", "scene3d", "domainseg", "egolanes"])
def test_trained_like_weight_family(kind):
    """(d): BN scales spread over 10^3, variances near eps, small decoder tensors."""
    from autoware_vision_pilot_amd import lib, synthetic, weights as vw
    from oracle import nets, pre_post

    seed = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}[kind]
    sd = synthetic.make_trained_like_state_dict(kind, seed)
    sdt = nets.to_torch(sd)
    rgb = kind == "egolanes"
    eng = lib.Engine(kind, vw.pack_state_dict(sd), precision="fp16x3")
    fails = []
    try:
        eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_RGB if rgb else lib.VP_PLANES_BGR)
        if kind == "sceneseg":
            eng.set_decode_mode(lib.VP_DECODE_CLASS_INDEX)
        for (h, w), fseed, smooth in (FRAMES[0], FRAMES[3], FRAMES[5]):
            frame = pre_post.synthetic_frame(h, w, fseed, smooth=smooth)
            x = pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=rgb)
            ref = nets.forward(kind, sdt, torch.from_numpy(x))[0].numpy()
            eng.infer(frame)
            got_cls = _judge("trained-like", kind, f"{h}x{w} seed {fseed} smooth {int(smooth)}", sdt, x, ref, eng.logits(), fails)
            if kind == "sceneseg":
                assert np.array_equal(eng.mask().astype(np.int64), got_cls)
    finally:
        eng.close()
    assert not fails, fails


@pytest.mark.parametrize("wseed", [5, 15, 25])
def test_autodrive_sweep(wseed):
    """(b): AutoDrive, 6 frame pairs x 3 weight seeds x {fp32, fp8-stored} weights; the streaming form on the way (pair k+1 reuses frame k's features)."""
    from autoware_vision_pilot_amd import lib, weights as vw
    from oracle import autodrive, pre_post

    sd = autodrive.make_state_dict(wseed)
    blob = vw.pack_state_dict(sd)
    frames = [pre_post.synthetic_frame(1080, 1920, 200 + i, smooth=(i % 3 != 2)) for i in range(5)]
    frames += [np.zeros((1080, 1920, 3), dtype=np.uint8), np.full((1080, 1920, 3), 255, dtype=np.uint8)]     # degenerate pair at the end
    xs = [torch.from_numpy(pre_post.preprocess(f, input_is_bgr=True, planes_rgb=True, out_h=512, out_w=1024, resize="pil_bilinear")) for f in frames]
    fails = []
    for fp8 in (False, True):
        sdt = {k: torch.from_numpy(v) for k, v in (autodrive.quantize_fp8_e4m3(sd) if fp8 else sd).items()}
        eng = lib.Engine("autodrive", blob, precision="fp16x3", weights_fp8=fp8)
        try:
            for i in range(len(frames) - 1):
                with torch.no_grad():
                    ref = np.array([float(v) for v in autodrive.forward(sdt, xs[i], xs[i + 1])], dtype=np.float32)
                tag = f"weight seed {wseed} pair {i}" + (" (all-0 -> all-255)" if i == len(frames) - 2 else "")
                try:
                    eng.infer_pair(frames[i], frames[i + 1])
                except lib.VpRangeError:
                    # the RANGE GUARD (include/vp_hip.h): the saturated frame drives this seed's multiplicative CTX gates past the fp16 exponent range of the
                    # planes; the engine says so (VP_ERR_RANGE) instead of returning garbage.  Only the degenerate pair may do that.
                    ROWS.append(("autodrive", "fp8-stored" if fp8 else "fp32-weights", tag, float("nan"), float("nan"), 0, 0, 0.0,
                                 f"RANGE GUARD raised (VP_ERR_RANGE): an activation left the fp16 range on the saturated frame; oracle (fp32) {ref.tolist()}"))
                    if i != len(frames) - 2:
                        fails.append(f"autodrive seed {wseed} fp8={fp8} pair {i}: range guard on an ordinary frame pair")
                    continue
                got = eng.logits().reshape(3)
                err = float(np.abs(got - ref).max())
                rel = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())     # the bar of every float tensor: |error| <= 1e-3 x max(1, |ref|)
                ROWS.append(("autodrive", "fp8-stored" if fp8 else "fp32-weights", tag, err, rel, 0, 0, 0.0,
                             f"(distance, curvature, flag) oracle {ref.tolist()} engine {got.tolist()}"))
                if not rel <= 1e-3:
                    fails.append(f"autodrive seed {wseed} fp8={fp8} pair {i}: got {got}, oracle {ref}")
        finally:
            eng.close()
    assert not fails, fails


def test_parity_sweep_r5_report():
    """Runs last in this file: writes the table."""
    if not ROWS:
        pytest.skip("sweep did not run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_sweep_r5.tsv"), "w") as f:
        f.write("# fp16x3 parity sweep, round 5 additions (tests/test_gpu_parity_sweep_r5.py): shipped configurations, AutoDrive, degenerate frames, trained-like weights\n")
        f.write("# section\tnetwork\tpass\tmax_abs_err\tmax_rel_err\tpixels_margin_lt_1e-3\tclass_flips\tlargest_flipped_margin\tnote\n")
        for r in ROWS:
            f.write("\t".join(str(v) if not isinstance(v, float) else f"{v:.3e}" for v in r) + "\n")
        net = [r for r in ROWS if r[0] != "autodrive" and r[0] != "three-head(concurrent)"]
        f.write(f"# {len(ROWS)} rows; network passes {len(net)}: {sum(r[6] for r in net)} class flips in total (every one at an oracle margin <= 2 x that pass's logit error), "
                f"worst rel err {max([r[4] for r in net] or [0.0]):.3e}, {sum(1 for r in net if r[8])} judged against fp64; AutoDrive passes "
                f"{sum(1 for r in ROWS if r[0] == 'autodrive')}: worst rel err {max([r[4] for r in ROWS if r[0] == 'autodrive' and r[4] == r[4]] or [0.0]):.3e}, {sum(1 for r in ROWS if r[0] == 'autodrive' and r[4] != r[4])} range-guard reports on the all-0 -> all-255 pair\n")
    print(f"parity sweep r5: {len(ROWS)} rows")
