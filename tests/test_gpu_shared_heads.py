"""BASELINE configs[2]: several heads on one camera frame with the shared sub-networks run once (vp_create_shared).

The reference builds Scene3D / DomainSeg on a pre-trained SceneSeg (scene_3d_network.py:9-13, domain_seg_network.py:9-12):
the state-dicts below graft SceneSeg's backbone (DomainSeg: backbone + context + neck) into the other network exactly as
that object sharing does (oracle.weights.share_backbone), the oracle runs each network on its own, and the shared-prefix
engines must reproduce it."""
import numpy as np
import pytest
import torch

from oracle import nets, pre_post, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shared_setup(state_dicts, frame720):
    from autoware_vision_pilot_amd import lib, weights as vw

    sd_seg = state_dicts("sceneseg")
    sd_3d = weights.share_backbone(dict(state_dicts("scene3d")), "scene3d", sd_seg, "sceneseg")
    sd_dom = weights.share_backbone(dict(state_dicts("domainseg")), "domainseg", sd_seg, "sceneseg", also_context_neck=True)
    sd_ego = weights.share_backbone(dict(state_dicts("egolanes")), "egolanes", sd_seg, "sceneseg")
    x = torch.from_numpy(pre_post.preprocess(frame720, input_is_bgr=True, planes_rgb=False))
    sds = {"scene3d": sd_3d, "domainseg": sd_dom, "egolanes": sd_ego}
    ref = {k: nets.forward(k, nets.to_torch(sd), x)[0].numpy() for k, sd in sds.items()}
    ref["sceneseg"] = nets.forward("sceneseg", nets.to_torch(sd_seg), x)[0].numpy()
    out = {}
    for prec in ("fp16x3", "fp16"):
        base = lib.Engine("sceneseg", vw.pack_state_dict(sd_seg), precision=prec)
        heads = {k: lib.Engine(k, vw.pack_state_dict(sd), precision=prec, base=base) for k, sd in sds.items()}
        out[prec] = (base, heads)
    yield ref, out
    for base, heads in out.values():
        for h in heads.values():
            h.close()
        base.close()


def test_shared_levels(shared_setup):
    _, out = shared_setup
    base, heads = out["fp16x3"]
    assert base.shared_level() == 0
    assert heads["scene3d"].shared_level() == 1      # backbone only (own DepthContext / DepthNeck)
    assert heads["domainseg"].shared_level() == 2    # backbone + context + neck
    assert heads["egolanes"].shared_level() == 1     # backbone taps feed its own feature fusion
    n_base = len(base.layers())
    assert len(heads["domainseg"].layers()) < 12 < len(heads["scene3d"].layers()) < n_base


@pytest.mark.parametrize("prec,tol_rel", [("fp16x3", 1e-3), ("fp16", 3e-2)])  # same bars as test_gpu_networks.py
def test_shared_heads_match_oracle(shared_setup, frame720, prec, tol_rel):
    ref, out = shared_setup
    base, heads = out[prec]
    for rep in range(2):  # second pass replays the captured graphs
        base.infer(frame720)
        got = {"sceneseg": base.logits().copy()}
        for k, h in heads.items():
            h.infer_shared()
            got[k] = h.logits().copy()
        for k, g in got.items():
            r = ref[k]
            assert g.shape == r.shape
            if prec == "fp16x3":  # the parity bar: |err| / max(1, |ref|) <= 1e-3 per element
                err = float((np.abs(g - r) / np.maximum(1.0, np.abs(r))).max())
            else:                 # throughput mode: max |err| relative to the tensor's range
                err = float(np.abs(g - r).max() / np.abs(r).max())
            assert err <= tol_rel, f"{k} {prec} pass {rep}: err {err:.3e} > {tol_rel:.0e}"


def test_shared_equals_standalone_bitwise(shared_setup, state_dicts, frame720):
    """Sharing is an execution-plan change only: same kernels, same order -> bit-identical logits."""
    from autoware_vision_pilot_amd import lib, weights as vw

    _, out = shared_setup
    base, heads = out["fp16"]
    sd = weights.share_backbone(dict(state_dicts("scene3d")), "scene3d", state_dicts("sceneseg"), "sceneseg")
    alone = lib.Engine("scene3d", vw.pack_state_dict(sd), precision="fp16")
    try:
        alone.infer(frame720)
        base.infer(frame720)
        heads["scene3d"].infer_shared()
        assert np.array_equal(alone.logits(), heads["scene3d"].logits())
    finally:
        alone.close()


def test_shared_errors_are_loud(shared_setup, state_dicts, frame720):
    from autoware_vision_pilot_amd import lib, weights as vw

    _, out = shared_setup
    base, heads = out["fp16"]
    with pytest.raises(ValueError, match="backbone parameters differ"):
        lib.Engine("scene3d", vw.pack_state_dict(state_dicts("scene3d")), precision="fp16", base=base)
    with pytest.raises(ValueError, match="precision"):
        lib.Engine("scene3d", vw.pack_state_dict(state_dicts("scene3d")), precision="fp16x3", base=base)
    with pytest.raises(ValueError, match="base"):
        heads["scene3d"].infer(frame720)
    with pytest.raises(ValueError, match="not a shared"):
        base.infer_shared()


@pytest.mark.parametrize("prec", ["fp16x3", "fp16"])
def test_enqueue_multi_is_the_same_frame(shared_setup, frame720, prec):
    """vp_enqueue_multi: base engine + heads as ONE graph launch, the backbone-only heads (Scene3D, EgoLanes) forked onto side
    streams behind the shared encoder, the context+neck-sharing head (DomainSeg) in order behind the base's neck.  Same kernels on
    the same data => bit-identical logits and class maps to the one-after-the-other vp_enqueue calls, frame after frame (graph
    replays), for every subset / order of heads, and after a decode-mode change (which must re-capture the combined graph)."""
    _, out = shared_setup
    base, heads = out[prec]
    order = ["scene3d", "domainseg", "egolanes"]

    def sequential(names, frame):
        base.upload_frame(frame)
        base.enqueue()
        for k in names:
            heads[k].enqueue()
        base.sync()
        return [base.logits().copy(), base.mask().copy()] + [a for k in names for a in (heads[k].logits().copy(), heads[k].mask().copy())]

    def forked(names, frame):
        base.upload_frame(frame)
        base.enqueue_multi([heads[k] for k in names])
        base.sync()
        return [base.logits().copy(), base.mask().copy()] + [a for k in names for a in (heads[k].logits().copy(), heads[k].mask().copy())]

    frame2 = np.ascontiguousarray(frame720[::-1])          # a second, different frame
    for names in (order, ["scene3d"], ["egolanes", "scene3d"], []):
        for fr in (frame720, frame2, frame720):
            want, got = sequential(names, fr), forked(names, fr)
            assert all(np.array_equal(a, b) for a, b in zip(want, got)), (prec, names)
    base.set_multi_fork(False)                              # several cameras in flight: one engine after the other, same call
    want, got = sequential(order, frame720), forked(order, frame720)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    base.set_multi_fork(True)
    base._ck(base._lib.vp_set_decode_mode(base._h, 2))      # class-index masks: both graphs are stale now
    try:
        want, got = sequential(order, frame2), forked(order, frame2)
        assert all(np.array_equal(a, b) for a, b in zip(want, got))
        assert got[1].max() <= 2                            # the class indices, not the {0, 255} mask
    finally:
        base._ck(base._lib.vp_set_decode_mode(base._h, 0))
    with pytest.raises(Exception):
        heads["scene3d"].enqueue_multi([])                  # only the encoder-owning engine can lead


def test_metric_configuration_golden_fixture(shared_setup, frame720):
    """BASELINE.json's metric configuration through the boundary call the bench times (vp_infer_multi: SceneSeg + Scene3D on one frame,
    one encoder pass) against the fixtures made by the REFERENCE's own modules: SceneSeg's class map from full_sceneseg.npz --
    identical except pixels inside the float tolerance band, none expected -- and Scene3D-on-SceneSeg's-encoder depth samples from
    metric_scene3d_on_sceneseg.npz within 1e-3."""
    import os

    from autoware_vision_pilot_amd import lib

    gdir = os.path.join(os.path.dirname(__file__), "golden")
    g_seg, g_3d = np.load(os.path.join(gdir, "full_sceneseg.npz")), np.load(os.path.join(gdir, "metric_scene3d_on_sceneseg.npz"))
    base, heads = shared_setup[1]["fp16x3"]
    base.set_decode_mode(lib.VP_DECODE_CLASS_INDEX)
    try:
        base.infer_multi([heads["scene3d"]], frame720)
        seg, depth = base.logits(), heads["scene3d"].logits()
        assert tuple(depth.shape) == tuple(g_3d["shape"])
        rel = lambda a, b: float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())
        assert rel(seg.ravel()[g_seg["samples_idx"]], g_seg["samples"]) <= 1e-3
        assert rel(depth.ravel()[g_3d["samples_idx"]], g_3d["samples"]) <= 1e-3
        assert rel(depth[0, ::8, ::8], g_3d["depth_ds8"]) <= 1e-3
        flips = base.mask() != g_seg["classes"]
        srt = np.sort(seg, axis=0)
        assert flips.sum() <= int(g_seg["margin_lt_1e-3"]) and ((srt[-1] - srt[-2])[flips] < 1e-3).all(), int(flips.sum())
    finally:
        base.set_decode_mode(lib.VP_DECODE_SEG_MASK)
