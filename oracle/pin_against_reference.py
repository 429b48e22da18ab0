#!/usr/bin/env python3
"""Pin the oracle against the reference's OWN nn.Modules and (re)generate tests/golden/*.npz.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Runs only where /root/reference exists (the build
container); the GPU box uses the committed fixtures.  Usage:  python oracle/pin_against_reference.py

What is executed from the reference (imported, never copied): Models/model_components/
{scene_context,scene_neck,scene_seg_head,depth_context,scene_3d_neck,scene_3d_head,domain_seg_head,
auto_steer_context,ego_path_neck,ego_lanes_head,backbone_feature_fusion}.py.  backbone.py needs torchvision
(absent) -> the EfficientNet-B0 stage stays "parity unpinned"; the full-network fixtures below are therefore
"oracle backbone + REFERENCE context/neck/head modules".
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/Models"

from oracle import nets, pre_post, weights  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEEDS = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
FRAME_SEED = 1
N_SAMPLES = 8192


def ref_modules(kind):
    sys.path.insert(0, REF)
    from model_components.auto_steer_context import AutoSteerContext
    from model_components.backbone_feature_fusion import BackboneFeatureFusion
    from model_components.depth_context import DepthContext
    from model_components.domain_seg_head import DomainSegHead
    from model_components.ego_lanes_head import EgoLanesHead
    from model_components.ego_path_neck import EgoPathNeck
    from model_components.scene_3d_head import Scene3DHead
    from model_components.scene_3d_neck import Scene3DNeck
    from model_components.scene_context import SceneContext
    from model_components.scene_neck import SceneNeck
    from model_components.scene_seg_head import SceneSegHead
    table = {
        "sceneseg": (SceneContext, SceneNeck, SceneSegHead),
        "scene3d": (DepthContext, Scene3DNeck, Scene3DHead),
        "domainseg": (SceneContext, SceneNeck, DomainSegHead),
        "egolanes": (AutoSteerContext, EgoPathNeck, EgoLanesHead),
    }
    c, n, h = table[kind]
    return c().eval(), n().eval(), h().eval(), BackboneFeatureFusion().eval()


def load_part(module, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    module.load_state_dict(sub, strict=True)
    return module


def sample_indices(n, total, seed=1234):
    return np.sort(np.random.default_rng(seed).choice(total, size=min(n, total), replace=False))


def small_decoder_inputs(kind, seed):
    """Seeded small-spatial feature pyramid for the fully-convolutional neck+head (context grid 2x3)."""
    rng = np.random.default_rng(10_000 + seed)
    c = weights.context_channels(kind)
    shapes = [(32, 32, 48), (24, 16, 24), (40, 8, 12), (80, 4, 6), (c, 2, 3)]
    return [torch.from_numpy(rng.standard_normal((1,) + s, dtype=np.float32)) for s in shapes]


@torch.no_grad()
def main():
    torch.manual_seed(0)
    os.makedirs(GOLDEN, exist_ok=True)
    frame = pre_post.synthetic_frame(720, 1280, FRAME_SEED)
    x = torch.from_numpy(pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=False))
    report = []
    for kind, seed in SEEDS.items():
        sd = nets.to_torch(weights.make_state_dict(kind, seed))
        p = weights.PREFIX[kind]
        ctx_m, neck_m, head_m, fuse_m = ref_modules(kind)
        load_part(ctx_m, sd, p["context"])
        load_part(neck_m, sd, p["neck"])
        load_part(head_m, sd, p["head"])

        # ---- (1) full network: oracle backbone + REFERENCE modules, vs all-oracle
        feats = nets.backbone(sd, p["backbone"], x)
        deep = fuse_m(feats) if kind == "egolanes" else feats[4]
        ref_ctx = ctx_m(deep)
        ref_neck = neck_m(ref_ctx, feats)
        ref_out = head_m(ref_neck) if kind == "egolanes" else head_m(ref_neck, feats)
        ora_out, inter = nets.forward(kind, sd, x, return_intermediates=True)
        errs = dict(ctx=(inter["ctx"] - ref_ctx).abs().max().item(), neck=(inter["neck"] - ref_neck).abs().max().item(),
                    out=(ora_out - ref_out).abs().max().item())
        if kind == "egolanes":
            errs["fusion"] = (inter["deep"] - deep).abs().max().item()
        out = ref_out[0].numpy()
        idx = sample_indices(N_SAMPLES, out.size)
        full = dict(samples_idx=idx.astype(np.int64), samples=out.ravel()[idx].astype(np.float32),
                    shape=np.array(out.shape, dtype=np.int64), frame_seed=np.int64(FRAME_SEED), weight_seed=np.int64(seed),
                    mean=np.float64(out.mean()), std=np.float64(out.std()))
        if kind == "sceneseg":
            cls = pre_post.argmax_classes(out)
            assert np.array_equal(cls, torch.max(ref_out[0].permute(1, 2, 0), dim=2)[1].numpy())  # scene_seg_infer.py:54
            srt = np.sort(out, axis=0)
            full["classes"] = cls.astype(np.uint8)
            full["margin_lt_1e-3"] = np.int64(((srt[-1] - srt[-2]) < 1e-3).sum())
            full["hist"] = np.bincount(cls.ravel(), minlength=3).astype(np.int64)
        elif kind == "egolanes":
            full["mask"] = pre_post.egolanes_priority_mask(out)
        elif kind == "domainseg":
            full["mask_packed"] = np.packbits(out[0] > 0)
        else:
            full["depth_ds8"] = out[0, ::8, ::8].astype(np.float32)
        np.savez_compressed(os.path.join(GOLDEN, f"full_{kind}.npz"), **full)

        # ---- (2) small-spatial neck+head through the REFERENCE modules (tiny fixture, all outputs kept)
        fs = small_decoder_inputs(kind, seed)
        ref_nk = neck_m(fs[4], fs)
        ref_o = head_m(ref_nk) if kind == "egolanes" else head_m(ref_nk, fs)
        ora_nk = nets.neck(sd, p["neck"], fs[4], fs)
        ora_o = nets.head_egolanes(sd, p["head"], ora_nk) if kind == "egolanes" else nets.head_full_res(sd, p["head"], ora_nk, fs)
        errs["small_out"] = (ora_o - ref_o).abs().max().item()
        np.savez_compressed(os.path.join(GOLDEN, f"decoder_small_{kind}.npz"), out=ref_o[0].numpy().astype(np.float32),
                            neck_ds=ref_nk[0, ::8].numpy().astype(np.float32), weight_seed=np.int64(seed))

        # ---- (3) context block alone on a seeded deep feature map
        rng = np.random.default_rng(20_000 + seed)
        f = torch.from_numpy(np.abs(rng.standard_normal((1, weights.context_channels(kind), 10, 20), dtype=np.float32)))
        ref_c = ctx_m(f)[0].numpy()
        errs["ctx_alone"] = float(np.abs(nets.context(sd, p["context"], f)[0].numpy() - ref_c).max())
        ci = sample_indices(4096, ref_c.size, seed=99)
        np.savez_compressed(os.path.join(GOLDEN, f"context_{kind}.npz"), samples_idx=ci.astype(np.int64),
                            samples=ref_c.ravel()[ci].astype(np.float32), weight_seed=np.int64(seed))
        report.append((kind, errs, float(out.std())))
        print(kind, {k: f"{v:.3e}" for k, v in errs.items()}, f"logit std {out.std():.3f} mean {out.mean():.3f}", flush=True)
        assert max(errs.values()) <= 1e-4 * max(1.0, float(np.abs(out).max())), errs

    # ---- (5) BASELINE.json's METRIC CONFIGURATION: SceneSeg + Scene3D on ONE camera frame, Scene3D built on the SceneSeg backbone
    # (scene_3d_network.py:9-13: Scene3DNetwork(pretrained SceneSeg) keeps SceneSeg's encoder) -- the pair bench.py times and
    # vp_create + vp_create_shared run.  SceneSeg's half is full_sceneseg.npz (same seed, same frame); this is Scene3D's half:
    # the REFERENCE DepthContext / Scene3DNeck / Scene3DHead modules on the taps of SceneSeg's encoder.
    from autoware_vision_pilot_amd import synthetic
    sd_seg = weights.make_state_dict("sceneseg", SEEDS["sceneseg"])
    sd3 = nets.to_torch(synthetic.share_backbone(weights.make_state_dict("scene3d", SEEDS["scene3d"]), "scene3d", sd_seg, "sceneseg"))
    p3 = weights.PREFIX["scene3d"]
    ctx_m, neck_m, head_m, _ = ref_modules("scene3d")
    load_part(ctx_m, sd3, p3["context"])
    load_part(neck_m, sd3, p3["neck"])
    load_part(head_m, sd3, p3["head"])
    feats_seg = nets.backbone(nets.to_torch(sd_seg), weights.PREFIX["sceneseg"]["backbone"], x)
    feats3 = nets.backbone(sd3, p3["backbone"], x)
    assert all(torch.equal(a, b) for a, b in zip(feats_seg, feats3)), "shared backbone: the taps must be SceneSeg's"
    ref3 = head_m(neck_m(ctx_m(feats3[4]), feats3), feats3)
    ora3 = nets.forward("scene3d", sd3, x)
    e3 = (ora3 - ref3).abs().max().item()
    print("metric configuration (Scene3D on the SceneSeg encoder): oracle vs reference modules", f"{e3:.3e}")
    assert e3 <= 1e-4 * max(1.0, float(ref3.abs().max()))
    d3 = ref3[0].numpy()
    idx3 = sample_indices(N_SAMPLES, d3.size, seed=4321)
    np.savez_compressed(os.path.join(GOLDEN, "metric_scene3d_on_sceneseg.npz"), samples_idx=idx3.astype(np.int64),
                        samples=d3.ravel()[idx3].astype(np.float32), depth_ds8=d3[0, ::8, ::8].astype(np.float32),
                        shape=np.array(d3.shape, dtype=np.int64), frame_seed=np.int64(FRAME_SEED),
                        weight_seeds=np.array([SEEDS["sceneseg"], SEEDS["scene3d"]], dtype=np.int64), mean=np.float64(d3.mean()), std=np.float64(d3.std()))

    # ---- (4) decode + preprocess fixtures (pure numpy definitions; pinned to themselves + torch.max above)
    small = pre_post.synthetic_frame(90, 160, 7, smooth=False)
    np.savez_compressed(os.path.join(GOLDEN, "preprocess.npz"), frame=small,
                        resized=pre_post.resize_bilinear_u8(small, 40, 80),
                        up=pre_post.resize_bilinear_u8(small[:20, :30], 47, 61))
    print("golden fixtures written to", GOLDEN)
    return report


if __name__ == "__main__":
    main()
