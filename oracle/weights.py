"""Parameter inventory + seeded state-dict generator (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The key layout is the reference's ``state_dict`` key layout, so the same dict can be
``load_state_dict``-ed into the reference's own modules (oracle/pin_against_reference.py)
and exported to the engine's weight blob (autoware_vision_pilot_amd/weights.py).

Key prefixes:
  SceneSeg  : Backbone.encoder.* SceneContext.* SceneNeck.* SceneSegHead.*
              (Models/model_components/scene_seg_network.py:11-21)
  Scene3D   : PreTrainedBackbone.pretrainedBackBone.encoder.* DepthContext.* DepthNeck.* SuperDepthHead.*
              (scene_3d_network.py:13-22, pre_trained_backbone.py:10)
  DomainSeg : DomainSegUpstream.{pretrainedBackBone.encoder,pretrainedContext,pretrainedNeck}.* DomainSegHead.*
              (domain_seg_network.py:11-14, domain_seg_upstream.py:10-20)
  EgoLanes  : BEVBackbone.encoder.* AutoSteerContext.* EgopathNeck.* EgoLanesHead.*
              (ego_lanes_network.py:14-26)

Init is NOT PyTorch's default: with default init the decoder contracts the signal
~0.6x/layer and argmax collapses to one class (SURVEY.md 8(d) init note), which
would make every parity check vacuous.  We use a variance-preserving init.
"""
import numpy as np

# torchvision efficientnet_b0 inverted-residual setting:
# (expand_ratio, kernel, stride, in_ch, out_ch, num_layers)  -- restated, see oracle/__init__.py
B0_STAGES = [
    (1, 3, 1, 32, 16, 1),
    (6, 3, 2, 16, 24, 2),
    (6, 5, 2, 24, 40, 2),
    (6, 3, 2, 40, 80, 3),
    (6, 5, 1, 80, 112, 3),
    (6, 5, 2, 112, 192, 4),
    (6, 3, 1, 192, 320, 1),
]
B0_STEM_OUT = 32
B0_LAST_OUT = 1280
BN_EPS = 1e-5

MODEL_KINDS = ("sceneseg", "scene3d", "domainseg", "egolanes")

PREFIX = {
    "sceneseg": dict(backbone="Backbone.encoder.", context="SceneContext.",
                     neck="SceneNeck.", head="SceneSegHead."),
    "scene3d": dict(backbone="PreTrainedBackbone.pretrainedBackBone.encoder.", context="DepthContext.",
                    neck="DepthNeck.", head="SuperDepthHead."),
    "domainseg": dict(backbone="DomainSegUpstream.pretrainedBackBone.encoder.",
                      context="DomainSegUpstream.pretrainedContext.",
                      neck="DomainSegUpstream.pretrainedNeck.", head="DomainSegHead."),
    "egolanes": dict(backbone="BEVBackbone.encoder.", context="AutoSteerContext.",
                     neck="EgopathNeck.", head="EgoLanesHead."),
}


def _bn(p, c):
    return [(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"),
            (p + ".running_mean", (c,), "bn_mean"), (p + ".running_var", (c,), "bn_var")]


def backbone_spec(prefix):
    """(key, shape, kind) for every tensor of torchvision efficientnet_b0().features."""
    s = [(prefix + "0.0.weight", (B0_STEM_OUT, 3, 3, 3), "conv")]
    s += _bn(prefix + "0.1", B0_STEM_OUT)
    for si, (e, k, st, cin, cout, n) in enumerate(B0_STAGES, start=1):
        for bi in range(n):
            ci = cin if bi == 0 else cout
            cexp = ci * e
            p = f"{prefix}{si}.{bi}.block."
            j = 0
            if e != 1:
                s.append((p + f"{j}.0.weight", (cexp, ci, 1, 1), "conv"))
                s += _bn(p + f"{j}.1", cexp)
                j += 1
            s.append((p + f"{j}.0.weight", (cexp, 1, k, k), "conv"))  # depthwise
            s += _bn(p + f"{j}.1", cexp)
            j += 1
            sq = max(1, ci // 4)
            s += [(p + f"{j}.fc1.weight", (sq, cexp, 1, 1), "conv"), (p + f"{j}.fc1.bias", (sq,), "bias"),
                  (p + f"{j}.fc2.weight", (cexp, sq, 1, 1), "conv"), (p + f"{j}.fc2.bias", (cexp,), "bias")]
            j += 1
            s.append((p + f"{j}.0.weight", (cout, cexp, 1, 1), "conv"))
            s += _bn(p + f"{j}.1", cout)
    s.append((prefix + "8.0.weight", (B0_LAST_OUT, 320, 1, 1), "conv"))
    s += _bn(prefix + "8.1", B0_LAST_OUT)
    return s


def _conv(p, cout, cin, k):
    return [(p + ".weight", (cout, cin, k, k), "conv"), (p + ".bias", (cout,), "bias")]


def _convT(p, cin, cout):
    return [(p + ".weight", (cin, cout, 2, 2), "convT"), (p + ".bias", (cout,), "bias")]


def _lin(p, cout, cin):
    return [(p + ".weight", (cout, cin), "linear"), (p + ".bias", (cout,), "bias")]


def context_spec(prefix, cin):
    """scene_context.py:14-22 / depth_context.py:14-22 (cin=1280); auto_steer_context.py:15-23 (cin=1456)."""
    return (_lin(prefix + "context_layer_0", 800, cin) + _lin(prefix + "context_layer_1", 800, 800)
            + _lin(prefix + "context_layer_2", 200, 800)
            + _conv(prefix + "context_layer_3", 128, 1, 3) + _conv(prefix + "context_layer_4", 256, 128, 3)
            + _conv(prefix + "context_layer_5", 512, 256, 3) + _conv(prefix + "context_layer_6", cin, 512, 3))


def neck_spec(prefix, cin):
    """scene_neck.py:11-24 / scene_3d_neck.py:11-24 (cin=1280); ego_path_neck.py:11-24 (cin=1456)."""
    return (_convT(prefix + "upsample_layer_0", cin, cin) + _conv(prefix + "skip_link_layer_0", cin, 80, 1)
            + _conv(prefix + "decode_layer_0", 768, cin, 3) + _conv(prefix + "decode_layer_1", 768, 768, 3)
            + _convT(prefix + "upsample_layer_1", 768, 768) + _conv(prefix + "skip_link_layer_1", 768, 40, 1)
            + _conv(prefix + "decode_layer_2", 512, 768, 3) + _conv(prefix + "decode_layer_3", 512, 512, 3)
            + _convT(prefix + "upsample_layer_2", 512, 512) + _conv(prefix + "skip_link_layer_2", 512, 24, 1)
            + _conv(prefix + "decode_layer_4", 512, 512, 3) + _conv(prefix + "decode_layer_5", 256, 512, 3))


def head_spec(kind, prefix):
    """scene_seg_head.py:11-19, scene_3d_head.py:13-21, domain_seg_head.py:11-19, ego_lanes_head.py:11-13."""
    if kind == "egolanes":
        return (_conv(prefix + "decode_layer_6", 256, 256, 3) + _conv(prefix + "decode_layer_7", 128, 256, 3)
                + _conv(prefix + "decode_layer_8", 3, 128, 3))
    c9, cout = {"sceneseg": (64, 3), "scene3d": (128, 1), "domainseg": (64, 1)}[kind]
    return (_convT(prefix + "upsample_layer_3", 256, 256) + _conv(prefix + "skip_link_layer_3", 256, 32, 1)
            + _conv(prefix + "decode_layer_6", 256, 256, 3) + _conv(prefix + "decode_layer_7", 128, 256, 3)
            + _convT(prefix + "upsample_layer_4", 128, 128)
            + _conv(prefix + "decode_layer_8", 128, 128, 3) + _conv(prefix + "decode_layer_9", c9, 128, 3)
            + _conv(prefix + "decode_layer_10", cout, c9, 3))


def context_channels(kind):
    return 1456 if kind == "egolanes" else 1280


def model_spec(kind):
    p = PREFIX[kind]
    c = context_channels(kind)
    return (backbone_spec(p["backbone"]) + context_spec(p["context"], c)
            + neck_spec(p["neck"], c) + head_spec(kind, p["head"]))


def _init(rng, shape, kind):
    f32 = np.float32
    if kind == "conv":
        fan_in = shape[1] * shape[2] * shape[3]
        return (rng.standard_normal(shape, dtype=f32) * f32(np.sqrt(2.0 / fan_in))).astype(f32)
    if kind == "linear":
        return (rng.standard_normal(shape, dtype=f32) * f32(np.sqrt(2.0 / shape[1]))).astype(f32)
    if kind == "convT":  # k2 s2: each output pixel sees exactly one tap -> fan_in = Cin
        return (rng.standard_normal(shape, dtype=f32) * f32(np.sqrt(1.0 / shape[0]))).astype(f32)
    if kind in ("bias", "bn_b", "bn_mean"):
        return (rng.standard_normal(shape, dtype=f32) * f32(0.1)).astype(f32)
    if kind in ("bn_w", "bn_var"):
        return rng.uniform(0.5, 1.5, size=shape).astype(f32)
    raise ValueError(kind)


def make_state_dict(kind, seed, spec=None):
    """Deterministic name->np.float32 array dict for ``kind`` (numpy PCG64, one stream, spec order)."""
    rng = np.random.default_rng(seed)
    return {k: _init(rng, shape, kd) for (k, shape, kd) in (spec or model_spec(kind))}


def share_backbone(dst_sd, dst_kind, src_sd, src_kind, also_context_neck=False):
    """Graft ``src``'s backbone (and optionally context+neck) tensors into ``dst`` under dst's key
    prefixes -- what Scene3DNetwork(pretrained) / DomainSegNetwork(pretrained) do by object sharing
    (pre_trained_backbone.py:10, domain_seg_upstream.py:10-20)."""
    parts = ["backbone"] + (["context", "neck"] if also_context_neck else [])
    for part in parts:
        sp, dp = PREFIX[src_kind][part], PREFIX[dst_kind][part]
        for k, v in src_sd.items():
            if k.startswith(sp):
                dst_sd[dp + k[len(sp):]] = v
    return dst_sd


def param_count(kind):
    return sum(int(np.prod(s)) for (_, s, kd) in model_spec(kind) if kd not in ("bn_mean", "bn_var"))
