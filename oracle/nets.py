"""Functional torch-CPU fp32 restatement of the four networks (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function takes a flat ``sd`` (name -> torch.float32 tensor, reference key layout) and a key prefix.
Citations are /root/reference paths.
"""
import torch
import torch.nn.functional as F

from .weights import B0_STAGES, BN_EPS, PREFIX, context_channels


def to_torch(sd_np):
    return {k: torch.from_numpy(v) for k, v in sd_np.items()}


# ----------------------------------------------------------------------------------------------
# EfficientNet-B0 .features  (torchvision, third-party -- PARITY UNPINNED, see oracle/__init__.py)
# call sites: Models/model_components/backbone.py:9 (construction), :13-22 (tap points)
# ----------------------------------------------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=BN_EPS)


def _cna(sd, p, x, stride=1, groups=1, act=True):
    """Conv2dNormActivation: conv(no bias, 'same'-style symmetric padding (k-1)//2) -> BN(eval) -> SiLU."""
    w = sd[p + ".0.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=(w.shape[-1] - 1) // 2, groups=groups)
    y = _bn(sd, p + ".1", y)
    return F.silu(y) if act else y


def _squeeze_excite(sd, p, x):
    s = x.mean(dim=(2, 3), keepdim=True)                       # AdaptiveAvgPool2d(1)
    s = F.silu(F.conv2d(s, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"]))
    s = torch.sigmoid(F.conv2d(s, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"]))
    return x * s


def _mbconv(sd, p, x, expand, stride, cin, cout):
    y, j = x, 0
    if expand != 1:
        y = _cna(sd, f"{p}{j}", y)
        j += 1
    y = _cna(sd, f"{p}{j}", y, stride=stride, groups=y.shape[1])  # depthwise
    j += 1
    y = _squeeze_excite(sd, f"{p}{j}", y)
    j += 1
    y = _cna(sd, f"{p}{j}", y, act=False)                          # project, no activation
    if stride == 1 and cin == cout:
        y = y + x                                                  # stochastic depth == identity in eval
    return y


def backbone(sd, prefix, image):
    """backbone.py:11-22 -> [l0, l2, l3, l4, l8] = f0 32x160x320, f1 24x80x160, f2 40x40x80, f3 80x20x40, f4 1280x10x20."""
    stage_out = [_cna(sd, prefix + "0", image, stride=2)]
    x = stage_out[0]
    for si, (e, k, st, cin, cout, n) in enumerate(B0_STAGES, start=1):
        for bi in range(n):
            x = _mbconv(sd, f"{prefix}{si}.{bi}.block.", x, e, st if bi == 0 else 1, cin if bi == 0 else cout, cout)
        stage_out.append(x)
    stage_out.append(_cna(sd, prefix + "8", x))
    return [stage_out[i] for i in (0, 2, 3, 4, 8)]


# ----------------------------------------------------------------------------------------------
# context / neck / heads  (reference Models/model_components -- PINNED by pin_against_reference.py)
# ----------------------------------------------------------------------------------------------
def _c(sd, p, x, pad):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad)


def _up(sd, p, x):
    return F.conv_transpose2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=2)


def context(sd, p, f):
    """scene_context.py:25-57 == depth_context.py:25-57 == auto_steer_context.py:26-59 (dropout = identity in eval)."""
    v = f.mean(dim=(2, 3))
    v = F.gelu(F.linear(v, sd[p + "context_layer_0.weight"], sd[p + "context_layer_0.bias"]))
    v = F.gelu(F.linear(v, sd[p + "context_layer_1.weight"], sd[p + "context_layer_1.bias"]))
    v = torch.sigmoid(F.linear(v, sd[p + "context_layer_2.weight"], sd[p + "context_layer_2.bias"]))
    m = v.reshape(1, 1, 10, 20)
    for i in (3, 4, 5, 6):
        m = F.gelu(_c(sd, p + f"context_layer_{i}", m, 1))
    return m * f + f


def neck(sd, p, ctx, feats):
    """scene_neck.py:26-60 == scene_3d_neck.py:26-60 == ego_path_neck.py:26-60."""
    x = ctx
    for blk, skip in ((0, feats[3]), (1, feats[2]), (2, feats[1])):
        x = _up(sd, p + f"upsample_layer_{blk}", x) + _c(sd, p + f"skip_link_layer_{blk}", skip, 0)
        x = F.gelu(_c(sd, p + f"decode_layer_{2 * blk}", x, 1))
        x = F.gelu(_c(sd, p + f"decode_layer_{2 * blk + 1}", x, 1))
    return x


def head_full_res(sd, p, nk, feats):
    """scene_seg_head.py:21-44 / scene_3d_head.py:23-47 / domain_seg_head.py:21-44 (same wiring, different widths)."""
    x = _up(sd, p + "upsample_layer_3", nk) + _c(sd, p + "skip_link_layer_3", feats[0], 0)
    x = F.gelu(_c(sd, p + "decode_layer_6", x, 1))
    x = F.gelu(_c(sd, p + "decode_layer_7", x, 1))
    x = _up(sd, p + "upsample_layer_4", x)
    x = F.gelu(_c(sd, p + "decode_layer_8", x, 1))
    x = F.gelu(_c(sd, p + "decode_layer_9", x, 1))
    return _c(sd, p + "decode_layer_10", x, 1)


def head_egolanes(sd, p, nk):
    """ego_lanes_head.py:18-26."""
    x = F.gelu(_c(sd, p + "decode_layer_6", nk, 1))
    x = F.gelu(_c(sd, p + "decode_layer_7", x, 1))
    return _c(sd, p + "decode_layer_8", x, 1)


def feature_fusion(feats):
    """backbone_feature_fusion.py:13-38: MaxPool2x2 applied 4/3/2/1x to f0..f3, concat with f4 -> 1456x10x20."""
    out = []
    for f, n in zip(feats[:4], (4, 3, 2, 1)):
        for _ in range(n):
            f = F.max_pool2d(f, 2, 2)
        out.append(f)
    out.append(feats[4])
    return torch.cat(out, dim=1)


@torch.no_grad()
def forward(kind, sd, image, return_intermediates=False):
    """scene_seg_network.py:24-29, scene_3d_network.py:25-30, domain_seg_network.py:17-19 (+domain_seg_upstream.py:22-26),
    ego_lanes_network.py:30-36.  image: 1x3x320x640 fp32 -> logits 1xCxhxw fp32."""
    p = PREFIX[kind]
    feats = backbone(sd, p["backbone"], image)
    deep = feature_fusion(feats) if kind == "egolanes" else feats[4]
    assert deep.shape[1] == context_channels(kind)
    ctx = context(sd, p["context"], deep)
    nk = neck(sd, p["neck"], ctx, feats)
    out = head_egolanes(sd, p["head"], nk) if kind == "egolanes" else head_full_res(sd, p["head"], nk, feats)
    if return_intermediates:
        return out, dict(feats=feats, deep=deep, ctx=ctx, neck=nk)
    return out
