"""AutoDrive (BASELINE configs[4]): functional torch-CPU fp32 restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the reference's own ``nn.Module``s by
``oracle/pin_against_reference.py`` (the modules are importable in the build container); vectors under
``tests/golden/autodrive*.npz``.

Reference (paths relative to /root/reference/Models/model_components):
  autodrive/autodrive_network.py:32-36   forward(prev, curr) = head(backbone(prev), backbone(curr))
  autodrive/autodrive_backbone.py:8-48   p1..p5: Conv(k3,s2)+BN+SiLU, CTX, SPPF, C2PSA
  autodrive/autodrive_head.py:70-87      cat -> 3 x (conv3x3 + SiLU) -> flatten -> FC 768 -> FC 512 -> 3 heads
  common_layers.py:5-17 (Conv), :183-227 (CTX), :230-243 (SPPF), :246-257 (C2PSA), :78-117 (Attention, PSABlock)
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3                     # common_layers.py:10
WIDTH = [3, 16, 32, 64, 128, 256]  # autodrive_network.py:12
NET_H, NET_W = 512, 1024           # autodrive_network.py:8-9
# (stage, cin of the strided conv, cout of the strided conv, CTX out, CTX map h, w)   autodrive_backbone.py:13-40
STAGES = [("p2", 16, 32, 64, 128, 256), ("p3", 64, 64, 128, 64, 128), ("p4", 128, 128, 128, 32, 64), ("p5", 128, 256, 256, 16, 32)]


# ------------------------------------------------------------------------------------------------ parameters
def _conv_bn(p, cout, cin, k, groups=1):
    return [(p + ".conv.weight", (cout, cin // groups, k, k), "conv"), (p + ".norm.weight", (cout,), "bn_w"),
            (p + ".norm.bias", (cout,), "bn_b"), (p + ".norm.running_mean", (cout,), "bn_mean"),
            (p + ".norm.running_var", (cout,), "bn_var")]


def _ctx(p, cin, cout, h, w, r=2):
    return [(p + ".exp0.weight", (h * w, cin, 3), "conv1d"), (p + ".exp0.bias", (h * w,), "bias"),
            (p + ".ctx0.weight", (cin // r, 1, 3, 3), "conv"), (p + ".ctx0.bias", (cin // r,), "bias"),
            (p + ".ctx1.weight", (cin, cin // r, 3, 3), "conv"), (p + ".ctx1.bias", (cin,), "bias"),
            (p + ".ctx2.weight", (cout, cin, 3, 3), "conv"), (p + ".ctx2.bias", (cout,), "bias")]


def model_spec():
    """(key, shape, init kind) in the reference ``state_dict`` order (num_batches_tracked omitted)."""
    s = _conv_bn("backbone.p1", 16, 3, 3)
    for name, cin, cmid, cout, h, w in STAGES:
        s += _conv_bn(f"backbone.{name}.0", cmid, cin, 3)
        s += _ctx(f"backbone.{name}.1", cmid, cout, h, w)
    s += _conv_bn("backbone.p5.2.cv1", 128, 256, 1) + _conv_bn("backbone.p5.2.cv2", 256, 512, 1)   # SPPF
    s += _conv_bn("backbone.p5.3.cv1", 256, 256, 1) + _conv_bn("backbone.p5.3.cv2", 256, 256, 1)   # C2PSA
    a = "backbone.p5.3.middle_block"
    s += _conv_bn(a + ".conv1.qkv", 256, 128, 1) + _conv_bn(a + ".conv1.conv1", 128, 128, 3, groups=128)
    s += _conv_bn(a + ".conv1.conv2", 128, 128, 1) + _conv_bn(a + ".conv2.0", 256, 128, 1) + _conv_bn(a + ".conv2.1", 128, 256, 1)
    s += [("head.conv_1.weight", (256, 512, 3, 3), "conv"), ("head.conv_1.bias", (256,), "bias"),
          ("head.conv_2.weight", (64, 256, 3, 3), "conv"), ("head.conv_2.bias", (64,), "bias"),
          ("head.conv_3.weight", (2, 64, 3, 3), "conv"), ("head.conv_3.bias", (2,), "bias"),
          ("head.fc1.0.weight", (768, 1024), "linear"), ("head.fc1.0.bias", (768,), "bias"),
          ("head.fc2.0.weight", (512, 768), "linear"), ("head.fc2.0.bias", (512,), "bias"),
          ("head.distance_head.0.weight", (1, 512), "linear"), ("head.distance_head.0.bias", (1,), "bias"),
          ("head.curvature_head.0.weight", (1, 512), "linear"), ("head.curvature_head.0.bias", (1,), "bias"),
          ("head.flag_head.weight", (1, 512), "linear"), ("head.flag_head.bias", (1,), "bias")]
    return s


def make_state_dict(seed):
    """Seeded init (scheme of oracle/weights.py) tuned so every stage carries O(1) signal and no output saturates:
    CTX gates multiplicatively (c4*x + x, with c4 driven by mean(x)), so Kaiming gains square the scale per stage
    (measured: P5 std 2e4, all three outputs clipped -> vacuous parity).  exp0 and ctx2 weights get gain 0.5 and the
    distance head's bias +1 (pre-ReLU value ~ +0.5): per-stage std 0.9 / 0.6 / 0.4 / 0.2, P5 std ~2, outputs
    (d, curvature, flag) ~ (0.5, 0.5, -1.1) on the fixture frames."""
    from .weights import _init
    rng = np.random.default_rng(seed)
    out = {}
    for k, shape, kind in model_spec():
        if kind == "conv1d":  # Conv1d on a length-1 sequence: only the centre tap ever multiplies data
            out[k] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(0.5 * np.sqrt(2.0 / shape[1]))).astype(np.float32)
        else:
            out[k] = _init(rng, shape, kind)
        if k.endswith("ctx2.weight"):
            out[k] = out[k] * np.float32(0.5)
        if k == "head.distance_head.0.bias":
            out[k] = out[k] + np.float32(1.0)
    return out


def param_count():
    return sum(int(np.prod(s)) for _, s, _ in model_spec())


def quantize_fp8_e4m3(sd):
    """BASELINE configs[4] "fp8 weights": per-output-channel symmetric e4m3 quantisation of every conv / linear weight
    (OCP e4m3fn: max 448, 3 mantissa bits, subnormals), returned DE-quantised to fp32 so oracle and engine consume
    the same numbers.  BN parameters and biases stay fp32."""
    out = {}
    for k, v in sd.items():
        if not k.endswith(".weight") or v.ndim < 2:
            out[k] = v
            continue
        flat = v.reshape(v.shape[0], -1).astype(np.float64)
        scale = np.maximum(np.abs(flat).max(axis=1, keepdims=True), 1e-30) / 448.0
        x = flat / scale
        mag = np.abs(x)
        e = np.floor(np.log2(np.maximum(mag, 2.0 ** -9)))
        e = np.clip(e, -6, 8)                      # normal exponents -6..8; below 2^-6 the step stays 2^-9 (subnormals)
        step = 2.0 ** (e - 3)
        q = np.minimum(np.round(mag / step) * step, 448.0) * np.sign(x)
        out[k] = (q * scale).reshape(v.shape).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------------------- layers
def _conv_bn_act(sd, p, x, stride=1, groups=1, act=True):
    """common_layers.py:5-14  Conv = conv(no bias, pad k//2) -> BN(eval, eps 1e-3) -> activation."""
    w = sd[p + ".conv.weight"]
    if p + ".norm.weight" not in sd:  # exporter-folded form (weights.py load_onnx_state_dict): conv carries the bias
        y = F.conv2d(x, w, sd[p + ".conv.bias"], stride=stride, padding=w.shape[-1] // 2, groups=groups)
        return F.silu(y) if act else y
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2, groups=groups)
    y = F.batch_norm(y, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"], sd[p + ".norm.weight"], sd[p + ".norm.bias"],
                     training=False, eps=BN_EPS)
    return F.silu(y) if act else y


def ctx_block(sd, p, x, h, w):
    """common_layers.py:202-227."""
    b = x.shape[0]
    y = x.mean(dim=(2, 3), keepdim=True)                                  # :206
    c0 = F.silu(F.conv1d(y.squeeze(-1), sd[p + ".exp0.weight"], sd[p + ".exp0.bias"], padding=1))   # :210-211
    c1 = F.silu(c0.view(b, 1, h, w))                                      # :212-213 (second SiLU)
    c2 = F.silu(F.conv2d(c1, sd[p + ".ctx0.weight"], sd[p + ".ctx0.bias"], padding=1))             # :216-217
    c4 = F.silu(F.conv2d(c2, sd[p + ".ctx1.weight"], sd[p + ".ctx1.bias"], padding=1))             # :218-219
    c4 = c4 * x + x                                                       # :222
    return F.conv2d(F.silu(c4), sd[p + ".ctx2.weight"], sd[p + ".ctx2.bias"], padding=1)           # :224-225


def sppf(sd, p, x):
    """common_layers.py:239-243."""
    x = _conv_bn_act(sd, p + ".cv1", x)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    return _conv_bn_act(sd, p + ".cv2", torch.cat((x, y1, y2, y3), 1))


def attention(sd, p, x, num_head):
    """common_layers.py:92-104."""
    b, c, h, w = x.shape
    dim_head = c // num_head
    dim_key = dim_head // 2
    qkv = _conv_bn_act(sd, p + ".qkv", x, act=False).view(b, num_head, dim_key * 2 + dim_head, h * w)
    q, k, v = qkv.split([dim_key, dim_key, dim_head], dim=2)
    attn = ((q.transpose(-2, -1) @ k) * dim_key ** -0.5).softmax(dim=-1)
    y = (v @ attn.transpose(-2, -1)).view(b, c, h, w) + _conv_bn_act(sd, p + ".conv1", v.reshape(b, c, h, w), groups=c, act=False)
    return _conv_bn_act(sd, p + ".conv2", y, act=False)


def psa_block(sd, p, x, num_head):
    """common_layers.py:107-118."""
    x = x + attention(sd, p + ".conv1", x, num_head)
    return x + _conv_bn_act(sd, p + ".conv2.1", _conv_bn_act(sd, p + ".conv2.0", x), act=False)


def c2psa(sd, p, x):
    """common_layers.py:246-257."""
    c_ = x.shape[1] // 2
    a, y = _conv_bn_act(sd, p + ".cv1", x).split((c_, c_), dim=1)
    y = psa_block(sd, p + ".middle_block", y, c_ // 64)
    return _conv_bn_act(sd, p + ".cv2", torch.cat((a, y), 1))


def backbone(sd, x, return_intermediates=False):
    """autodrive_backbone.py:42-48: 1x3x512x1024 -> P5 1x256x16x32."""
    inter = {}
    y = _conv_bn_act(sd, "backbone.p1", x, stride=2)
    inter["p1"] = y
    for name, _, _, _, h, w in STAGES:
        y = _conv_bn_act(sd, f"backbone.{name}.0", y, stride=2)
        y = ctx_block(sd, f"backbone.{name}.1", y, h, w)
        inter[name + "_ctx"] = y
    y = sppf(sd, "backbone.p5.2", y)
    inter["sppf"] = y
    y = c2psa(sd, "backbone.p5.3", y)
    inter["p5"] = y
    return (y, inter) if return_intermediates else y


def head(sd, f_prev, f_curr):
    """autodrive_head.py:70-87 (dropout = identity in eval)."""
    x = torch.cat([f_prev, f_curr], dim=1)
    for i in (1, 2, 3):
        x = F.silu(F.conv2d(x, sd[f"head.conv_{i}.weight"], sd[f"head.conv_{i}.bias"], padding=1))
    x = x.flatten(1)
    x = F.silu(F.linear(x, sd["head.fc1.0.weight"], sd["head.fc1.0.bias"]))
    x = F.silu(F.linear(x, sd["head.fc2.0.weight"], sd["head.fc2.0.bias"]))
    d = F.relu(F.linear(x, sd["head.distance_head.0.weight"], sd["head.distance_head.0.bias"]))
    c = torch.tanh(F.linear(x, sd["head.curvature_head.0.weight"], sd["head.curvature_head.0.bias"]))
    f = F.linear(x, sd["head.flag_head.weight"], sd["head.flag_head.bias"])
    return d, c, f


def forward(sd, image_prev, image_curr):
    """autodrive_network.py:32-36 -> (d_norm, curvature, flag_logit), each [1,1]."""
    with torch.no_grad():
        return head(sd, backbone(sd, image_prev), backbone(sd, image_curr))
