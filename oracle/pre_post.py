"""Preprocess + decode restatements in numpy (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Preprocess
----------
Reference: OnnxRuntimeBackend::preprocess (VisionPilot/middleware_recipes/common/backends/
onnx_runtime_backend.cpp:41-60: cv::resize INTER_LINEAR to 640x320, convertTo 1/255, subtract/divide
with BGR-ordered ImageNet constants, split -> planes B,G,R); EgoLanesOnnxEngine::preprocessEgoLanes
(VisionPilot/production_release/src/inference/onnxruntime_engine.cpp:72-102: resize, BGR->RGB, /255,
(x-MEAN[c])/STD[c] -> planes R,G,B); Python ToTensor+Normalize (Models/inference/scene_seg_infer.py:15-20).

cv::resize is OpenCV (third-party, not under /root/reference, not installed here): PARITY UNPINNED.
Our definition -- which the engine must reproduce BIT-EXACTLY -- is an integer bilinear modelled on
OpenCV's published u8 INTER_LINEAR scheme: half-pixel centres, edge clamp, 11-bit fixed-point taps
(2048 = 1.0), horizontal pass in int32, vertical pass
    dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
At scale 1 it is the identity.  Float part: t = q / 255 ; y = (t - mean) / std in IEEE fp32 (exactly
what torchvision's ToTensor/Normalize do).

Decode
------
argmax (first max wins): scene_seg_infer.py:52-55 torch.max(dim=2); C++ strict '>' from -1e9,
ROS2/models/src/run_model_node.cpp:144-163 (class 1 -> 255 else 0); binary '>0 -> 255' :164-171 and
domain_seg_infer.py:54-58 (0/1 floats); lane priority mask {other 2 > right 1 > left 0 > none 255}
common/visualizers/cuda_visualization_kernels.cu:45-75; LaneSegmentation 0/1 float planes
onnxruntime_engine.cpp:151-192; cv::resize INTER_NEAREST run_model_node.cpp:176-177;
depth cv::resize INTER_LINEAR (float) run_model_node.cpp:100-104.
"""
import numpy as np

NET_H, NET_W = 320, 640
MEAN_RGB = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD_RGB = np.array([0.229, 0.224, 0.225], dtype=np.float32)
COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


def linear_taps_u8(src, dst):
    """Per-destination (index0, index1, tap0, tap1) with 11-bit fixed-point taps; index1 is clamped and
    gets tap 0 at the borders, so callers never read out of range."""
    scale = np.float64(src) / np.float64(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    a1 = np.rint(f * np.float32(COEF_ONE)).astype(np.int32)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_ONE)).astype(np.int32)
    s1 = np.minimum(s + 1, src - 1).astype(np.int32)
    return s, s1, a0, a1


def resize_bilinear_u8(img, out_h=NET_H, out_w=NET_W):
    """img: HxWxC uint8 -> out_h x out_w x C uint8 (our integer bilinear)."""
    img = np.ascontiguousarray(img)
    h, w, _ = img.shape
    x0, x1, a0, a1 = linear_taps_u8(w, out_w)
    y0, y1, b0, b1 = linear_taps_u8(h, out_h)
    s = img.astype(np.int32)
    hp = s[:, x0, :] * a0[None, :, None] + s[:, x1, :] * a1[None, :, None]     # H x out_w x C, int32
    r0 = hp[y0] >> 4
    r1 = hp[y1] >> 4
    v = (((b0[:, None, None] * r0) >> 16) + ((b1[:, None, None] * r1) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------- Pillow's antialiased resample
# Image.resize(size, Image.BILINEAR) -- Models/visualizations/AutoDrive/video_visualization.py:29-33 (the AutoDrive frame path; also
# image_visualization.py:30) -- and Image.resize(size) with Pillow's default filter (BICUBIC) in the scene networks' visualisation
# scripts (Models/visualizations/Scene3D/video_visualization.py:87, DomainSeg/video_visualization.py:112).  Pillow is a third-party
# dependency of the reference (requirements: pillow, unpinned); it IS installed here (12.2.0), so this restatement of its
# src/libImaging/Resample.c (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) is PINNED
# bit for bit against PIL itself (tests/test_oracle_golden.py::test_pil_resample_is_pillows).  The filter support scales with the
# down-scaling factor (that is the antialiasing), coefficients are normalised in double and quantised to 22 fractional bits, each
# pass accumulates in int32 from 1 << 21 and clips through `>> 22` to u8: the horizontal pass writes a u8 image, the vertical pass
# reads it (two roundings).
PIL_PRECISION_BITS = 32 - 8 - 2
PIL_BILINEAR, PIL_BICUBIC = 2, 3          # PIL.Image.Resampling values


def _pil_filter(x, which):
    x = np.abs(x)
    if which == PIL_BILINEAR:
        return np.where(x < 1.0, 1.0 - x, 0.0)
    a = -0.5
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def pil_resample_coeffs(in_size, out_size, which=PIL_BILINEAR):
    """-> (bounds [out][2] = first source index, tap count; coefficients [out][ksize] int32 with 22 fractional bits)."""
    support0 = 1.0 if which == PIL_BILINEAR else 2.0
    scale = filterscale = np.float64(in_size) / np.float64(out_size)
    if filterscale < 1.0:
        filterscale = np.float64(1.0)
    support = support0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = np.float64(1.0) / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _pil_filter((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss, which)
        ww = np.float64(0.0)
        for v in w:          # Pillow sums in tap order
            ww += v
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, (-0.5 + w * (1 << PIL_PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PIL_PRECISION_BITS)).astype(np.int64))
        kk[xx, :xmax] = q
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pil_pass(img, bounds, kk, axis):
    """one 8-bit pass along `axis` of an HxWxC u8 image"""
    src = np.moveaxis(img.astype(np.int32), axis, 0)
    out = np.empty((len(bounds),) + src.shape[1:], dtype=np.uint8)
    for i, (lo, n) in enumerate(bounds):
        acc = np.full(src.shape[1:], 1 << (PIL_PRECISION_BITS - 1), dtype=np.int32)
        for t in range(n):
            acc = acc + src[lo + t] * kk[i, t]
        out[i] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_pil_u8(img, out_h, out_w, which=PIL_BILINEAR):
    """HxWxC uint8 -> out_h x out_w x C uint8, Pillow's Image.resize((out_w, out_h), which): horizontal pass, then vertical;
    a pass whose size does not change is skipped (ImagingResample: need_horizontal / need_vertical)."""
    img = np.ascontiguousarray(img)
    h, w, _ = img.shape
    if w != out_w:
        img = _pil_pass(img, *pil_resample_coeffs(w, out_w, which), axis=1)
    if h != out_h:
        img = _pil_pass(img, *pil_resample_coeffs(h, out_h, which), axis=0)
    return img


def normalize_planes(img_u8, input_is_bgr, planes_rgb, norm_form="torchvision"):
    """uint8 HxWx3 -> fp32 3xHxW.  ``planes_rgb`` False gives B,G,R plane order (middleware 'common'
    backends), True gives R,G,B (EgoLanes engine, Python).  Constants always follow the plane's colour.
    norm_form: "torchvision" = q / 255 (to_tensor, Models/inference/scene_seg_infer.py:15-20); "opencv" = q * fl(1/255), the C++
    front-ends' cv::Mat::convertTo(CV_32FC3, 1.0 / 255.0) (VisionPilot/middleware_recipes/common/backends/onnx_runtime_backend.cpp:45,
    tensorrt_backend.cpp:164, production_release/src/inference/onnxruntime_engine.cpp:85) -- OpenCV's cvtScale for 8U -> 32F multiplies
    by the scale converted to float (one rounding); then fp32 subtract and IEEE divide in both (cv::subtract / cv::divide with the
    Scalar converted to float, onnx_runtime_backend.cpp:48-49; the scalar loop (x - MEAN[c]) / STD[c], onnxruntime_engine.cpp:98)."""
    if norm_form == "opencv":
        x = img_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    else:
        x = img_u8.astype(np.float32) / np.float32(255.0)
    rgb = x[..., ::-1] if input_is_bgr else x
    rgb = (rgb - MEAN_RGB) / STD_RGB
    out = rgb if planes_rgb else rgb[..., ::-1]
    return np.ascontiguousarray(out.transpose(2, 0, 1)).astype(np.float32)


def preprocess(frame_u8, input_is_bgr=True, planes_rgb=False, out_h=NET_H, out_w=NET_W, resize="cv", norm_form="torchvision"):
    """Any-size u8 frame -> 1x3xout_hxout_w fp32 network input (320x640 for the scene networks, 512x1024 for AutoDrive).
    resize: "cv" = the integer bilinear above (the C++ nodes' cv::resize), "pil_bilinear" / "pil_bicubic" = Pillow's antialiased
    resample (the Python scripts' Image.resize: AutoDrive's BILINEAR, the scene visualisations' default BICUBIC)."""
    if resize == "cv":
        small = resize_bilinear_u8(frame_u8, out_h, out_w)
    else:
        small = resize_pil_u8(frame_u8, out_h, out_w, {"pil_bilinear": PIL_BILINEAR, "pil_bicubic": PIL_BICUBIC}[resize])
    return normalize_planes(small, input_is_bgr, planes_rgb, norm_form)[None]


# ------------------------------------------------------------------------------------------ decode
def argmax_classes(logits):
    """CxHxW fp32 -> HxW int64, lowest index wins ties (strict '>' scan == torch.max)."""
    best = np.zeros(logits.shape[1:], dtype=np.int64)
    score = np.full(logits.shape[1:], np.float32(-1e9), dtype=np.float32)
    for c in range(logits.shape[0]):
        m = logits[c] > score
        score = np.where(m, logits[c], score)
        best = np.where(m, c, best)
    return best


def seg_mask_u8(logits):
    """run_model_node.cpp:144-171: C>1 -> 255 where argmax==1 ; C==1 -> 255 where >0."""
    if logits.shape[0] > 1:
        return np.where(argmax_classes(logits) == 1, 255, 0).astype(np.uint8)
    return np.where(logits[0] > 0.0, 255, 0).astype(np.uint8)


def binary_float(logits):
    """domain_seg_infer.py:54-58: HxWx1 float 0/1."""
    return (logits > 0).astype(np.float32).transpose(1, 2, 0)


def egolanes_priority_mask(logits):
    """cuda_visualization_kernels.cu:45-75."""
    b0, b1, b2 = (logits[i] > 0.0 for i in range(3))
    return np.where(b2, 2, np.where(b1, 1, np.where(b0, 0, 255))).astype(np.uint8)


def egolanes_planes(logits, threshold=0.0):
    """onnxruntime_engine.cpp:151-192: three fp32 0/1 planes."""
    return (logits > np.float32(threshold)).astype(np.float32)


def nearest_index(src, dst):
    """OpenCV resizeNN index table: min(floor(x * (1/(dst/src))), src-1), computed in double."""
    inv = np.float64(dst) / np.float64(src)
    ifx = np.float64(1.0) / inv
    return np.minimum(np.floor(np.arange(dst, dtype=np.float64) * ifx).astype(np.int64), src - 1)


def resize_nearest_u8(mask, out_h, out_w):
    return mask[nearest_index(mask.shape[0], out_h)][:, nearest_index(mask.shape[1], out_w)]


def linear_taps_f32(src, dst):
    scale = np.float64(src) / np.float64(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    return s, np.minimum(s + 1, src - 1).astype(np.int32), (np.float32(1.0) - f).astype(np.float32), f


def resize_bilinear_f32(plane, out_h, out_w):
    """Float bilinear, half-pixel centres, edge clamp, horizontal then vertical, fp32 (no FMA contraction)."""
    x0, x1, a0, a1 = linear_taps_f32(plane.shape[1], out_w)
    y0, y1, b0, b1 = linear_taps_f32(plane.shape[0], out_h)
    p = plane.astype(np.float32)
    hp = (p[:, x0] * a0[None, :]).astype(np.float32) + (p[:, x1] * a1[None, :]).astype(np.float32)
    return ((hp[y0] * b0[:, None]).astype(np.float32) + (hp[y1] * b1[:, None]).astype(np.float32)).astype(np.float32)


# ------------------------------------------------------------------------------------ synthetic data
from autoware_vision_pilot_amd.synthetic import synthetic_frame  # noqa: E402,F401  (shared data generation)


# ------------------------------------------------------------------------------------------ visualisation
_VIZ_LUT = None


def viz_lut():
    """createColorMask colour tables, BGR (masks_visualization_engine.cpp:41-58): [viz_type][label] -> 3 bytes."""
    global _VIZ_LUT
    if _VIZ_LUT is None:
        lut = np.zeros((3, 256, 3), dtype=np.uint8)
        lut[0, 1:] = (0, 0, 255)                                    # "scene": inRange(1, 255) -> red
        lut[1, 0], lut[1, 255] = (255, 93, 61), (145, 28, 255)      # "domain"
        lut[2, 0], lut[2, 1], lut[2, 2] = (255, 0, 0), (255, 0, 200), (0, 153, 0)   # "egolanes"
        _VIZ_LUT = lut
    return _VIZ_LUT


def visualize_mask(mask_u8, frame_bgr, viz_type):
    """MasksVisualizationEngine::visualize (:11-38): colour mask -> INTER_NEAREST resize to the frame -> addWeighted(0.5, 0.5).
    cv::addWeighted on u8 is saturate_cast<uchar>(cvRound(float)): round half to even."""
    h, w = frame_bgr.shape[:2]
    color = viz_lut()[viz_type][resize_nearest_u8(mask_u8, h, w)]
    s = color.astype(np.int32) + frame_bgr.astype(np.int32)
    half = s >> 1
    return np.where(s & 1, half + (half & 1), half).astype(np.uint8)


def viridis_lut_bgr():
    """cv::COLORMAP_VIRIDIS as 256 x BGR u8: OpenCV's colormap.cpp embeds matplotlib's published viridis data and converts it
    with convertTo(CV_8U, 255); restated from matplotlib's own copy (PARITY UNPINNED against OpenCV, absent here).  The engine
    carries the same table as csrc/viridis_lut.inc (tools/gen_viridis_lut.py); tests compare the two."""
    from matplotlib import _cm_listed

    rgb = np.asarray(_cm_listed._viridis_data, dtype=np.float32)
    return np.rint(rgb * np.float32(255.0)).astype(np.uint8)[:, ::-1].copy()


def visualize_depth(depth_f32):
    """depth_visualization_engine.cpp:9-26: minMaxLoc -> convertTo(CV_8UC1, 255/(max-min), -min*255/(max-min)) ->
    applyColorMap(VIRIDIS).  convertTo: alpha / beta formed in double, applied in float as a fused multiply-add, cvRound
    (half-to-even), saturate.  The fma is evaluated here in float64 (product of two float32 is exact there)."""
    d = np.ascontiguousarray(depth_f32, dtype=np.float32)
    mn, mx = float(d.min()), float(d.max())
    if mx > mn:
        a = np.float32(255.0 / (mx - mn))
        b = np.float32(-mn * 255.0 / (mx - mn))
        t = (d.astype(np.float64) * np.float64(a) + np.float64(b)).astype(np.float32)
        u = np.clip(np.rint(t), 0, 255).astype(np.uint8)
    else:
        u = np.zeros(d.shape, dtype=np.uint8)
    return viridis_lut_bgr()[u]
