"""AutoSpeed detector pre/post-processing restated in numpy (TEST INFRASTRUCTURE -- see oracle/__init__.py).

SURVEY.md section 8 row N4, second half: the letterbox preprocess and the class-aware NMS decode that wrap the AutoSpeed
detector in the reference's C++ engines.  The detector network itself is a different model family and is NOT part of the hot path.

Reference (VisionPilot/middleware_recipes/common/backends/autospeed/):
  * preprocessAutoSpeed      onnxruntime_engine.cpp:71-113   scale = min(W/w, H/h) in fp32, new size = int(orig * scale) (truncation),
                             cv::resize INTER_LINEAR, canvas of (114,114,114), paste at ((W-new_w)/2, (H-new_h)/2) (integer division),
                             convertTo 1/255, planes R, G, B.
  * postProcess              onnxruntime_engine.cpp:170-237  per box: strict-'>' argmax of the class scores FROM 0 (class -1 if no score is
                             positive), `max_score < conf_thresh -> skip`, xywh -> xyxy in letterbox space, back to the image
                             ((v - pad) / scale), clamp to [0, orig], then applyNMS.
  * computeIoU / applyNMS    onnxruntime_engine.cpp:239-290  sort by confidence (descending), greedy, suppresses SAME-CLASS boxes with IoU > thresh.
  * Detection                detection.hpp:8-12              {x1, y1, x2, y2, confidence, class_id}.

PARITY UNPINNED (and said so): the reference's only implementation of these stages is the C++ engine file above, which needs the ONNX Runtime and
OpenCV headers -- neither is in this image, so it is unbuildable here and this restatement is what the device stages are checked against (rounds 3-5
compiled that file against hand-written stand-in headers and kept its outputs as a fixture; a build on stand-ins is not the reference, so the harness,
the stand-ins and the fixture were removed in round 6).  The reference's own tests hold no vectors for these stages.  Third-party
arithmetic: cv::resize and convertTo are OpenCV; the resize is the integer bilinear of oracle/pre_post.py, `convertTo(CV_32F, 1/255)` is taken as
fp32(u8) * fp32(1/255).  The reference's std::sort is not stable, so the order
of EQUAL confidences is unspecified there (libstdc++'s introsort reorders them); here (and in the engine) ties keep the box order.  The
float arithmetic follows the C++ expression for expression in IEEE fp32 without contraction (x86-64 baseline has no FMA).
"""
import numpy as np

from . import pre_post

F = np.float32


def letterbox_geometry(orig_h, orig_w, net_h=640, net_w=640):
    """scale_, new size, pad -- onnxruntime_engine.cpp:78-98."""
    scale = min(F(net_w) / F(orig_w), F(net_h) / F(orig_h))      # static_cast<float>(target) / int -> fp32 division
    new_w, new_h = int(F(orig_w) * scale), int(F(orig_h) * scale)
    return F(scale), new_w, new_h, (net_w - new_w) // 2, (net_h - new_h) // 2


def preprocess(frame_bgr_u8, net_h=640, net_w=640, resize_fn=None):
    """[3][net_h][net_w] fp32 planes R, G, B in [0, 1] + (scale, pad_x, pad_y).  resize_fn(img, new_h, new_w): the cv::resize stand-in
    (default: the integer bilinear of pre_post.py)."""
    h, w = frame_bgr_u8.shape[:2]
    scale, new_w, new_h, pad_x, pad_y = letterbox_geometry(h, w, net_h, net_w)
    canvas = np.full((net_h, net_w, 3), 114, np.uint8)
    canvas[pad_y:pad_y + new_h, pad_x:pad_x + new_w] = (resize_fn or pre_post.resize_bilinear_u8)(frame_bgr_u8, new_h, new_w)
    planes = canvas.astype(F) * F(1.0 / 255.0)
    return np.ascontiguousarray(planes[:, :, ::-1].transpose(2, 0, 1)), (scale, pad_x, pad_y)


def iou(a, b):
    """computeIoU, onnxruntime_engine.cpp:239-255; a, b = (x1, y1, x2, y2) fp32."""
    iw = max(F(0), F(min(a[2], b[2]) - max(a[0], b[0])))
    ih = max(F(0), F(min(a[3], b[3]) - max(a[1], b[1])))
    inter = F(iw * ih)
    union = F(F(F(F(a[2] - a[0]) * F(a[3] - a[1])) + F(F(b[2] - b[0]) * F(b[3] - b[1]))) - inter)
    return F(inter / union) if union > 0 else F(0)


def postprocess(raw, conf_thresh, iou_thresh, scale, pad_x, pad_y, orig_w, orig_h):
    """raw: [num_attrs][num_boxes] fp32 (4 box rows cx, cy, w, h in letterbox pixels, then the class scores).
    Returns the kept detections as an [n][6] fp32 array (x1, y1, x2, y2, confidence, class_id) in the reference's output order."""
    raw = np.asarray(raw, F)
    num_attrs, num_boxes = raw.shape
    scores = raw[4:]
    best = np.zeros(num_boxes, F)
    cls = np.full(num_boxes, -1, np.int32)
    for c in range(num_attrs - 4):                                   # strict '>' from 0: the first maximum wins, NaN never does
        take = scores[c] > best
        best = np.where(take, scores[c], best)
        cls = np.where(take, c, cls)
    keep = np.nonzero(~(best < F(conf_thresh)))[0]
    cx, cy, w, h = raw[0, keep], raw[1, keep], raw[2, keep], raw[3, keep]
    two, px, py, sc = F(2), F(pad_x), F(pad_y), F(scale)

    def back(v, pad, hi):
        return np.maximum(F(0), np.minimum(F(hi), (v - pad) / sc)).astype(F)

    x1, y1 = back((cx - w / two).astype(F), px, orig_w), back((cy - h / two).astype(F), py, orig_h)
    x2, y2 = back((cx + w / two).astype(F), px, orig_w), back((cy + h / two).astype(F), py, orig_h)
    order = np.argsort(-best[keep].astype(np.float64), kind="stable")   # descending confidence, ties in box order
    det = np.stack([x1, y1, x2, y2, best[keep], cls[keep].astype(F)], axis=1)[order]
    suppressed = np.zeros(len(det), bool)
    out = []
    thr = F(iou_thresh)
    for i in range(len(det)):
        if suppressed[i]:
            continue
        out.append(det[i])
        for j in range(i + 1, len(det)):
            if not suppressed[j] and det[i, 5] == det[j, 5] and iou(det[i, :4], det[j, :4]) > thr:
                suppressed[j] = True
    return np.array(out, F).reshape(-1, 6)
