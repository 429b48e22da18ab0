"""Shared inputs of the AutoSpeed pre / post-processing tests (CPU-emulated and GPU): synthetic frames and detector output tensors with
the cases the reference's code distinguishes -- ties in confidence, boxes of different classes that overlap, boxes outside the frame,
scores of zero and below, a threshold of zero, nothing above the threshold, more detections than the caller's buffer."""
import numpy as np

from oracle import autospeed


def frame(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = ((x * 3 + y * 5 + seed * 17) % 256).astype(np.int32)
    img = np.stack([(base + 40 * c) % 256 for c in range(3)], axis=2) + rng.integers(-20, 21, size=(h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def raw_tensor(num_boxes, num_classes, seed, net=640, clusters=40, p_obj=0.12):
    """[4 + classes][boxes] fp32 like a YOLO head's: most boxes near zero score, clusters of overlapping boxes around a few objects."""
    rng = np.random.default_rng(seed)
    raw = np.zeros((4 + num_classes, num_boxes), np.float32)
    centres = rng.uniform(-20, net + 20, size=(clusters, 2))
    sizes = rng.uniform(10, 220, size=(clusters, 2))
    owner = rng.integers(0, clusters, size=num_boxes)
    raw[0] = centres[owner, 0] + rng.normal(0, 6, num_boxes)
    raw[1] = centres[owner, 1] + rng.normal(0, 6, num_boxes)
    raw[2] = np.abs(sizes[owner, 0] * rng.uniform(0.8, 1.25, num_boxes))
    raw[3] = np.abs(sizes[owner, 1] * rng.uniform(0.8, 1.25, num_boxes))
    raw[4:] = rng.uniform(-0.05, 0.08, size=(num_classes, num_boxes))
    obj = rng.random(num_boxes) < p_obj
    cls = owner % num_classes
    conf = np.round(rng.uniform(0.2, 0.99, num_boxes), 2)            # two decimals: plenty of EQUAL confidences
    raw[4 + cls[obj], np.nonzero(obj)[0]] = conf[obj]
    second = obj & (rng.random(num_boxes) < 0.3)                     # a second class scoring exactly the same: the first one wins
    raw[4 + (cls[second] + 1) % num_classes, np.nonzero(second)[0]] = conf[second]
    if num_boxes > 8:
        raw[4:, 3] = 0.0                                             # all scores zero: class -1, confidence 0
        raw[4:, 5] = -1.0
        raw[4:, 7] = np.nan                                          # NaN never wins the strict '>'
    return raw


def untie(raw, seed):
    """Every positive class score made distinct, NaNs removed: the reference's std::sort is not stable (onnxruntime_engine.cpp:252), so on such
    tensors the order of the detections is fully specified by the reference's code -- the TIE_FREE_CASES below."""
    rng = np.random.default_rng(seed)
    r = raw.copy()
    pos = r[4:] > 0.1
    r[4:][pos] += (rng.permutation(int(pos.sum())).astype(np.float32) + 1) * np.float32(1e-6)
    r[4:][np.isnan(r[4:])] = 0.0
    best = r[4:].max(axis=0)
    assert len(np.unique(best[best > 0.1])) == int((best > 0.1).sum())
    return r


# tie-free decode + NMS cases: (boxes, classes, tensor seed, untie seed), (conf threshold, IoU threshold, scale, pad_x, pad_y, orig_w, orig_h) -- 80 % of
# a YOLO head at three letterbox geometries, one class, a threshold above every score (nothing kept), an IoU threshold of 0, a tiny tensor
TIE_FREE_CASES = [((2100, 4, 300, 0), (0.25, 0.45, 0.5, 0, 140, 1280, 720)), ((2100, 8, 301, 1), (0.5, 0.3, 0.3333333432674408, 0, 140, 1920, 1080)),
                  ((700, 1, 302, 2), (0.05, 0.45, 1.0, 0, 0, 640, 640)), ((2100, 4, 303, 3), (1.5, 0.45, 0.5, 0, 140, 1280, 720)),
                  ((37, 3, 304, 4), (0.1, 0.0, 1.3141683340072632, 122, 0, 301, 487)), ((2100, 4, 305, 5), (0.2, 0.6, 0.3333333432674408, 0, 140, 1920, 1080))]
# letterbox geometry the reference's expressions give (scale = min(W/w, H/h) in fp32, truncated new size, integer-division pads): (h, w) -> (scale, pad_x, pad_y)
LETTERBOX_GEOMETRY = [((720, 1280), (0.5, 0, 140)), ((1080, 1920), (0.3333333432674408, 0, 140)), ((487, 301), (1.3141683340072632, 122, 0)),
                      ((640, 640), (1.0, 0, 0)), ((33, 900), (0.7111111283302307, 0, 308)), ((641, 1283), (0.4988308548927307, 0, 160))]


def check(det_got, n_got, raw, conf, iou, geom, orig_w, orig_h, cap=None):
    scale, pad_x, pad_y = geom
    want = autospeed.postprocess(raw, conf, iou, scale, pad_x, pad_y, orig_w, orig_h)
    assert n_got == len(want), (n_got, len(want))
    k = len(want) if cap is None else min(cap, len(want))
    assert det_got.shape == (k, 6)
    assert np.array_equal(det_got.view(np.uint32), want[:k].view(np.uint32))      # bit for bit, in the reference's order
    return len(want)
