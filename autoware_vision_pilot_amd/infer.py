"""Python operator API: same class names, call signatures, return types and errors as the reference's
``Models/inference/*_infer.py`` wrappers, executed by libvp_hip on an MI355X (no torch, no CPU fallback).

    SceneSegNetworkInfer(checkpoint_path).inference(PIL 640x320 RGB) -> int64  [320,640]   scene_seg_infer.py:11-57
    Scene3DNetworkInfer(checkpoint_path).inference(image)            -> fp32   [320,640,1] scene_3d_infer.py:12-58
    DomainSegNetworkInfer(checkpoint_path).inference(image)          -> fp32   [320,640,1] (0/1) domain_seg_infer.py:12-60
    EgoLanesNetworkInfer(checkpoint_path).inference(image)           -> fp32   [3,80,160]  ego_lanes_infer.py:8-62

``checkpoint_path`` is a VPW1 blob (``weights.export_checkpoint`` converts the reference ``.pth``), or blob bytes.
``image`` is anything with ``.size == (640, 320)`` convertible by ``numpy.asarray`` to HxWx3 uint8 RGB (a PIL image),
exactly what the reference's callers pass after ``PIL.Image.resize((640, 320))``.
"""
import numpy as np

from . import lib as _lib

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def image_loader(image):
    """ToTensor + Normalize (scene_seg_infer.py:15-20): HxWx3 uint8 RGB -> 1x3xHxW fp32."""
    a = np.asarray(image)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError("image must be HxWx3 uint8 RGB")
    x = (a.astype(np.float32) / np.float32(255.0) - _MEAN) / _STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))[None]


class _ModelEngine(_lib.Engine):
    """``self.model`` of the wrappers below: an Engine whose resize mode can be BORROWED by ``inference_resized``.  The borrowed mode
    (Pillow's antialiased bicubic) stays set across consecutive ``inference_resized`` calls -- no per-frame rebuild of the tap tables and
    no graph recapture -- and the caller's own mode comes back the moment the caller looks at it or feeds a frame
    directly (``model.resize_mode()``, ``model.infer(frame)``, ``model.upload_frame`` ...)."""
    _lent = None    # the caller's mode while the PIL one is borrowed

    def _give_back(self):
        if self._lent is not None:
            prev, self._lent = self._lent, None
            super().set_resize_mode(prev)

    def borrow_resize_mode(self, mode):
        cur = super().resize_mode()
        if cur != mode:
            if self._lent is None:
                self._lent = cur
            super().set_resize_mode(mode)

    def infer_borrowed(self, frame_u8):
        super().infer(frame_u8)

    def resize_mode(self):
        self._give_back()
        return super().resize_mode()

    def set_resize_mode(self, mode):
        self._lent = None
        super().set_resize_mode(mode)

    def infer(self, frame_u8):
        self._give_back()
        super().infer(frame_u8)

    def infer_multi(self, heads, frame_u8):
        self._give_back()
        super().infer_multi(heads, frame_u8)

    def upload_frame(self, frame_u8, index=None):
        self._give_back()
        super().upload_frame(frame_u8, index)

    # every other frame-consuming entry of the Engine gives the mode back as well: a caller that mixes them with
    # inference_resized() always runs on its OWN resize mode
    def infer_pair(self, prev_u8, cur_u8):
        self._give_back()
        return super().infer_pair(prev_u8, cur_u8)

    def enqueue(self):
        self._give_back()
        super().enqueue()

    def enqueue_multi(self, heads):
        self._give_back()
        super().enqueue_multi(heads)

    def profile_layers(self, iters=10):
        self._give_back()
        return super().profile_layers(iters)


class _NetworkInfer:
    _kind = None

    def __init__(self, checkpoint_path="", precision="fp16", gpu_id=0):
        if checkpoint_path is None or len(checkpoint_path) == 0:
            raise ValueError("No path to checkpiont file provided in class initialization")
        self.device = f"hip:{gpu_id}"
        self.model = _ModelEngine(self._kind, checkpoint_path, precision=precision, gpu_id=gpu_id)
        self.model.set_input_format(_lib.VP_RGB8, _lib.VP_PLANES_RGB)

    def _forward(self, image, check_size=True):
        if check_size:
            width, height = image.size
            if width != 640 or height != 320:
                raise ValueError("Incorrect input size - input image must have height of 320px and width of 640px")
        self.model.infer_tensor(image_loader(image))
        return self.model.logits()  # CxHxW fp32

    def _post(self):
        raise NotImplementedError

    def inference_resized(self, frame):
        """The visualisation scripts' call pair ``image_pil = image_pil.resize((640, 320))`` + ``inference(image_pil)``
        (Models/visualizations/Scene3D/video_visualization.py:87-88, DomainSeg/video_visualization.py:112-113) as ONE device pass on a
        frame of any size (HxWx3 uint8 RGB, e.g. a PIL image): Pillow's default-filter resize (BICUBIC, antialiased) is done by the
        engine, bit-exact against Pillow (VP_RESIZE_PIL_BICUBIC).  Same return value as ``inference``."""
        a = np.asarray(frame)
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            raise ValueError("frame must be HxWx3 uint8 RGB")
        # the PIL mode is BORROWED (see _ModelEngine): it stays set while inference_resized calls follow one another, and a later
        # self.model.infer(frame) / resize_mode() by the caller sees the mode the caller chose (VP_RESIZE_CV_LINEAR by default)
        self.model.borrow_resize_mode(_lib.VP_RESIZE_PIL_BICUBIC)
        self.model.infer_borrowed(a)
        return self._post()


class SceneSegNetworkInfer(_NetworkInfer):
    _kind = "sceneseg"

    def __init__(self, checkpoint_path="", precision="fp16", gpu_id=0):
        super().__init__(checkpoint_path, precision, gpu_id)
        self.model.set_decode_mode(_lib.VP_DECODE_CLASS_INDEX)

    def _post(self):
        return self.model.mask().astype(np.int64)  # argmax index map, first max wins (scene_seg_infer.py:52-55)

    def inference(self, image):
        self._forward(image)
        return self._post()


class Scene3DNetworkInfer(_NetworkInfer):
    _kind = "scene3d"

    def _post(self):
        return np.ascontiguousarray(self.model.logits().transpose(1, 2, 0))

    def inference(self, image):
        self._forward(image)
        return self._post()


class DomainSegNetworkInfer(_NetworkInfer):
    _kind = "domainseg"

    def _post(self):
        return (self.model.mask() > 0).astype(np.float32)[..., None]  # 0/1 floats (domain_seg_infer.py:54-58)

    def inference(self, image):
        self._forward(image)
        return self._post()


class EgoLanesNetworkInfer(_NetworkInfer):
    _kind = "egolanes"

    def _post(self):
        return self.model.logits()

    def inference(self, image):
        return self._forward(image, check_size=False)  # ego_lanes_infer.py:50-62 has no size check


class AutoDriveInfer:
    """Models/model_components/autodrive/autodrive_network.py:15-36 behind the engine: ``forward(prev, curr)`` on two
    HxWx3 uint8 frames (any size; the engine does the visualisation script's preprocess -- PIL's antialiased
    Image.resize((1024, 512), Image.BILINEAR), bit-exact against Pillow, RGB planes, to_tensor + ImageNet normalisation,
    video_visualization.py:29-33) -> (d_norm, curvature, flag_logit);
    ``step(frame)`` is the streaming form (pairs each frame with the previous one).  ``frames_are_bgr`` matches
    OpenCV captures.  ``weights_fp8`` selects the per-channel e4m3 weight format of BASELINE configs[4]."""

    D_MAX_M = 150.0

    def __init__(self, checkpoint_path="", precision="fp16", gpu_id=0, weights_fp8=False, frames_are_bgr=True):
        if checkpoint_path is None or len(checkpoint_path) == 0:
            raise ValueError("No path to checkpiont file provided in class initialization")
        self.model = _lib.Engine("autodrive", checkpoint_path, precision=precision, gpu_id=gpu_id, weights_fp8=weights_fp8)
        self.model.set_input_format(_lib.VP_BGR8 if frames_are_bgr else _lib.VP_RGB8, _lib.VP_PLANES_RGB)

    def forward(self, image_prev, image_curr):
        self.model.infer_pair(np.asarray(image_prev), np.asarray(image_curr))
        d, c, f = self.model.logits().reshape(3)
        return float(d), float(c), float(f)

    def step(self, image):
        self.model.infer(np.asarray(image))
        d, c, f = self.model.logits().reshape(3)
        return float(d), float(c), float(f)

    @staticmethod
    def to_distance_meters(d_norm):
        """autodrive_head.py:89-92."""
        return AutoDriveInfer.D_MAX_M * (1.0 - d_norm)

