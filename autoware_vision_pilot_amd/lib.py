"""ctypes binding of libvp_hip.so (C ABI: include/vp_hip.h).

There is NO CPU fallback: if the shared library is missing or no HIP device is visible every entry point
raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C autoware_vision_pilot_amd/csrc``.

Note for mixed processes (tests, bench): PyTorch-ROCm bundles its own ``libamdhip64.so``; import ``torch``
BEFORE this module in any process that uses both, so that a single HIP runtime is loaded.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvp_hip.so")

VP_SCENESEG, VP_SCENE3D, VP_DOMAINSEG, VP_EGOLANES, VP_AUTODRIVE = 0, 1, 2, 3, 4
VP_WEIGHTS_FP8 = 16
VP_PLAN_LATENCY = 32  # creation flag: this engine's kernel plan targets one network on one camera, one frame at a time (vp_hip.h)
VP_FP16, VP_FP16X3 = 0, 1
VP_BGR8, VP_RGB8 = 0, 1
VP_PLANES_BGR, VP_PLANES_RGB = 0, 1
VP_DECODE_SEG_MASK, VP_DECODE_LANE_LABEL, VP_DECODE_CLASS_INDEX = 0, 1, 2
KINDS = {"sceneseg": VP_SCENESEG, "scene3d": VP_SCENE3D, "domainseg": VP_DOMAINSEG, "egolanes": VP_EGOLANES, "autodrive": VP_AUTODRIVE}
PRECISIONS = {"fp16": VP_FP16, "fp16x3": VP_FP16X3, "fp32": VP_FP16X3}

# every symbol include/vp_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_SIGS = {
    "vp_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_create_from_memory": (C.c_int, [C.POINTER(_P), C.c_int, _P, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_create_shared": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_create_shared_from_memory": (C.c_int, [C.POINTER(_P), _P, C.c_int, _P, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_create_batched": (C.c_int, [C.POINTER(_P), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_create_batched_from_memory": (C.c_int, [C.POINTER(_P), C.c_int, _P, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_create_shared_frame": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_create_shared_frame_from_memory": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int, _P, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_frames": (C.c_int, [_P]),
    "vp_upload_frame_n": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "vp_shared_level": (C.c_int, [_P]),
    "vp_infer_shared": (C.c_int, [_P]),
    "vp_destroy": (None, [_P]),
    "vp_last_error": (C.c_char_p, [_P]),
    "vp_set_input_format": (C.c_int, [_P, C.c_int, C.c_int]),
    "vp_set_decode_mode": (C.c_int, [_P, C.c_int]),
    "vp_get_decode_mode": (C.c_int, [_P]),
    "vp_set_resize_mode": (C.c_int, [_P, C.c_int]),
    "vp_get_resize_mode": (C.c_int, [_P]),
    "vp_resample_coeffs": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]),
    "vp_gpu_id": (C.c_int, [_P]),
    "vp_device_count": (C.c_int, []),
    "vp_host_logits_current": (C.c_int, [_P]),
    "vp_input_hw": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_infer": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "vp_infer_tensor": (C.c_int, [_P, _P]),
    "vp_infer_pair": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "vp_logits": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int64)]),
    "vp_mask_u8": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_decode_logits_host": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vp_mask_resized_u8": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "vp_depth_resized_f32": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "vp_visualize_mask_bgr8": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int]),
    "vp_frame_hw": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_set_outputs": (C.c_int, [_P, C.c_int]),
    "vp_set_pinned_staging": (C.c_int, [_P, C.c_int]),
    "vp_register_frames": (C.c_int, [C.c_void_p, C.c_size_t]),
    "vp_unregister_frames": (C.c_int, [C.c_void_p]),
    "vp_output_shape": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "vp_set_finite_check": (C.c_int, [_P, C.c_int]),
    "vp_infer_multi": (C.c_int, [_P, C.POINTER(_P), C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "vp_enqueue_multi": (C.c_int, [_P, C.POINTER(_P), C.c_int]),
    "vp_set_multi_fork": (C.c_int, [_P, C.c_int]),
    "vp_visualize_depth_bgr8": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "vp_input_tensor": (C.c_int, [_P, _P]),
    "vp_upload_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "vp_enqueue": (C.c_int, [_P]),
    "vp_sync": (C.c_int, [_P]),
    "vp_fetch_outputs": (C.c_int, [_P]),
    "vp_device_outputs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "vp_use_graph": (C.c_int, [_P, C.c_int]),
    "vp_timer_begin": (C.c_int, [_P]),
    "vp_timer_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "vp_layer_count": (C.c_int, [_P]),
    "vp_layer_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vp_layer_kernel": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p)]),
    "vp_layer_launch": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p)]),
    "vp_autosteer_angle": (C.c_float, [_P, C.c_int]),
    "vp_copy_outputs_device": (C.c_int, [_P, _P, _P]),
    "vp_profile_layers": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "vp_layer_flops_executed": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "vp_tensor_count": (C.c_int, [_P]),
    "vp_tensor_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_tensor_read": (C.c_int, [_P, C.c_int, _P]),
    "vp_op_conv2d": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P,
                               C.c_int, C.c_int, C.c_int, _P, C.c_char_p, C.c_size_t]),
    "vp_compose_upconv": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_char_p, C.c_size_t]),
    "vp_op_upconv": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P,
                               C.c_char_p, C.c_size_t]),
    "vp_detect_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "vp_detect_destroy": (None, [_P]),
    "vp_detect_last_error": (C.c_char_p, [_P]),
    "vp_detect_preprocess": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "vp_detect_input_device": (C.c_int, [_P, C.POINTER(_P)]),
    "vp_detect_letterbox": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vp_detect_set_letterbox": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vp_detect_postprocess": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, C.c_int, C.POINTER(C.c_int)]),
    "vp_version": (C.c_char_p, []),
    "vp_set_lane_ring": (C.c_int, [_P, C.c_int]),
    "vp_lane_ring_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int)]),
    "vp_lane_ring_fetch": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    "vp_set_norm_form": (C.c_int, [_P, C.c_int]),
    "vp_get_norm_form": (C.c_int, [_P]),
    "vp_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "vp_get_option": (C.c_char_p, [C.c_char_p]),
    "vp_clear_options": (None, []),
    "vp_plan_hash": (C.c_ulonglong, [_P]),
    "vp_weight_bytes": (C.c_int, [_P, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "vp_split_weight_rows": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    "vp_fp8_encode_rows": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "vp_convert_onnx": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
}
# multi-camera exchange (csrc/vp_comm.cpp): part of libvp_hip.so, absent from the CPU-emulated test build
_COMM_SIGS = {
    "vp_comm_unique_id": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "vp_comm_create": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.c_size_t]),
    "vp_comm_destroy": (None, [_P]),
    "vp_comm_last_error": (C.c_char_p, [_P]),
    "vp_comm_rank": (C.c_int, [_P]),
    "vp_comm_world": (C.c_int, [_P]),
    "vp_gather": (C.c_int, [_P, _P, C.c_int]),
    "vp_comm_device_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "vp_comm_fetch": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
}
EXPORTED_SYMBOLS = tuple(_SIGS) + tuple(_COMM_SIGS)
VP_OUT_LOGITS, VP_OUT_MASK = 1, 2
VP_NORM_TORCHVISION, VP_NORM_OPENCV = 0, 1
VP_RESIZE_CV_LINEAR, VP_RESIZE_PIL_BILINEAR, VP_RESIZE_PIL_BICUBIC = 0, 1, 2
VP_GATHER_MASK, VP_GATHER_LOGITS = 0, 1
VP_COMM_ID_BYTES = 128

_lib = None


class VpError(RuntimeError):
    pass


class VpRangeError(VpError):
    """VP_ERR_RANGE: a weight beyond the fp16 range at load, or inf / NaN in a network's output (an activation left the fp16 range)."""


def load():
    """dlopen libvp_hip.so and bind every symbol; raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VpError(f"{LIB_PATH} not built -- run __graft_entry__.build(); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the multi-camera exchange is bound where the build carries it (a single-camera host may build without the RCCL headers:
    # csrc/Makefile VP_NO_COMM=1); Comm raises on first use otherwise.  tests/test_host_cpu.py checks the shipped library has them all.
    for name, (res, args) in _COMM_SIGS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def set_option(key, value):
    """vp_set_option: a developer knob of the dispatch rules (process-wide, engines created afterwards); value None removes it.
    The library never reads the environment -- tools that want the old `VP_X=... python tool.py` spelling call options_from_env()."""
    rc = load().vp_set_option(key.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise ValueError(f"vp_set_option({key!r}): unknown key")


def get_option(key):
    v = load().vp_get_option(key.encode())
    return None if v is None else v.decode()


def clear_options():
    load().vp_clear_options()


def options_from_env(environ=None):
    """DEVELOPER TOOLS ONLY (tools/*.py call it explicitly): copy the VP_* variables the dispatch rules know from the environment into
    the library's option table.  Returns what was set."""
    environ = os.environ if environ is None else environ
    done = {}
    for k, v in environ.items():
        if k.startswith("VP_") and load().vp_set_option(k.encode(), v.encode()) == 0:
            done[k] = v
    return done


def register_frames(buf):
    """vp_register_frames: page-lock a host buffer (a C-contiguous uint8 ndarray that OUTLIVES the registration, e.g. a frame pool) so frames
    inside it go to the device by one DMA with no staging copy.  Keep the array alive and call unregister_frames(buf) before freeing it."""
    a = np.asarray(buf)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("frame pool must be C-contiguous")
    rc = load().vp_register_frames(a.ctypes.data_as(C.c_void_p), a.nbytes)
    if rc != 0:
        raise VpError(f"vp_register_frames failed (rc {rc})")
    return a


def unregister_frames(buf):
    rc = load().vp_unregister_frames(np.asarray(buf).ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise VpError(f"vp_unregister_frames failed (rc {rc})")


def version():
    return load().vp_version().decode()


def fp8_encode_rows(w):
    """vp_fp8_encode_rows (host only): fp32 [rows][per_row] -> (uint8 e4m3 codes, fp32 row scales)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    rows, per = w.shape
    codes = np.empty((rows, per), np.uint8)
    scale = np.empty(rows, np.float32)
    if load().vp_fp8_encode_rows(_ptr(w), rows, per, _ptr(codes), _ptr(scale)) != 0:
        raise ValueError("vp_fp8_encode_rows failed")
    return codes, scale


def split_weight_rows(w):
    """vp_split_weight_rows (host only): fp32 [rows][per_row] -> (hi fp16, lo fp16, post_scale fp32 [rows])."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    rows, per = w.shape
    hi = np.empty((rows, per), np.float16)
    lo = np.empty((rows, per), np.float16)
    post = np.empty(rows, np.float32)
    rc = load().vp_split_weight_rows(_ptr(w), rows, per, _ptr(hi), _ptr(lo), _ptr(post))
    if rc != 0:
        raise (VpRangeError if rc == -5 else ValueError)(f"vp_split_weight_rows failed ({rc})")
    return hi, lo, post


def convert_onnx(onnx_path, vpw_path):
    """ONNX file (reference exporter settings) -> VPW1 blob file with the library's native reader (csrc/onnx_reader.cpp;
    host only, no GPU needed).  `Engine(kind, "model.onnx")` does the same conversion in memory."""
    err = C.create_string_buffer(512)
    rc = load().vp_convert_onnx(os.fsencode(onnx_path), os.fsencode(vpw_path), err, len(err))
    if rc != 0:
        raise (ValueError if rc == -1 else VpError)(f"vp_convert_onnx failed ({rc}): {err.value.decode(errors='replace')}")
    return vpw_path


class Engine:
    """Thin RAII wrapper over a vp_engine handle."""

    def __init__(self, kind, weights, precision="fp16", gpu_id=0, base=None, weights_fp8=False, frames=1, frame_index=None, plan_latency=False):
        """base: another Engine -> shared-prefix engine (vp_create_shared): reuses the base engine's backbone (and
        context + neck when their parameters are identical) on the frame the base last processed.
        frames > 1: batched encoder (vp_create_batched) -- preprocess + backbone of `frames` cameras per pass, no head;
        heads are Engine(kind, weights, base=encoder, frame_index=f) (vp_create_shared_frame)."""
        lib = load()
        self._lib = lib
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        k = KINDS[kind] if isinstance(kind, str) else int(kind)
        pr = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        if weights_fp8:
            pr |= VP_WEIGHTS_FP8
        if plan_latency:
            pr |= VP_PLAN_LATENCY
        self._base = base  # keeps the base engine alive as long as this one
        path = None if isinstance(weights, (bytes, bytearray, memoryview, np.ndarray)) else (os.fsencode(weights) if weights else b"")
        if frames != 1 or frame_index is not None:
            buf = None if path is not None else (np.frombuffer(weights, dtype=np.uint8) if not isinstance(weights, np.ndarray) else weights)
            if base is None:
                rc = (lib.vp_create_batched(C.byref(self._h), k, path, pr, gpu_id, frames, err, len(err)) if path is not None else
                      lib.vp_create_batched_from_memory(C.byref(self._h), k, _ptr(buf), buf.nbytes, pr, gpu_id, frames, err, len(err)))
            else:
                fi = int(frame_index or 0)
                rc = (lib.vp_create_shared_frame(C.byref(self._h), base._h, fi, k, path, pr, gpu_id, err, len(err)) if path is not None else
                      lib.vp_create_shared_frame_from_memory(C.byref(self._h), base._h, fi, k, _ptr(buf), buf.nbytes, pr, gpu_id, err, len(err)))
        elif isinstance(weights, (bytes, bytearray, memoryview, np.ndarray)):
            buf = np.frombuffer(weights, dtype=np.uint8) if not isinstance(weights, np.ndarray) else weights
            if base is not None:
                rc = lib.vp_create_shared_from_memory(C.byref(self._h), base._h, k, _ptr(buf), buf.nbytes, pr, gpu_id, err, len(err))
            else:
                rc = lib.vp_create_from_memory(C.byref(self._h), k, _ptr(buf), buf.nbytes, pr, gpu_id, err, len(err))
        elif base is not None:
            rc = lib.vp_create_shared(C.byref(self._h), base._h, k, os.fsencode(weights) if weights else b"", pr, gpu_id, err, len(err))
        else:
            rc = lib.vp_create(C.byref(self._h), k, os.fsencode(weights) if weights else b"", pr, gpu_id, err, len(err))
        if rc != 0:
            self._h = C.c_void_p()
            msg = err.value.decode(errors="replace")
            raise (ValueError if rc == -1 else VpRangeError if rc == -5 else VpError)(f"vp_create failed ({rc}): {msg}")
        self.kind = kind

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._h = None
            self._lib.vp_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: globals may already be gone
            pass

    def _ck(self, rc):
        if rc < 0:
            msg = self._lib.vp_last_error(self._h).decode(errors="replace")
            raise (ValueError if rc == -1 else VpRangeError if rc == -5 else VpError)(f"libvp_hip error {rc}: {msg}")
        return rc

    def set_finite_check(self, on):
        """inf / NaN probe on the logits at the end of every pass (default on): VpRangeError from the synchronising call."""
        self._ck(self._lib.vp_set_finite_check(self._h, int(bool(on))))

    # ---- configuration
    def set_input_format(self, pixel_format, plane_order):
        self._ck(self._lib.vp_set_input_format(self._h, pixel_format, plane_order))

    def set_decode_mode(self, mode):
        self._ck(self._lib.vp_set_decode_mode(self._h, mode))

    def set_resize_mode(self, mode):
        """VP_RESIZE_CV_LINEAR (the C++ nodes' cv::resize model), VP_RESIZE_PIL_BILINEAR / _BICUBIC (the Python scripts' Image.resize)."""
        self._ck(self._lib.vp_set_resize_mode(self._h, mode))

    def resize_mode(self):
        return self._lib.vp_get_resize_mode(self._h)

    def set_norm_form(self, form):
        """VP_NORM_TORCHVISION: q / 255 (to_tensor; default), VP_NORM_OPENCV: q * fl(1/255) (cv::Mat::convertTo, the C++ front-ends)."""
        self._ck(self._lib.vp_set_norm_form(self._h, form))

    def norm_form(self):
        return self._lib.vp_get_norm_form(self._h)

    def plan_hash(self):
        return int(self._lib.vp_plan_hash(self._h))

    def set_lane_ring(self, on=True):
        """AutoSteer hand-over on an EgoLanes engine: keep [t-1 | t] of the raw logits (fp32 6 x 80 x 160) on the device (vp_set_lane_ring)."""
        self._ck(self._lib.vp_set_lane_ring(self._h, 1 if on else 0))

    def lane_ring(self):
        """(6 x 80 x 160 fp32 host copy of the ring, frames_valid)  -- frames_valid < 2: the reference skips AutoSteer (main.cpp:521)."""
        buf = np.empty((6, 80, 160), np.float32)
        n = C.c_int()
        self._ck(self._lib.vp_lane_ring_fetch(self._h, _ptr(buf), C.byref(n)))
        return buf, n.value

    def weight_bytes(self):
        """Device bytes of the weight tensors by storage class: {'fp8': e4m3 codes, 'fp16': fp16 planes, 'fp32': fp32 rows} (vp_weight_bytes)."""
        a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        self._ck(self._lib.vp_weight_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"fp8": a.value, "fp16": b.value, "fp32": c.value}

    def input_hw(self):
        h, w = C.c_int(), C.c_int()
        self._ck(self._lib.vp_input_hw(self._h, C.byref(h), C.byref(w)))
        return h.value, w.value

    # ---- synchronous path
    def infer(self, frame_u8):
        f = np.ascontiguousarray(frame_u8, dtype=np.uint8)
        if f.ndim != 3 or f.shape[2] != 3:
            raise ValueError("frame must be HxWx3 uint8")
        self._ck(self._lib.vp_infer(self._h, _ptr(f), f.shape[0], f.shape[1], f.strides[0]))

    def infer_pair(self, prev_u8, curr_u8):
        """AutoDrive.forward(image_prev, image_curr): both HxWx3 uint8 frames of the same geometry."""
        a = np.ascontiguousarray(prev_u8, dtype=np.uint8)
        b = np.ascontiguousarray(curr_u8, dtype=np.uint8)
        if a.shape != b.shape or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("frames must be two HxWx3 uint8 arrays of the same shape")
        self._ck(self._lib.vp_infer_pair(self._h, _ptr(a), _ptr(b), a.shape[0], a.shape[1], a.strides[0]))

    def infer_shared(self):
        """Run this shared-prefix engine's own layers on the frame its base engine processed last."""
        self._ck(self._lib.vp_infer_shared(self._h))

    def shared_level(self):
        return self._lib.vp_shared_level(self._h)

    def infer_tensor(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        h, w = self.input_hw()
        if x.size != 3 * h * w:
            raise ValueError(f"tensor must be 1x3x{h}x{w}")
        self._ck(self._lib.vp_infer_tensor(self._h, _ptr(x)))

    def logits(self):
        p = C.POINTER(C.c_float)()
        shape = (C.c_int64 * 4)()
        self._ck(self._lib.vp_logits(self._h, C.byref(p), shape))
        n = int(shape[1] * shape[2] * shape[3])
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(tuple(shape)[1:]).copy()

    def mask(self):
        p = C.POINTER(C.c_uint8)()
        h, w = C.c_int(), C.c_int()
        self._ck(self._lib.vp_mask_u8(self._h, C.byref(p), C.byref(h), C.byref(w)))
        return np.ctypeslib.as_array(p, shape=(h.value * w.value,)).reshape(h.value, w.value).copy()

    def mask_resized(self, h, w):
        out = np.empty((h, w), dtype=np.uint8)
        self._ck(self._lib.vp_mask_resized_u8(self._h, _ptr(out), h, w))
        return out

    def depth_resized(self, h, w):
        out = np.empty((h, w), dtype=np.float32)
        self._ck(self._lib.vp_depth_resized_f32(self._h, _ptr(out), h, w))
        return out

    def visualize_depth(self, h, w):
        """Colourised depth (depth_visualization_engine.cpp): plane 0 of the logits at h x w, min-max -> u8 -> VIRIDIS; BGR8."""
        out = np.empty((h, w, 3), dtype=np.uint8)
        self._ck(self._lib.vp_visualize_depth_bgr8(self._h, _ptr(out), h, w))
        return out

    def visualize_mask(self, viz_type, frame_hw):
        """Blended BGR visualisation (masks_visualization_engine.cpp) of the last inference at the last frame's size;
        ValueError if ``frame_hw`` is not the geometry of the frame that was inferred last."""
        out = np.empty((frame_hw[0], frame_hw[1], 3), dtype=np.uint8)
        self._ck(self._lib.vp_visualize_mask_bgr8(self._h, int(viz_type), _ptr(out), int(frame_hw[0]), int(frame_hw[1])))
        return out

    def frame_hw(self):
        h, w = C.c_int(), C.c_int()
        self._ck(self._lib.vp_frame_hw(self._h, C.byref(h), C.byref(w)))
        return h.value, w.value

    def set_outputs(self, logits=True, mask=True):
        """Which outputs infer() / infer_shared() / infer_multi() copy to the host; the rest is fetched on first use."""
        self._ck(self._lib.vp_set_outputs(self._h, (VP_OUT_LOGITS if logits else 0) | (VP_OUT_MASK if mask else 0)))

    def host_logits_current(self):
        """True when the pinned host copy of the logits belongs to the last pass (vp_host_logits_current)."""
        return bool(self._lib.vp_host_logits_current(self._h))

    def output_shape(self):
        """(1, C, H, W) of the output tensor without fetching it (vp_output_shape)."""
        sh = (C.c_int64 * 4)()
        self._ck(self._lib.vp_output_shape(self._h, sh))
        return tuple(int(v) for v in sh)

    def set_pinned_staging(self, on):
        self._ck(self._lib.vp_set_pinned_staging(self._h, int(bool(on))))

    def set_multi_fork(self, on):
        """Latency mode (default): enqueue_multi / infer_multi fork the backbone-only heads; off for several cameras in flight."""
        self._ck(self._lib.vp_set_multi_fork(self._h, 1 if on else 0))

    def enqueue_multi(self, heads):
        """Asynchronous: this base engine and its shared-prefix ``heads`` as one graph launch (level-1 heads forked behind the
        backbone, overlapping this engine's own decoder); same results and stream order as enqueue() on each."""
        arr = (C.c_void_p * max(1, len(heads)))(*[h._h for h in heads])
        self._ck(self._lib.vp_enqueue_multi(self._h, arr, len(heads)))

    def infer_multi(self, heads, frame_u8):
        """One frame through this base engine and its shared-prefix ``heads``: one H2D, one host synchronisation."""
        f = np.ascontiguousarray(frame_u8, dtype=np.uint8)
        if f.ndim != 3 or f.shape[2] != 3:
            raise ValueError("frame must be HxWx3 uint8")
        arr = (C.c_void_p * max(1, len(heads)))(*[h._h for h in heads])
        self._ck(self._lib.vp_infer_multi(self._h, arr, len(heads), _ptr(f), f.shape[0], f.shape[1], f.strides[0]))

    def input_tensor(self):
        h, w = self.input_hw()
        out = np.empty((1, 3, h, w), dtype=np.float32)
        self._ck(self._lib.vp_input_tensor(self._h, _ptr(out)))
        return out

    # ---- device-resident path
    def upload_frame(self, frame_u8, index=None):
        """index: slot of a batched encoder (vp_upload_frame_n); None = the single-frame call."""
        f = np.ascontiguousarray(frame_u8, dtype=np.uint8)
        if index is None:
            self._keep = f
            self._ck(self._lib.vp_upload_frame(self._h, _ptr(f), f.shape[0], f.shape[1], f.strides[0]))
        else:
            self._keep_n = getattr(self, "_keep_n", {})
            self._keep_n[index] = f
            self._ck(self._lib.vp_upload_frame_n(self._h, int(index), _ptr(f), f.shape[0], f.shape[1], f.strides[0]))

    def frames(self):
        return self._lib.vp_frames(self._h)

    def enqueue(self):
        self._ck(self._lib.vp_enqueue(self._h))

    def sync(self):
        self._ck(self._lib.vp_sync(self._h))

    def fetch_outputs(self):
        self._ck(self._lib.vp_fetch_outputs(self._h))

    def device_outputs(self):
        a, b = C.c_void_p(), C.c_void_p()
        self._ck(self._lib.vp_device_outputs(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def use_graph(self, on):
        self._ck(self._lib.vp_use_graph(self._h, int(bool(on))))

    def timer_begin(self):
        self._ck(self._lib.vp_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float()
        self._ck(self._lib.vp_timer_end(self._h, C.byref(ms)))
        return ms.value

    # ---- introspection
    def layers(self):
        out = []
        for i in range(self._ck(self._lib.vp_layer_count(self._h))):
            name, fl, by = C.c_char_p(), C.c_double(), C.c_double()
            self._ck(self._lib.vp_layer_info(self._h, i, C.byref(name), C.byref(fl), C.byref(by)))
            out.append((name.value.decode(), fl.value, by.value))
        return out

    def layer_flops_executed(self):
        """matrix work each launch EXECUTES (differs from layers()' reference-formulation count for the composed up-sampling stages)"""
        out = []
        for i in range(self._ck(self._lib.vp_layer_count(self._h))):
            fl = C.c_double()
            self._ck(self._lib.vp_layer_flops_executed(self._h, i, C.byref(fl)))
            out.append(fl.value)
        return out

    def layer_kernels(self):
        out = []
        for i in range(self._ck(self._lib.vp_layer_count(self._h))):
            k = C.c_char_p()
            self._ck(self._lib.vp_layer_kernel(self._h, i, C.byref(k)))
            out.append(k.value.decode())
        return out

    def layer_launches(self):
        """launch geometry beyond the kernel tag ("nsplit=4", "groups=32", ""), per launch"""
        out = []
        for i in range(self._ck(self._lib.vp_layer_count(self._h))):
            k = C.c_char_p()
            self._ck(self._lib.vp_layer_launch(self._h, i, C.byref(k)))
            out.append(k.value.decode())
        return out

    def copy_outputs_device(self, logits_ptr=None, mask_ptr=None):
        self._ck(self._lib.vp_copy_outputs_device(self._h, logits_ptr, mask_ptr))

    def profile_layers(self, iters=10):
        n = self._ck(self._lib.vp_layer_count(self._h))
        ms = np.zeros(n, dtype=np.float32)
        self._ck(self._lib.vp_profile_layers(self._h, iters, _ptr(ms), n))
        return ms

    def tensors(self):
        out = []
        for i in range(self._ck(self._lib.vp_tensor_count(self._h))):
            name, c, h, w = C.c_char_p(), C.c_int(), C.c_int(), C.c_int()
            self._ck(self._lib.vp_tensor_info(self._h, i, C.byref(name), C.byref(c), C.byref(h), C.byref(w)))
            out.append((name.value.decode(), c.value, h.value, w.value))
        return out

    def tensor_read(self, i):
        _, c, h, w = self.tensors()[i]
        out = np.empty((c, h, w), dtype=np.float32)
        self._ck(self._lib.vp_tensor_read(self._h, i, _ptr(out)))
        return out


def decode_logits_host(logits_chw, decode_mode=VP_DECODE_SEG_MASK, gpu_id=0):
    """createMaskFromTensorHIP twin (vp_decode_logits_host): host CxHxW fp32 logits -> HxW u8 mask."""
    x = np.ascontiguousarray(logits_chw, dtype=np.float32)
    c, h, w = x.shape
    out = np.empty((h, w), dtype=np.uint8)
    rc = load().vp_decode_logits_host(gpu_id, _ptr(x), c, h, w, int(decode_mode), _ptr(out))
    if rc != 0:
        raise (ValueError if rc == -1 else VpError)(f"vp_decode_logits_host failed ({rc})")
    return out


def _comm_lib():
    lib = load()
    if not hasattr(lib, "vp_comm_create"):
        raise VpError("this libvp_hip.so was built without the multi-camera exchange (csrc/Makefile VP_NO_COMM=1)")
    return lib


class Comm:
    """RCCL communicator behind the C ABI (vp_comm_*): per-frame all-gather of the per-camera result records."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(VP_COMM_ID_BYTES)
        err = C.create_string_buffer(512)
        rc = _comm_lib().vp_comm_unique_id(buf, err, len(err))
        if rc != 0:
            raise VpError(f"vp_comm_unique_id failed ({rc}): {err.value.decode(errors='replace')}")
        return buf.raw

    def __init__(self, unique_id, rank, world, gpu_id, record_bytes_max):
        self._lib = _comm_lib()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        idbuf = C.create_string_buffer(bytes(unique_id), VP_COMM_ID_BYTES)
        rc = self._lib.vp_comm_create(C.byref(self._h), idbuf, rank, world, gpu_id, record_bytes_max, err, len(err))
        if rc != 0:
            self._h = C.c_void_p()
            raise (ValueError if rc == -1 else VpError)(f"vp_comm_create failed ({rc}): {err.value.decode(errors='replace')}")
        self.rank, self.world = rank, world

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._h = None
            self._lib.vp_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            msg = self._lib.vp_comm_last_error(self._h).decode(errors="replace")
            raise (ValueError if rc == -1 else VpError)(f"libvp_hip comm error {rc}: {msg}")
        return rc

    def gather(self, engine, what=VP_GATHER_MASK):
        """Enqueue the all-gather of ``engine``'s last record on the engine's stream (no host sync)."""
        self._ck(self._lib.vp_gather(engine._h, self._h, int(what)))

    def fetch(self, engine, dtype=np.uint8):
        """[world][record] host copy of the last gather (synchronises the engine's stream)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self._lib.vp_comm_fetch(self._h, engine._h, C.byref(p), C.byref(n)))
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.world * n.value,)).copy()
        return raw.view(dtype).reshape(self.world, -1)

    def device_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self._lib.vp_comm_device_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value


def op_conv2d(x, weight, bias, ks=3, mode=0, act=0, res=None, res_mode=0, precision=VP_FP16, tile=-1, bk=-1, nsplit=-1, gpu_id=0):
    """Single-operator entry for unit parity tests (vp_op_conv2d)."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    cin, h, w = x.shape
    cout = weight.shape[1] if mode == 1 else weight.shape[0]
    if mode == 2:  # (ConvTranspose weight [cin][cout][2][2], skip 1x1 weight [cout][cs]) and the two biases, concatenated
        cout = len(bias) // 2
    oh, ow = (2 * h, 2 * w) if mode in (1, 2) else (h, w)
    out = np.empty((cout, oh, ow), dtype=np.float32)
    r = np.ascontiguousarray(res, dtype=np.float32) if res is not None else None
    err = C.create_string_buffer(512)
    rc = lib.vp_op_conv2d(gpu_id, precision, mode, _ptr(x), cin, h, w, _ptr(weight), _ptr(bias), cout, ks, act, res_mode,
                          _ptr(r) if r is not None else None, tile, bk, nsplit, _ptr(out), err, len(err))
    if rc != 0:
        raise VpError(f"vp_op_conv2d failed ({rc}): {err.value.decode(errors='replace')}")
    return out


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def compose_upconv(wt, bt, w3, b3, ws=None, bs=None, gpu_id=0):
    """vp_compose_upconv: ConvTranspose2d(k2, s2) [+ 1x1 skip link] -> Conv3x3 multiplied out (fp64).  wt [cin][cm][2][2], w3 [cout][cm][3][3],
    ws [cm][cs] or [cm][cs][1][1].  Returns (wx [4][4][cout][cin], wsk [9][cout][cs] or None, bias [9][cout])."""
    lib = load()
    wt, bt, w3, b3, ws, bs = (_f32(v) for v in (wt, bt, w3, b3, ws, bs))
    cin, cm = wt.shape[:2]
    cout = w3.shape[0]
    cs = 0 if ws is None else ws.shape[1]
    wx = np.empty((4, 4, cout, cin), dtype=np.float64)
    wsk = np.empty((9, cout, cs), dtype=np.float64) if cs else None
    bias = np.empty((9, cout), dtype=np.float64)
    err = C.create_string_buffer(512)
    rc = lib.vp_compose_upconv(gpu_id, _ptr(wt), _ptr(bt), _ptr(ws) if cs else None, _ptr(bs) if cs else None, _ptr(w3), _ptr(b3), cin, cm, cout, cs,
                               _ptr(wx), _ptr(wsk) if cs else None, _ptr(bias), err, len(err))
    if rc != 0:
        raise VpError(f"vp_compose_upconv failed ({rc}): {err.value.decode(errors='replace')}")
    return wx, wsk, bias


def op_upconv(x, wt, bt, w3, b3, skip=None, ws=None, bs=None, act=1, shape=-1, nsplit=0, gpu_id=0, precision="fp16x3"):
    """vp_op_upconv: one composed up-sampling stage through the engine's kernel (parity mode, or the fp16 engines' form of it): x [cin][h][w],
    skip [cs][2h][2w] -> [cout][2h][2w]."""
    lib = load()
    x, wt, bt, w3, b3, skip, ws, bs = (_f32(v) for v in (x, wt, bt, w3, b3, skip, ws, bs))
    cin, h, w = x.shape
    cm = wt.shape[1]
    cout = w3.shape[0]
    cs = 0 if skip is None else skip.shape[0]
    out = np.empty((cout, 2 * h, 2 * w), dtype=np.float32)
    err = C.create_string_buffer(512)
    rc = lib.vp_op_upconv(gpu_id, _ptr(x), cin, h, w, _ptr(skip) if cs else None, cs, _ptr(wt), _ptr(bt), _ptr(ws) if cs else None,
                          _ptr(bs) if cs else None, _ptr(w3), _ptr(b3), cm, cout, act, shape, nsplit, PRECISIONS[precision], _ptr(out), err, len(err))
    if rc != 0:
        raise VpError(f"vp_op_upconv failed ({rc}): {err.value.decode(errors='replace')}")
    return out


def autosteer_angle(logits):
    """vp_autosteer_angle: AutoSteerOnnxEngine::postProcess -- first arg-max over the head's logits (61 classes) minus 30 degrees (host only)."""
    a = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1)
    return float(load().vp_autosteer_angle(_ptr(a), int(a.size)))


def resample_coeffs(in_size, out_size, mode):
    """vp_resample_coeffs: (bounds [out][2], coefficients [out][ksize]) of a VP_RESIZE_PIL_* mode for one axis (host only)."""
    cap = out_size * (2 * int(np.ceil(2.0 * max(1.0, in_size / out_size))) + 1)
    b = np.zeros((out_size, 2), dtype=np.int32)
    k = np.zeros(cap, dtype=np.int32)
    ks = load().vp_resample_coeffs(in_size, out_size, mode, _ptr(b), _ptr(k), cap)
    if ks <= 0:
        raise VpError(f"vp_resample_coeffs failed ({ks})")
    return b, k[: out_size * ks].reshape(out_size, ks)


class Detector:
    """AutoSpeed detector pre / post-processing on the device (vp_detect_*): the Python twin of AutoSpeedOnnxEngine's preprocessAutoSpeed and
    postProcess (VisionPilot/middleware_recipes/common/backends/autospeed/onnxruntime_engine.cpp:71-113, :170-290).  The detector network
    itself is the caller's: `preprocess(frame)` -> its input tensor, `postprocess(raw, conf, iou)` -> the kept detections."""

    def __init__(self, net_h=640, net_w=640, max_boxes=8400, max_attrs=84, gpu_id=0):
        self._lib = load()
        self._h = _P()
        self.net_h, self.net_w = net_h, net_w
        err = C.create_string_buffer(512)
        rc = self._lib.vp_detect_create(C.byref(self._h), gpu_id, net_h, net_w, max_boxes, max_attrs, err, len(err))
        if rc != 0:
            self._h = None
            raise VpError(f"vp_detect_create failed ({rc}): {err.value.decode(errors='replace')}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vp_detect_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise VpError(f"{what} failed ({rc}): {self._lib.vp_detect_last_error(self._h).decode(errors='replace')}")

    def preprocess(self, frame_bgr_u8):
        """HxWx3 BGR uint8 -> ([3][net_h][net_w] float32 planes R, G, B in [0, 1], (scale, pad_x, pad_y))."""
        f = frame_bgr_u8
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3 or f.strides[2] != 1 or f.strides[1] != 3:
            f = np.ascontiguousarray(f, dtype=np.uint8)
        out = np.empty((3, self.net_h, self.net_w), np.float32)
        self._check(self._lib.vp_detect_preprocess(self._h, _ptr(f), f.shape[0], f.shape[1], f.strides[0], _ptr(out)), "vp_detect_preprocess")
        return out, self.letterbox()

    def letterbox(self):
        s, px, py = C.c_float(), C.c_int(), C.c_int()
        self._check(self._lib.vp_detect_letterbox(self._h, C.byref(s), C.byref(px), C.byref(py)), "vp_detect_letterbox")
        return np.float32(s.value), px.value, py.value

    def set_letterbox(self, scale, pad_x, pad_y, orig_w, orig_h):
        self._check(self._lib.vp_detect_set_letterbox(self._h, float(scale), pad_x, pad_y, orig_w, orig_h), "vp_detect_set_letterbox")

    def postprocess(self, raw, conf_thresh, iou_thresh, cap=None):
        """raw [num_attrs][num_boxes] float32 -> ([n][6] float32 rows x1, y1, x2, y2, confidence, class_id; total kept)."""
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        cap = raw.shape[1] if cap is None else cap
        out = np.zeros((max(cap, 1), 6), np.float32)
        n = C.c_int()
        self._check(self._lib.vp_detect_postprocess(self._h, _ptr(raw), 0, raw.shape[0], raw.shape[1], float(conf_thresh), float(iou_thresh), _ptr(out), cap,
                                                    C.byref(n)), "vp_detect_postprocess")
        k = min(n.value, cap)
        det = out[:k].copy()
        det[:, 5] = out[:k, 5].view(np.int32).astype(np.float32)   # class_id is an int in the struct
        return det, n.value
