"""Minimal protobuf WRITER for the tests that lay out ONNX ModelProto files by hand (the `onnx` package is absent)."""
import numpy as np


def _pb_varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _pb_ld(field, payload):
    return _pb_varint((field << 3) | 2) + _pb_varint(len(payload)) + payload


def _onnx_tensor(name, a, raw=True):
    dt = {np.dtype("float32"): 1, np.dtype("float16"): 10, np.dtype("int64"): 7}[a.dtype]
    msg = b"".join(_pb_varint((1 << 3) | 0) + _pb_varint(d) for d in a.shape) + _pb_varint((2 << 3) | 0) + _pb_varint(dt)
    msg += _pb_ld(8, name.encode())
    msg += _pb_ld(9, a.tobytes()) if raw else _pb_ld(4, a.astype("<f4").tobytes())
    return msg


def onnx_model(tensors):
    """ModelProto bytes whose graph holds `tensors` (name -> array) as raw_data initializers and nothing else."""
    graph = _pb_ld(2, b"main_graph") + b"".join(_pb_ld(5, _onnx_tensor(k, np.ascontiguousarray(v))) for k, v in tensors.items())
    return _pb_varint((1 << 3) | 0) + _pb_varint(8) + _pb_ld(2, b"pytorch") + _pb_ld(7, graph)
