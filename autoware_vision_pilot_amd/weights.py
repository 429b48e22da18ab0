"""state_dict -> "VPW1" weight blob for libvp_hip (parsed by csrc/engine.cpp WeightBlob).

Layout (little endian): b"VPW1", u32 count, then per tensor
    u16 name_len, name bytes, u8 ndim, u32 dims[ndim], float32 data (C order).
Keys are the reference state_dict keys verbatim, so a real checkpoint exported with
``export_checkpoint`` loads unchanged; ``num_batches_tracked`` and non-float tensors are dropped.
BatchNorm folding and fp16 (hi, lo) packing happen inside the engine, not here.
"""
import struct

import numpy as np


def pack_state_dict(sd):
    """sd: mapping name -> array-like (numpy or torch tensor).  Returns bytes."""
    items = []
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if a.dtype.kind != "f":
            continue
        items.append((k, np.ascontiguousarray(a, dtype=np.float32)))
    out = [b"VPW1", struct.pack("<I", len(items))]
    for k, a in items:
        kb = k.encode()
        out.append(struct.pack("<H", len(kb)))
        out.append(kb)
        out.append(struct.pack("<B", a.ndim))
        out.append(struct.pack(f"<{a.ndim}I", *a.shape))
        out.append(a.tobytes())
    return b"".join(out)


def export_checkpoint(pth_path, out_path):
    """Convert a reference ``.pth`` state_dict (Models/inference/scene_seg_infer.py:30-31) to a blob file."""
    import torch  # only needed for reading the pickle

    sd = torch.load(pth_path, weights_only=True, map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]  # AutoDrive checkpoints wrap the state_dict (visualizations/AutoDrive/video_visualization.py:43-44)
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    with open(out_path, "wb") as f:
        f.write(pack_state_dict(sd))
    return out_path


# ------------------------------------------------------------------------------------------------ ONNX initializers
# SURVEY.md 8f N2 (partial).  The reference ships / exports its models as ONNX (Models/exports/convert_pytorch_to_onnx.py:
# opset 18, export_params=True, external_data=False), and the ROS parameters name `model_path: *.onnx`.  This reads the
# GRAPH INITIALIZERS of such a file with a self-contained protobuf wire-format parser (no `onnx` package) so that files
# whose initializers keep the state_dict names -- torch.export-based exports, or do_constant_folding=False -- convert to
# the engine's blob.  Not handled (stated, not guessed): exporter-fused Conv+BatchNorm initializers with anonymous
# names ("onnx::Conv_123"); no exporter-made file exists in the reference tree to pin that mapping against.
def _varint(buf, i):
    r, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, i
        s += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message; length-delimited values as memoryview slices."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = bytes(buf[i:i + 8]), i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = bytes(buf[i:i + 4]), i + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


_ONNX_DTYPES = {1: np.float32, 10: np.float16, 11: np.float64, 7: np.int64, 6: np.int32}


def _tensor_proto(buf):
    """onnx.TensorProto: dims=1, data_type=2, float_data=4, int64_data=7, name=8, raw_data=9, double_data=10."""
    dims, dtype, name, raw, floats, doubles, int64s = [], 1, "", None, [], [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            if wt == 0:
                dims.append(v)
            else:  # packed
                j = 0
                while j < len(v):
                    d, j = _varint(v, j)
                    dims.append(d)
        elif f == 2:
            dtype = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            floats.append(np.frombuffer(bytes(v), dtype="<f4") if wt == 2 else np.frombuffer(v, dtype="<f4"))
        elif f == 10:
            doubles.append(np.frombuffer(bytes(v), dtype="<f8") if wt == 2 else np.frombuffer(v, dtype="<f8"))
        elif f == 7 and wt == 2:
            j = 0
            while j < len(v):
                d, j = _varint(v, j)
                int64s.append(d)
    if dtype not in _ONNX_DTYPES:
        return name, None
    if raw is not None:
        a = np.frombuffer(raw, dtype=np.dtype(_ONNX_DTYPES[dtype]).newbyteorder("<"))
    elif floats:
        a = np.concatenate(floats)
    elif doubles:
        a = np.concatenate(doubles)
    elif int64s:
        a = np.array(int64s, dtype=np.int64)
    else:
        a = np.zeros(0, dtype=_ONNX_DTYPES[dtype])
    return name, a.reshape(dims) if dims else a.reshape(())


def load_onnx_initializers(path):
    """name -> numpy array for every initializer of the graph (ModelProto.graph = field 7, GraphProto.initializer = 5)."""
    with open(path, "rb") as fh:
        model = memoryview(fh.read())
    out = {}
    for f, wt, v in _fields(model):
        if f == 7 and wt == 2:
            for gf, gwt, gv in _fields(v):
                if gf == 5 and gwt == 2:
                    name, a = _tensor_proto(gv)
                    if a is not None and name:
                        out[name] = a
    return out


def export_onnx(onnx_path, out_path):
    """ONNX file whose initializers carry the reference state_dict names -> VPW1 blob (floating-point tensors only)."""
    sd = load_onnx_initializers(onnx_path)
    anon = [k for k in sd if k.startswith("onnx::")]
    if anon:
        raise ValueError(f"{len(anon)} exporter-fused anonymous initializers (e.g. {anon[0]}): export with the torch.export-based "
                         "exporter or do_constant_folding=False, or convert the .pth checkpoint with export_checkpoint")
    with open(out_path, "wb") as f:
        f.write(pack_state_dict(sd))
    return out_path
