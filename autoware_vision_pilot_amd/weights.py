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


def unpack_blob(blob):
    """Inverse of pack_state_dict: VPW1 bytes -> {name: float32 array}."""
    blob = bytes(blob)
    if blob[:4] != b"VPW1":
        raise ValueError("not a VPW1 blob")
    (count,), i, out = struct.unpack_from("<I", blob, 4), 8, {}
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", blob, i)
        name = blob[i + 2:i + 2 + nl].decode()
        i += 2 + nl
        nd = blob[i]
        dims = struct.unpack_from(f"<{nd}I", blob, i + 1)
        i += 1 + 4 * nd
        n = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(blob, dtype="<f4", count=n, offset=i).reshape(dims).copy()
        i += 4 * n
    if i != len(blob):
        raise ValueError("trailing bytes after the last tensor")
    return out


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


# ------------------------------------------------------------------------------------------------ ONNX files
# SURVEY.md 8f N2.  The reference ships / exports its models as ONNX (Models/exports/convert_pytorch_to_onnx.py:144-154:
# opset 18, export_params=True, do_constant_folding=True, external_data=False) and the ROS parameters name
# `model_path: *.onnx`.  This reads such a file with a self-contained protobuf wire-format parser (no `onnx` package):
#   * initializers that kept their state_dict names (plain convs, linears, stand-alone norms) pass through verbatim;
#   * a Conv / ConvTranspose whose weight and bias are exporter-made anonymous tensors ("onnx::Conv_626": Conv+BatchNorm
#     folded by constant folding) is named from the node's module scope ("/backbone/p3/p3.0/conv/Conv" ->
#     "backbone.p3.0.conv"), emitted as `<conv>.weight` + `<conv>.bias` with no norm tensors -- csrc/engine.cpp fold_conv
#     takes that form as is; a second invocation of the same module (suffix "_1", identical tensors) is dropped;
#   * a bias-free Linear exported as MatMul with a transposed anonymous weight becomes `<module>.weight`.
# Pinned against real exporter output: tests/golden/tiny_export.onnx (nested Sequentials, repeated module) and, where
# /root/reference exists, the reference's own AutoDrive module (oracle/pin_autodrive_onnx.py).
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


def _graph(path):
    """(initializers, nodes) of the model's graph; nodes = [(op_type, name, inputs, outputs)] in file order
    (ModelProto.graph = 7; GraphProto.node = 1, initializer = 5; NodeProto.input = 1, output = 2, name = 3, op_type = 4)."""
    with open(path, "rb") as fh:
        model = memoryview(fh.read())
    inits, nodes = {}, []
    for f, wt, v in _fields(model):
        if f != 7 or wt != 2:
            continue
        for gf, gwt, gv in _fields(v):
            if gwt != 2:
                continue
            if gf == 5:
                name, a = _tensor_proto(gv)
                if a is not None and name:
                    inits[name] = a
            elif gf == 1:
                ins, outs, name, op = [], [], "", ""
                for nf, nwt, nv in _fields(gv):
                    if nwt != 2:
                        continue
                    if nf == 1:
                        ins.append(bytes(nv).decode())
                    elif nf == 2:
                        outs.append(bytes(nv).decode())
                    elif nf == 3:
                        name = bytes(nv).decode()
                    elif nf == 4:
                        op = bytes(nv).decode()
                nodes.append((op, name, ins, outs))
    return inits, nodes


def _is_anonymous(name):
    return name.startswith("onnx::") or "." not in name


def _scope_prefix(node_name):
    """'/backbone/p5/p5.3/middle_block/conv2/conv2.0/conv/Conv' -> 'backbone.p5.3.middle_block.conv2.0.conv': the
    TorchScript exporter writes one path segment per module level and spells a container's child as '<container>.<idx>'."""
    comps = []
    for seg in node_name.strip("/").split("/")[:-1]:
        if comps and seg.startswith(comps[-1] + "."):
            comps[-1] = seg
        else:
            comps.append(seg)
    return ".".join(comps)


def load_onnx_state_dict(path):
    """name -> numpy array in the reference state_dict's naming (see the section comment); raises on a weight it cannot name."""
    inits, nodes = _graph(path)
    out = {k: v for k, v in inits.items() if not _is_anonymous(k) and v.dtype.kind == "f"}
    unnamed = []
    for op, name, ins, _ in nodes:
        if op in ("Conv", "ConvTranspose") and len(ins) >= 2 and ins[1] in inits and _is_anonymous(ins[1]):
            tensors = [("weight", inits[ins[1]])]
            if len(ins) >= 3 and ins[2] in inits:
                tensors.append(("bias", inits[ins[2]]))
        elif op == "MatMul" and len(ins) == 2 and ins[1] in inits and _is_anonymous(ins[1]) and inits[ins[1]].ndim == 2:
            tensors = [("weight", np.ascontiguousarray(inits[ins[1]].T))]
        else:
            continue
        prefix = _scope_prefix(name)
        if not prefix:
            unnamed.append(f"{op} node '{name}' ({ins[1]})")
            continue
        head, _, leaf = prefix.rpartition(".")
        base, sep, idx = leaf.rpartition("_")
        if sep and idx.isdigit():  # 2nd, 3rd ... invocation of one module: same tensors under the un-suffixed name
            first = (head + "." if head else "") + base
            if first + ".weight" in out and all(
                    first + "." + t in out and out[first + "." + t].shape == a.shape and np.array_equal(out[first + "." + t], a)
                    for t, a in tensors):
                continue
        for t, a in tensors:
            key = prefix + "." + t
            if key in out and not np.array_equal(out[key], a):
                raise ValueError(f"two different tensors map to '{key}' (node '{name}')")
            out[key] = a
    if unnamed:
        raise ValueError("cannot name exporter-folded weights without a module scope: " + "; ".join(unnamed[:4]))
    return out


def export_onnx(onnx_path, out_path):
    """ONNX file made by the reference's exporter settings -> VPW1 blob (floating-point tensors only)."""
    sd = load_onnx_state_dict(onnx_path)
    if not sd:
        raise ValueError(f"{onnx_path}: no floating-point weights found")
    with open(out_path, "wb") as f:
        f.write(pack_state_dict(sd))
    return out_path


def main(argv=None):
    """python -m autoware_vision_pilot_amd.weights <in.pth|in.pt|in.onnx> <out.vpw>"""
    import argparse

    ap = argparse.ArgumentParser(prog="python -m autoware_vision_pilot_amd.weights",
                                 description="Convert a reference checkpoint (.pth / .pt state_dict) or an ONNX file made by the "
                                             "reference's exporter to the engine's VPW1 weight blob.")
    ap.add_argument("src")
    ap.add_argument("dst")
    a = ap.parse_args(argv)
    out = export_onnx(a.src, a.dst) if a.src.lower().endswith(".onnx") else export_checkpoint(a.src, a.dst)
    n = len(unpack_blob(open(out, "rb").read()))
    print(f"{out}: {n} tensors")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
