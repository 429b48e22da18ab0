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
