"""Multi-camera sharding: one camera per GPU, one process per GPU (BASELINE configs[3]).

The reference has no inference-side collective (single camera, batch 1: onnx_runtime_backend.cpp:59).  Cameras are
independent units, so the path shards with NO data-path collective; the only exchange is the optional gather of
fixed-size per-camera result records (mask / lane logits) for a downstream fused consumer.  Payloads are <= 360 KB
per rank, latency-bound over xGMI, so it is a single all_gather (RCCL on GPU tensors, gloo on CPU tensors in the
tests) -- no bucketing, no ring tuning.
"""
from dataclasses import dataclass

import numpy as np


def cameras_for_rank(n_cameras, rank, world):
    """Round-robin camera -> rank map (camera i on rank i when n_cameras == world)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_cameras, world))


@dataclass
class ResultRecord:
    """Fixed-size per-camera record: header (camera id, frame id, h, w) + u8 mask payload."""
    camera: int
    frame: int
    mask: np.ndarray  # HxW uint8

    HEADER = 16  # 4 x int32

    def pack(self):
        hdr = np.array([self.camera, self.frame, self.mask.shape[0], self.mask.shape[1]], dtype=np.int32).view(np.uint8)
        return np.concatenate([hdr, np.ascontiguousarray(self.mask, dtype=np.uint8).ravel()])

    @staticmethod
    def unpack(buf):
        hdr = np.ascontiguousarray(buf[:ResultRecord.HEADER]).view(np.int32)
        cam, frame, h, w = (int(v) for v in hdr)
        return ResultRecord(cam, frame, np.asarray(buf[ResultRecord.HEADER:ResultRecord.HEADER + h * w]).reshape(h, w).copy())

    @staticmethod
    def nbytes(h, w):
        return ResultRecord.HEADER + h * w


def gather_records(record, dist, device="cpu"):
    """all_gather one packed record per rank; returns the list of ResultRecord in rank order on every rank."""
    import torch

    mine = torch.from_numpy(record.pack()).to(device)
    out = torch.empty(dist.get_world_size() * mine.numel(), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, mine)
    flat = out.cpu().numpy()
    n = mine.numel()
    return [ResultRecord.unpack(flat[i * n:(i + 1) * n]) for i in range(dist.get_world_size())]


def max_over_ranks(value, dist, device="cpu"):
    """Timing reduction used by bench.py: MAX of a float over ranks."""
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
