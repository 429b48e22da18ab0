"""MI355X-native (gfx950) engine for the VisionPilot per-frame hot path.

Product code only: the HIP engine behind the C ABI (csrc/, include/vp_hip.h), its ctypes binding (lib.py),
the weight exporter / ONNX reader (weights.py), the synthetic weight + frame generators every benchmark and test draws from
(synthetic.py) and the Python operator API mirroring Models/inference/*_infer.py (infer.py).
Nothing here imports ``oracle`` and nothing here falls back to the CPU.
"""
__version__ = "0.1.0"
