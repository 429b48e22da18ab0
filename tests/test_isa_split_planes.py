"""The (hi, lo) planes of the parity mode must be taken from ONE fp32 value: hi = fp16(v), lo = fp16(v - hi).  The gfx950 backend likes
to fold `fp16(a * b)` into v_fma_mixlo_f16 / v_fma_mixhi_f16 (the exact product rounded once) while the subtraction still uses the fp32
product: the planes of a few elements in 2^12 then disagree by one fp16 ulp -- 1e-4 of a layer's output, invisible to the CPU emulation
(it compiles no such fold) and caught on the GPU only as a parity failure of a whole network (round 3, mbconv_back_kernel).  This test
compiles the kernels that split in registers for gfx950 and asserts the fold is absent from them (hipcc cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "autoware_vision_pilot_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _isa(unit, tmp_path):
    out = tmp_path / (unit + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(CSRC, unit), "-o", str(out)], check=True, capture_output=True)
    kernels, name = {}, None
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            kernels[name] = 0
        elif name and re.search(r"v_fma_mix(lo|hi)_f16", line):
            kernels[name] += 1
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_mixed_precision_fold_in_the_kernels_that_split_in_registers(tmp_path):
    k = _isa("kernels_mbconv.hip", tmp_path)
    split = {n: c for n, c in k.items() if "mbconv_back_kernel" in n or "mbconv_front_kernel" in n}
    assert len(split) >= 6, sorted(k)                       # 2 back + 4 front instantiations
    assert not any(split.values()), {n: c for n, c in split.items() if c}
    k = _isa("kernels_backbone.hip", tmp_path)
    # se_gate_scale (the batched / fp16 path's squeeze-excite tail) writes scaled projection weights as (hi, lo) planes; the depthwise
    # kernels' split instantiations (second template argument true: ...ELb1E...) store (hi, lo) activations
    split = {n: c for n, c in k.items() if "se_gate_scale_kernel" in n or re.search(r"dwconv_pool_kernelILi\dELb1E", n)}
    assert len(split) >= 4, sorted(k)
    assert not any(split.values()), {n: c for n, c in split.items() if c}
