"""Data-race check of the HIP kernels on the CPU (tests/emul/race_check.cpp): the kernel sources compiled for the host with
-fsanitize=thread, one OS thread per work-item, and only the GPU's own happens-before edges (__syncthreads, wave operations,
kernel boundaries).  ThreadSanitizer must stay silent on the production kernels -- LDS ring buffers of the MFMA convolutions,
fused pools, reductions -- and must report the deliberately racy canary kernel (so silence means something)."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))


@pytest.fixture(scope="module")
def race_check():
    import build as emul_build

    if not os.path.exists(emul_build.CLANG):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    try:
        return emul_build.build_race_check()
    except RuntimeError as e:  # e.g. a toolchain without the ThreadSanitizer runtime
        if "tsan" in str(e).lower() or "sanitize" in str(e).lower():
            pytest.skip("ThreadSanitizer runtime not available")
        raise


def test_canary_race_is_reported(race_check):
    r = subprocess.run([race_check, "--canary"], capture_output=True, text=True, timeout=300)
    assert "ThreadSanitizer: data race" in r.stderr and "canary_kernel" in r.stderr


def test_production_kernels_are_race_free(race_check):
    # the two halves of the quick subset (convolution family / everything else) side by side: a run is bound by thread creation (one OS thread per
    # work-item), not by the cores
    procs = [subprocess.Popen([race_check, "--quick", half], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for half in ("--conv-only", "--no-conv")]
    for p in procs:
        out, err = p.communicate(timeout=1500)
        assert "ThreadSanitizer" not in err, err[:3000]
        assert p.returncode == 0 and "all launches done" in out, (p.returncode, out[-500:], err[-500:])
