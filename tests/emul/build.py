"""TEST INFRASTRUCTURE: build tests/emul/_build/libvp_emul.so -- the non-MFMA kernel sources compiled for the HOST on top
of the HIP-on-CPU shim (shim/hip/hip_runtime.h), so the CPU suite can execute the real kernel code against the oracle."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "autoware_vision_pilot_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
KERNEL_FILES = ("kernels_misc.hip", "kernels_backbone.hip", "kernels_autodrive.hip")
HEADERS = ("common.hpp", "kernels.hpp", "act_io.hpp")


def build(force=False):
    lib = os.path.join(OUT, "libvp_emul.so")
    srcs = [os.path.join(CSRC, f) for f in KERNEL_FILES + HEADERS] + [os.path.join(HERE, "harness.cpp"), os.path.join(HERE, "shim", "hip", "hip_runtime.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(s) for s in srcs):
        return lib
    os.makedirs(OUT, exist_ok=True)
    for f in KERNEL_FILES + HEADERS:
        text = open(os.path.join(CSRC, f)).read().replace("extern __shared__", "extern")
        with open(os.path.join(OUT, f.replace(".hip", ".cpp")), "w") as o:
            o.write(text)
    cmd = [CLANG, "-std=c++20", "-O1", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wno-everything", "-I", os.path.join(HERE, "shim"), "-I", OUT,
           os.path.join(HERE, "harness.cpp"), "-o", lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("emulation build failed:\n" + r.stderr[-4000:])
    return lib


if __name__ == "__main__":
    print(build(force=True))
