"""TEST INFRASTRUCTURE: build tests/emul/_build/libvp_emul.so -- EVERY csrc/ source (kernels, engine, C ABI) compiled for the
HOST on top of the HIP-on-CPU shim (shim/hip/hip_runtime.h), so the CPU suite can execute the real kernel and engine code
against the oracle.  Source rewrites (listed here, nothing else differs from what hipcc compiles):
  * `extern __shared__`  ->  `extern VP_EMU_LDS`           (dynamic LDS is per-worker storage in harness.cpp)
  * the inline-asm statements (an AGPR read, two register pins) -> their plain C++ equivalents;
  * the register-budget attribute of one kernel (`amdgpu_waves_per_eu`) is dropped.
Linked with -Bsymbolic and meant to be dlopen-ed RTLD_LOCAL: it exports the same symbols as libvp_hip.so and must neither
capture nor be captured by that library when both live in one test process."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "autoware_vision_pilot_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
UNITS = ("kernels_conv.hip", "kernels_convt_rs.hip", "kernels_head.hip", "kernels_gemm_dma.hip", "kernels_conv3x3.hip", "kernels_conv3x3_x3.hip", "kernels_conv3x3_map.hip", "kernels_upconv.hip", "kernels_backbone.hip", "kernels_mbconv.hip",
         "kernels_misc.hip", "kernels_autodrive.hip", "kernels_detect.hip", "engine.cpp", "engine_dispatch.cpp", "engine_upconv.cpp", "engine_io.cpp", "onnx_reader.cpp", "vp_api.cpp", "vp_detect.cpp", "options.cpp")
HEADERS = ("common.hpp", "kernels.hpp", "act_io.hpp", "se_phases.hpp", "conv_epilogue.hpp", "lds_dma.hpp", "engine.hpp", "engine_internal.hpp", "vp_handle.hpp", "viridis_lut.inc")
REWRITES = (
    ('asm volatile("" : "+v"(lane_o_));', "(void)0;"),
    ("extern __shared__", "extern VP_EMU_LDS"),
    ('asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[i][j][4 * g + r]));', "v = acc[i][j][4 * g + r];"),
    ('asm volatile("" ::"v"(a_[i]), "v"(b_[j]));', "(void)0;"),
    ("__attribute__((amdgpu_waves_per_eu(WM == 4 ? 3 : 2))) ", ""),
    ('asm volatile("" : "+v"(v));', "(void)0;"),
)


def build(force=False):
    lib = os.path.join(OUT, "libvp_emul.so")
    deps = [os.path.join(CSRC, f) for f in UNITS + HEADERS] + [os.path.join(HERE, f) for f in ("harness.cpp", "build.py", os.path.join("shim", "hip", "hip_runtime.h"))]
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(s) for s in deps):
        return lib
    os.makedirs(OUT, exist_ok=True)
    for f in UNITS + HEADERS:
        text = open(os.path.join(CSRC, f)).read()
        for a, b in REWRITES:
            text = text.replace(a, b)
        text = text.replace('#include "../../include/vp_hip.h"', '#include "%s"' % os.path.join(ROOT, "include", "vp_hip.h"))
        with open(os.path.join(OUT, f.replace(".hip", ".cpp")), "w") as o:
            o.write(text)
    flags = [CLANG, "-std=c++20", "-O1", "-march=native", "-fPIC", "-pthread", "-ffp-contract=off", "-Wno-everything", "-I", os.path.join(HERE, "shim"), "-I", OUT]

    def cc(src):
        obj = os.path.join(OUT, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        r = subprocess.run(flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("emulation build failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    srcs = [os.path.join(OUT, f.replace(".hip", ".cpp")) for f in UNITS] + [os.path.join(HERE, "harness.cpp")]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run([CLANG, "-shared", "-pthread", "-Wl,-Bsymbolic", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("emulation link failed:\n" + r.stderr[-4000:])
    return lib


def build_race_check(force=False):
    """tests/emul/_build/race/race_check: the same sources + race_check.cpp under ThreadSanitizer (see shim: VP_EMU_TSAN)."""
    build(force)  # refreshes the rewritten sources under _build/
    out = os.path.join(OUT, "race")
    exe = os.path.join(out, "race_check")
    deps = [os.path.join(CSRC, f) for f in UNITS + HEADERS] + [os.path.join(HERE, f) for f in ("harness.cpp", "race_check.cpp", "build.py", os.path.join("shim", "hip", "hip_runtime.h"))]
    if not force and os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(s) for s in deps):
        return exe
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "vp_hip_path.h"), "w") as o:
        o.write('#include "%s"\n' % os.path.join(ROOT, "include", "vp_hip.h"))
    flags = [CLANG, "-std=c++20", "-O1", "-g", "-fPIC", "-pthread", "-ffp-contract=off", "-Wno-everything", "-fsanitize=thread", "-DVP_EMU_TSAN",
             "-I", os.path.join(HERE, "shim"), "-I", OUT, "-I", out]

    def cc(src):
        obj = os.path.join(out, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        r = subprocess.run(flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("race-check build failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    srcs = [os.path.join(OUT, f.replace(".hip", ".cpp")) for f in UNITS] + [os.path.join(HERE, "harness.cpp"), os.path.join(HERE, "race_check.cpp")]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run([CLANG, "-fsanitize=thread", "-pthread", "-o", exe] + objs, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("race-check link failed:\n" + r.stderr[-4000:])
    return exe


if __name__ == "__main__":
    print(build(force=True))
    print(build_race_check(force=True))
