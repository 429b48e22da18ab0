#!/usr/bin/env python3
"""Static resource table of every device kernel in libvp_hip.so (no GPU needed): registers, spills, scratch, occupancy.
    python tools/kernel_resources.py > profiles/r01_kernel_resources.tsv
Compiles each csrc/*.hip with `hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage` (the Makefile's flags)
and tabulates the remarks.  Dynamic LDS (most MFMA kernels) is set by the launcher and not part of the static figure."""
import glob
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autoware_vision_pilot_amd", "csrc")
rows = []
with tempfile.TemporaryDirectory() as d:
    for src in sorted(glob.glob(os.path.join(CSRC, "kernels_*.hip"))):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", os.path.join(d, "o.o"),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (.*?) \[-Rpass-analysis", line)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"):
                cur = {"file": os.path.basename(src), "name": t.split(":", 1)[1].strip()}
                rows.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.split(":", 1)
                cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("# per-kernel resource usage of libvp_hip.so's device code (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)")
print("# file\tkernel\tVGPRs\tAGPRs\tSGPRs\tscratch B/lane\toccupancy waves/SIMD\tstatic LDS B\tVGPR spills\tSGPR spills")
seen = set()
for r, n in zip(rows, names):
    n = re.sub(r"\((vp::)?[A-Za-z].*\)$", "", n.replace("void ", "")).replace("vp::", "")
    if (r["file"], n) in seen:
        continue
    seen.add((r["file"], n))
    print("\t".join([r["file"], n] + [r.get(k, "?") for k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
                                                               "LDS Size [bytes/block]", "VGPRs Spill", "SGPRs Spill")]))
