#!/usr/bin/env python3
"""SQ counters of the 3x3 convolution kernels per dispatch -> table (matrix-pipe busy share, clock, wait shares).
    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
        SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d OUT -- python tools/pmc_conv.py fp16x3
    python tools/pmc_sq_summarize.py OUT > profiles/r0N_pmc_sq_conv.tsv
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) (the counter ticks
once per cycle a SIMD's matrix pipe is busy: 32 per v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md); wait_* / active = share of
SQ_WAVE_CYCLES.  The LAST (warm) dispatch of every (kernel, grid) pair is reported, in launch order."""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

root = sys.argv[1]
cc = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
if not cc:
    sys.exit("no counter_collection.csv")
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
disp = OrderedDict()
for f in cc:
    for r in csv.DictReader(open(f)):
        k = r["Dispatch_Id"]
        d = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "?"), "c": defaultdict(float)})
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"] and k not in dur:
            dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
last = OrderedDict()
for k, d in disp.items():
    if "conv3x3" not in d["name"] and "gemm_dma" not in d["name"] and "convt_rs" not in d["name"] and "upconv_x3" not in d["name"]:
        continue
    last[(d["name"], d["grid"])] = (k, d)
print("# kernel\tgrid_threads\tdur_us\tclock_GHz\tmfma_busy\twait_any\twait_inst_any\tactive_inst_any\tinsts_valu")
for (name, grid), (k, d) in last.items():
    c = d["c"]
    us = dur.get(k, 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    wave = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    clock = gui / 8.0 / (us * 1e3) if us > 0 else 0.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui > 0 else 0.0
    n = name.split("(")[0].replace("void ", "").replace("vp::", "")
    print(f"{n}\t{grid}\t{us:.1f}\t{clock:.2f}\t{busy:.3f}\t{c.get('SQ_WAIT_ANY', 0) / wave:.2f}\t{c.get('SQ_WAIT_INST_ANY', 0) / wave:.2f}\t"
          f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wave:.2f}\t{c.get('SQ_INSTS_VALU', 0):.3g}")
