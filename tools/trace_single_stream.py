#!/usr/bin/env python3
"""Developer tool: one engine, graph replays back to back on one stream -- run under `rocprofv3 --kernel-trace` to get every
kernel's start / duration / gap inside the replayed graph (what per-launch HIP events in eager mode cannot show).
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/trace_single_stream.py [kind] [precision] [replays]
    python tools/trace_single_stream.py --summarise OUT > table.tsv     (last replay: kernel, start us, duration us, gap us)"""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(root):
    import csv

    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # replays are separated by the preprocess kernel
    starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[2] or "pil_resample_h" in r[2]]
    if len(starts) < 3:
        sys.exit("no replays found")
    a, b = starts[-2], starts[-1]     # the last COMPLETE replay
    seg = rows[a:b]
    t0 = seg[0][0]
    print("# kernel\tstart_us\tduration_us\tgap_before_us")
    prev_end = None
    tot_d = tot_g = 0.0
    for s, e, n in seg:
        n = n.split("(")[0].replace("void ", "").replace("vp::", "")
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{n[:70]}\t{(s - t0) / 1e3:.1f}\t{(e - s) / 1e3:.1f}\t{gap:.1f}")
        tot_d += (e - s) / 1e3
        tot_g += max(gap, 0.0)
        prev_end = e
    print(f"# {len(seg)} kernels, busy {tot_d:.1f} us, gaps {tot_g:.1f} us, span {(seg[-1][1] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
        sys.exit(0)
    import numpy as np  # noqa: F401
    import torch  # noqa: F401

    from autoware_vision_pilot_amd import lib, synthetic, weights as vw

    lib.options_from_env()  # developer tool: VP_* knobs from the environment -> vp_set_option (the library itself never reads the environment)

    kind = sys.argv[1] if len(sys.argv) > 1 else "sceneseg"
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp16x3"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    seed = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}[kind]
    eng = lib.Engine(kind, vw.pack_state_dict(synthetic.make_state_dict(kind, seed)), precision=prec)
    eng.upload_frame(synthetic.synthetic_frame(720, 1280, 1))
    for _ in range(n):
        eng.enqueue()
    eng.sync()
    eng.close()
