#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_conv.py) into per-kernel HBM-side bytes per launch.

python tools/pmc_summarize.py <fetch_dir> <write_dir> <out.json>
Counter unit is KiB (cdna_hip_programming.md section 7).  The calibration dispatch (elementwise add, 1 GiB read + 1 GiB
written) fixes the counter -> byte factors measured in the SAME run; they are applied to every kernel."""
import csv, glob, json, os, re, sys
from collections import defaultdict

def load(d, counter):
    rows = defaultdict(list)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
GIB = float(1 << 30)
def calib(rows):
    for k, v in rows.items():
        if "elementwise" in k and len(v) >= 3 and max(v) > 1e5:
            return k, sorted(v)[len(v) // 2]
    raise SystemExit("calibration dispatch not found")
kf, cf = calib(fetch)
kw, cw = calib(write)
f_factor, w_factor = GIB / (cf * 1024.0), GIB / (cw * 1024.0)
out = {"unit": "bytes per launch, HBM/fabric side of L2 (Infinity Cache hits included)",
       "calibration": {"kernel": kf, "known_bytes_read": GIB, "known_bytes_written": GIB, "FETCH_SIZE_KiB": cf, "WRITE_SIZE_KiB": cw,
                       "fetch_factor": round(f_factor, 4), "write_factor": round(w_factor, 4)},
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("void vp::") and "vp::" not in k:
        continue
    fv, wv = fetch.get(k, []), write.get(k, [])
    name = re.sub(r"\s+", " ", k)
    out["kernels"][name] = {"launches_seen": max(len(fv), len(wv)),
                            "fetch_bytes": round(sum(fv) / max(1, len(fv)) * 1024.0 * f_factor),
                            "write_bytes": round(sum(wv) / max(1, len(wv)) * 1024.0 * w_factor)}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash: bench.py ignores a PMC file taken on other kernel sources)
out["source_hash"] = bench.kernel_source_hash()
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["calibration"]))
for k, v in sorted(out["kernels"].items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]) * kv[1]["launches_seen"])[:12]:
    print(f'{v["launches_seen"]:4d} x  fetch {v["fetch_bytes"]/1e6:8.2f} MB  write {v["write_bytes"]/1e6:8.2f} MB  {k[:110]}')
