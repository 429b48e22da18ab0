#!/usr/bin/env python3
"""Headline benchmark: frames/s of the VisionPilot per-frame hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched by torch.distributed.run, one rank
per GPU).  A *step* is one pass of the hot path over one frame: integer-bilinear preprocess of a synthetic
1280x720 BGR frame that is already resident in HBM -> SceneSeg forward (EfficientNet-B0 encoder + context +
neck + head, 367 GFLOP) -> argmax decode, replayed as one hipGraph on the engine's own stream
(BASELINE.json configs[1]: "SceneSeg 1280x720 batch=1 on 1 MI355X, fp16").  Random-init weights of that
architecture, synthetic frame: data = "synthetic".

Multi-GPU: the path shards by camera: rank r owns camera r, weights replicated, no data-path
collective (the reference has none) -> "scaling": "weak".  ``--gather`` adds the optional per-frame RCCL
all-gather of the per-camera masks (BASELINE configs[3]).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : dominant kernel family's ALGORITHMIC TFLOP/s from per-launch HIP-event timing on the engine
                 stream vs the dense fp16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline : the CPU oracle (torch fp32 restatement of the reference path) timed on this box's host cores
                 on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before libvp_hip: one shared HIP runtime)

PEAK_FP16_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
FRAME_GFLOP = {"sceneseg": 367.0, "scene3d": 397.0, "domainseg": 366.6, "egolanes": 196.7}


def pmc_traffic(tag):
    """HBM-side bytes per launch of the kernel instantiation behind ``tag`` from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of tools/pmc_conv.py, calibrated in-run on a
    1 GiB streaming kernel: FETCH_SIZE x2, WRITE_SIZE x1 on gfx950; tools/pmc_summarize.py).  None if not profiled."""
    import re
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    m = re.match(r"conv3x3_halo<co(\d+),px(\d+),x(\d)(,regepi)?>", tag)
    if m:
        co, px, x, reg = int(m.group(1)), int(m.group(2)), m.group(3), m.group(4)
        pat = rf"conv3x3_halo_kernel<{co}, {px // 16}, 16, \d, \d, {'true' if x == '3' else 'false'}, 0, {'true' if reg else 'false'}>"
    else:
        m = re.match(r"conv_gemm<bk(\d+),co(\d+),px(\d+),x(\d)(?:,regepi(\d))?>", tag)
        if not m:
            pat = r"vp::" + re.escape(tag.split("<")[0]) + r"_kernel" + (re.escape("<" + tag.split("<")[1]) if "<" in tag else "")
        else:
            pat = (rf"conv_gemm_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}, \d, \d, {'true' if m.group(4) == '3' else 'false'}, "
                   rf"\d, (true|false), {m.group(5) or 0}>")
    hit = [v for k, v in ks.items() if re.search(pat, k)]
    n = sum(v["launches_seen"] for v in hit)
    if not n:
        return None
    f = sum(v["fetch_bytes"] * v["launches_seen"] for v in hit) / n
    w = sum(v["write_bytes"] * v["launches_seen"] for v in hit) / n
    return {"bytes": round(f + w), "fetch_bytes": round(f), "write_bytes": round(w)}


def cpu_baseline(kind, sd, frame, seconds, nthreads=0):
    """The CPU oracle (oracle/nets.py, torch fp32) on `nthreads` host threads (0 = all, capped at 32): whole frames
    (preprocess + forward + decode) for about `seconds`.  SURVEY.md 8(d) also asks for a per-core figure: --cpu-threads 1
    (tools/cpu_baseline.py runs this leg alone, no GPU needed)."""
    from oracle import nets, pre_post

    nthreads = nthreads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(nthreads)
    tsd = nets.to_torch(sd)
    n_done, t_cpu = 0, 0.0
    nets.forward(kind, tsd, torch.from_numpy(pre_post.preprocess(frame)))  # warm-up
    while t_cpu < seconds and n_done < 50:
        t1 = time.perf_counter()
        x = torch.from_numpy(pre_post.preprocess(frame))
        y = nets.forward(kind, tsd, x)[0].numpy()
        pre_post.seg_mask_u8(y) if kind != "egolanes" else pre_post.egolanes_priority_mask(y)
        t_cpu += time.perf_counter() - t1
        n_done += 1
    return {"value": round(n_done / t_cpu, 4), "unit": "frames/s", "cores": nthreads, "kind": "port",
            "sample": f"{n_done} frames of the same 1280x720 workload (preprocess+forward+decode), torch "
                      f"{torch.__version__} CPU fp32, {t_cpu:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--kind", default="sceneseg")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp16x3"])
    ap.add_argument("--frame", default="1280x720")
    ap.add_argument("--streams", type=int, default=3,
                    help="frames in flight per GPU (independent engines/HIP streams, round-robin): the latency-bound "
                         "encoder of frame n+1 overlaps the MFMA-bound decoder of frame n")
    ap.add_argument("--batch", type=int, default=1,
                    help="EXPERIMENTAL, default off: cameras per batched-encoder pass (vp_create_batched + one shared-prefix head per "
                         "camera); a step is then one pass = BATCH frames.  The reported default configuration is --batch 1")
    ap.add_argument("--gather", action="store_true", help="all-gather per-camera masks every step (RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = all host cores, max 32)")
    ap.add_argument("--latency-iters", type=int, default=100)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))

    from autoware_vision_pilot_amd import lib, synthetic, weights as vw

    fw, fh = (int(v) for v in args.frame.split("x"))
    seed = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}[args.kind]
    sd = synthetic.make_state_dict(args.kind, seed)
    blob = vw.pack_state_dict(sd)
    frame = synthetic.synthetic_frame(fh, fw, 10 + rank)  # camera r
    if args.batch > 1:
        class Group:  # one batched encoder + one head per camera, all on the encoder's stream
            def __init__(self):
                self.enc = lib.Engine(args.kind, blob, precision=args.precision, gpu_id=local_rank, frames=args.batch)
                self.heads = [lib.Engine(args.kind, blob, precision=args.precision, gpu_id=local_rank, base=self.enc, frame_index=f)
                              for f in range(args.batch)]
                for f in range(args.batch):
                    self.enc.upload_frame(synthetic.synthetic_frame(fh, fw, 10 + rank + 100 * f), index=f)

            def enqueue(self):
                self.enc.enqueue()
                for h in self.heads:
                    h.enqueue()

            def sync(self):
                self.enc.sync()

            def close(self):
                for h in self.heads:
                    h.close()
                self.enc.close()

        engines = [Group() for _ in range(max(1, args.streams))]
        for e in engines:
            e.enqueue()
            e.enqueue()
            e.sync()
        eng = engines[0].heads[0]
        if args.gather:
            raise SystemExit("--gather is a --batch 1 option")
    else:
        engines = [lib.Engine(args.kind, blob, precision=args.precision, gpu_id=local_rank) for _ in range(max(1, args.streams))]
        eng = engines[0]
        for e in engines:
            e.upload_frame(frame)  # resident in HBM before the timed region
            e.enqueue()            # first pass is eager (sets kernel attributes), second captures the graph
            e.enqueue()
            e.sync()

    gather_buf = mask_t = None
    if args.gather and dist is not None:
        mask_t = torch.empty(320 * 640 if args.kind != "egolanes" else 80 * 160, dtype=torch.uint8, device="cuda")
        gather_buf = torch.empty(world * mask_t.numel(), dtype=torch.uint8, device="cuda")

    counter = [0]

    def step():
        e = engines[counter[0] % len(engines)]
        counter[0] += 1
        e.enqueue()
        if gather_buf is not None:
            e.copy_outputs_device(None, mask_t.data_ptr())
            e.sync()
            dist.all_gather_into_tensor(gather_buf, mask_t)

    def fence():
        for e in engines:
            e.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fps_total = world * args.steps * args.batch / elapsed

    # ---- per-frame latency (sync per iteration, benchmark.py:17-47 protocol), rank-local
    lat = []
    for _ in range(args.latency_iters):
        t1 = time.perf_counter()
        (engines[0] if args.batch > 1 else eng).enqueue()  # batch mode: one whole pass (encoder + BATCH heads)
        (engines[0] if args.batch > 1 else eng).sync()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat = np.array(lat)

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel family: per-launch HIP events on the engine stream (eager replay)
        ms = eng.profile_layers(10)
        layers, kernels = eng.layers(), eng.layer_kernels()
        fam = {}
        for (name, fl, by), k, t in zip(layers, kernels, ms):
            f = fam.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, n=0, worst=("", 0.0)))
            f["ms"] += float(t)
            f["flops"] += fl
            f["bytes"] += by
            f["n"] += 1
            if t > f["worst"][1]:
                f["worst"] = (name, float(t))
        # dominant = the single kernel instantiation (= one rocprofv3 kernel name) with the largest total time; "+splitk"
        # ops are two launches (conv + finish kernel) per timing interval, so they cannot give a per-kernel duration
        # The path is a dense contraction (bound = MFMA): the roofline kernel is the instantiation with the
        # largest total time among those carrying >= 5 % of the frame's FLOPs; `by_time` lists the top kernels of ANY kind
        # (the latency-bound depthwise launches are within a few us of it in eager timing) with their own bound and fraction.
        single = [k for k in fam if "+splitk" not in k]
        tot_fl = sum(f["flops"] for f in fam.values())
        heavy = [k for k in single if fam[k]["flops"] >= 0.05 * tot_fl]
        dom = max(heavy or single, key=lambda k: fam[k]["ms"])
        d = fam[dom]

        def frac_of(k):
            f = fam[k]
            if k.startswith("conv"):
                return "mfma", f["flops"] / (f["ms"] * 1e-3) / 1e12 / PEAK_FP16_TFLOPS
            return "hbm", f["bytes"] / (f["ms"] * 1e-3) / 1e9 / 8000.0

        by_time = [{"kernel": k, "launches": fam[k]["n"], "time_share": round(fam[k]["ms"] / float(ms.sum()), 3),
                    "bound": frac_of(k)[0], "frac": round(frac_of(k)[1], 4)}
                   for k in sorted(single, key=lambda k: -fam[k]["ms"])[:4]]
        frame_tflops = FRAME_GFLOP[args.kind] * (fps_total / world) / 1e3
        if dom.startswith("conv"):
            achieved, peak, unit, bound = d["flops"] / (d["ms"] * 1e-3) / 1e12, PEAK_FP16_TFLOPS, "TFLOP/s", "mfma"
        else:
            achieved, peak, unit, bound = d["bytes"] / (d["ms"] * 1e-3) / 1e9, 8000.0, "GB/s", "hbm"
        tr = pmc_traffic(dom) if args.precision == "fp16" and args.kind == "sceneseg" else None
        roofline = {
            "bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "traffic": (tr or {}).get("bytes"),
            "traffic_detail": dict(tr, source="profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                              "tools/pmc_conv.py, average over that instantiation's launches)") if tr else None,
            "kernel": dom, "launches_per_frame": d["n"], "avg_launch_us": round(1e3 * d["ms"] / d["n"], 2),
            "algorithmic_gflop_per_launch": round(d["flops"] / d["n"] / 1e9, 3),
            "algorithmic_mb_per_launch": round(d["bytes"] / d["n"] / 1e6, 3),
            "slowest_layer": d["worst"][0], "slowest_layer_us": round(1e3 * d["worst"][1], 1),
            "kernel_time_share": round(d["ms"] / float(ms.sum()), 3), "by_time": by_time,
            "whole_frame": {"achieved": round(frame_tflops, 2), "frac": round(frame_tflops / PEAK_FP16_TFLOPS, 4),
                            "gflop_per_frame": FRAME_GFLOP[args.kind], "unit": "TFLOP/s"},
            "sustained_mfma_peak": {"value": 1700.0, "unit": "TFLOP/s",
                                    "note": "bare v_mfma_f32_32x32x16_f16 loop on this GPU: 1570-1820 TFLOP/s, shader clock drops to "
                                            "1.55-1.85 GHz under MFMA load (tools/mfma_peak.hip, profiles/r01_mfma_peak.txt)"},
            "note": "per-launch HIP events on the engine stream (eager replay, single stream); fp16x3 issues 3 MFMAs per "
                    "algorithmic product, achieved counts algorithmic FLOPs only",
        }
        out = {
            "metric": "frames/sec (SceneSeg 1280x720 -> 640x320 net input, preprocess+forward+decode)",
            "value": round(fps_total, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp16" if args.precision == "fp16" else "fp16x3(fp32-class)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {args.kind} {fw}x{fh} batch=1, one camera per GPU, {args.precision}",
                       "frames_per_step_per_gpu": args.batch, "batched_encoder": args.batch > 1, "net_input": "1x3x320x640", "gather": bool(args.gather),
                       "frames_in_flight_per_gpu": len(engines)},
            "fps_per_gpu": round(fps_total / world, 2),
            "p50_ms": round(float(np.percentile(lat, 50)), 4), "p99_ms": round(float(np.percentile(lat, 99)), 4),
            # FpsTimer-style split (common/benchmark/fps_timer.cpp:37-63) from the per-launch HIP events (eager, single stream)
            "split_us": {"preprocess": round(1e3 * sum(float(t) for (n, _, _), t in zip(layers, ms) if n == "preprocess"), 1),
                         "inference": round(1e3 * sum(float(t) for (n, _, _), t in zip(layers, ms) if n not in ("preprocess", "decode")), 1),
                         "output_decode": round(1e3 * sum(float(t) for (n, _, _), t in zip(layers, ms) if n == "decode"), 1)},
            "roofline": roofline,
        }
        # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.kind, sd, frame, args.cpu_seconds, args.cpu_threads)
    for e in engines:
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
