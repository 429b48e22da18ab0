#!/usr/bin/env python3
"""Headline benchmark: frames/s of the VisionPilot per-frame hot path on MI355X, on BASELINE.json's metric configuration.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched by torch.distributed.run, one rank per GPU).

Workload (BASELINE.json ``metric``: "frames/sec/GPU (SceneSeg+Scene3D 1280x720); p50 per-frame latency"): one synthetic
1280x720 BGR camera frame per GPU goes through integer-bilinear preprocess -> EfficientNet-B0 encoder (run ONCE, shared:
the reference builds Scene3D on SceneSeg's pre-trained backbone, scene_3d_network.py:9-13) -> SceneSeg context + neck + head
-> argmax decode, and -> Scene3D context + neck + head (vp_create + vp_create_shared engines on one HIP stream, each replayed
as a hipGraph).  760.9 GFLOP per frame.  A *step* is one such frame.  Random-init weights of that architecture, synthetic
frame: data = "synthetic".

``value`` is measured in the PARITY mode (fp16x3: every tensor a (hi, lo) fp16 pair, three fp16 MFMAs per product, fp32
accumulate -- floats within 1e-3 of the fp32 oracle, class maps equal to the oracle's except tie flips inside that float tolerance
(counted per pass in profiles/r06_parity_sweep.tsv; tests/test_gpu_networks.py, tests/test_gpu_parity_sweep.py), with the frame
resident in HBM and ``--streams`` frames in flight per GPU.  The same JSON line also carries
  single_stream_fps / p50_ms : one frame at a time (back-to-back / synchronised per frame), parity mode
  fp16_value ...             : the same three figures in plain fp16 (the reference's "fp16" configuration; NOT parity-grade)
  host_to_host_fps           : through the synchronous boundary call (vp_infer_multi: pageable host frame in, H2D, both
                               networks, D2H of logits + masks, one sync -- what TensorRTBackend::doInference brackets,
                               tensorrt_backend.cpp:184-199), one host thread per in-flight engine
  roofline                   : dominant kernel's ALGORITHMIC TFLOP/s from per-launch HIP events on the engine stream vs
                               the dense fp16 MFMA peak (2.5 PFLOP/s), plus the whole-frame fraction
  cpu_baseline               : the CPU oracle (torch fp32 restatement of the reference path) on this box's host cores
  sceneseg_* / autodrive_*   : BASELINE configs[1] (SceneSeg alone, 1280x720, fp16 and the parity mode) and configs[4] (AutoDrive,
                               1920x1080 frames, fp8-stored weights) on the same box, same protocol (N = 1 only)
  gather_fps                 : the metric configuration with the per-frame RCCL all-gather enqueued (world 1): the N = 1 anchor of --gather
p50 / p99 are taken over --latency-iters (default 1000) synchronised iterations after 50 warm-up ones: the reference's protocol
(Models/data_utils/benchmark.py:17-47).
Timing: with ``--steps K`` given (the driver's form) a timed region is EXACTLY K steps, fenced on both sides; the region is repeated until
``--min-seconds`` (1 s) have been measured and the MEDIAN region is reported (``timed_regions`` in the line; a 20-step region is 37 ms, and one such
sample moves by +-2 %).  Without ``--steps``: 300, raised until one region lasts ``--min-seconds``.

Multi-GPU: the path shards by camera: rank r owns camera r, weights replicated, no data-path collective (the
reference has none) -> "scaling": "weak".  ``--gather`` adds the per-frame RCCL all-gather of the per-camera masks through
the C ABI (vp_gather), enqueued on the engine's stream behind the frame's graph with no host synchronisation.
"""
import argparse
import json
import math
import os
import re
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before libvp_hip: one shared HIP runtime)

PEAK_FP16_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
FRAME_GFLOP = {"sceneseg": 367.0, "scene3d": 397.0, "domainseg": 366.6, "egolanes": 196.7}
BACKBONE_GFLOP = 3.136
SEEDS = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
WORKLOADS = {  # name -> network kinds; the first owns the encoder, the others are shared-prefix heads on it
    "seg+3d": ("sceneseg", "scene3d"),
    "seg+3d+ego": ("sceneseg", "scene3d", "egolanes"),
    "sceneseg": ("sceneseg",), "scene3d": ("scene3d",), "domainseg": ("domainseg",), "egolanes": ("egolanes",),
}


def workload_gflop(kinds):
    return sum(FRAME_GFLOP[k] for k in kinds) - BACKBONE_GFLOP * (len(kinds) - 1)


def kernel_source_hash():
    """sha256 over every kernel source and header of the library: tools/pmc_summarize.py stamps the
    PMC file with it, and a file taken on other sources is ignored (roofline.traffic = null) instead of going silently stale."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "autoware_vision_pilot_amd", "csrc")
    for f in sorted(os.listdir(csrc)):          # EVERY kernel source and header (round 4: the counters cover the whole plan)
        if not f.endswith((".hip", ".hpp", ".inc")):
            continue
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(tag):
    """HBM-side bytes per launch of the kernel instantiation behind ``tag`` from the committed rocprofv3 PMC passes
    (profiles/r0N_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of tools/pmc_conv.py, calibrated in-run on a
    1 GiB streaming kernel: FETCH_SIZE x2, WRITE_SIZE x1 on gfx950; tools/pmc_summarize.py).  The file must carry the hash of
    the CURRENT kernel sources (kernel_source_hash); None if not profiled on them."""
    mu = re.match(r"upconv_x3w(8|4)<", tag)
    m8 = re.match(r"conv3x3_x3w(8|4)<co(\d+),px(\d+)", tag)
    m = re.match(r"conv3x3_halo<co(\d+),px(\d+),x(\d)(,regepi)?>", tag)
    if mu:   # composed up-sampling stages: upconv_x3_kernel<CO_TILE, TH, WCO, WPX, HDB, ...>
        sk = "true" if "+splitk" in tag else "false"      # <..., HDB, ACT, SPLITK, X1, ABL>: the split launches are their own instantiation
        pat = (r"upconv_x3_kernel<128, 16, 2, 4, true, \d, " if mu.group(1) == "8" else r"upconv_x3_kernel<128, 8, 2, 2, false, \d, ") + sk
    elif m8:   # the pipelined shapes: <CO_TILE, TH, WCO, WPX, HDB, ...>
        co8, px8 = int(m8.group(2)), int(m8.group(3))
        if m8.group(1) == "8":
            pat = r"conv3x3_x3_kernel<128, 16, 2, 4, true"
        elif px8 == 256:
            pat = rf"conv3x3_x3_kernel<{co8}, 16, 1, 4, false"
        else:
            pat = rf"conv3x3_x3_kernel<{co8}, 8, 2, 2, false"
    elif m:
        co, px, x, reg = int(m.group(1)), int(m.group(2)), m.group(3), m.group(4)
        pat = rf"conv3x3_halo_kernel<{co}, {px // 16}, 16, \d, \d, {'true' if x == '3' else 'false'}, 0, {'true' if reg else 'false'}(, \w+)*>"
    else:
        m = re.match(r"conv_gemm<bk(\d+),co(\d+),px(\d+),x(\d)(?:,regepi(\d))?>", tag)
        if not m:
            # every other kernel family (mbconv_front / _back, conv3x3_map, gemm_dma, head_conv3x3, convt_rs, stem, ...): the tag's template
            # arguments are abbreviations, so the counters are averaged over the family's instantiations
            base = re.split(r"[<+]", tag)[0]
            pat = r"vp::" + re.escape(base) + r"_kernel\b"
        else:
            pat = (rf"conv_gemm_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}, \d, \d, {'true' if m.group(4) == '3' else 'false'}, "
                   rf"\d, (true|false), {m.group(5) or 0}>")
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        doc = json.load(open(path))
        if doc.get("source_hash") != kernel_source_hash():
            continue  # counters taken on OTHER kernel sources say nothing about the kernels this run timed
        ks = doc["kernels"]
        hit = [v for k, v in ks.items() if re.search(pat, k)]
        n = sum(v["launches_seen"] for v in hit)
        if not n:
            continue
        f = sum(v["fetch_bytes"] * v["launches_seen"] for v in hit) / n
        w = sum(v["write_bytes"] * v["launches_seen"] for v in hit) / n
        return {"bytes": round(f + w), "fetch_bytes": round(f), "write_bytes": round(w),
                "source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_conv.py, average over "
                          "that instantiation's launches)"}
    return None


def cpu_baseline(kinds, sds, frame, seconds, nthreads=0):
    """The CPU oracle (oracle/nets.py, torch fp32) on `nthreads` host threads (0 = all, capped at 32): whole frames of the
    same workload for about `seconds`.  Each network runs its own full forward, as the reference's Models/inference classes
    do (one *NetworkInfer object per network, each with its own backbone pass: scene_seg_infer.py:38-55)."""
    from oracle import nets, pre_post

    nthreads = nthreads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(nthreads)
    tsds = [nets.to_torch(sd) for sd in sds]

    def one():
        x = torch.from_numpy(pre_post.preprocess(frame))
        for k, tsd in zip(kinds, tsds):
            y = nets.forward(k, tsd, x)[0].numpy()
            pre_post.seg_mask_u8(y) if k != "egolanes" else pre_post.egolanes_priority_mask(y)

    one()  # warm-up
    n_done, t_cpu = 0, 0.0
    while t_cpu < seconds and n_done < 50:
        t1 = time.perf_counter()
        one()
        t_cpu += time.perf_counter() - t1
        n_done += 1
    return {"value": round(n_done / t_cpu, 4), "unit": "frames/s", "cores": nthreads, "kind": "port",
            "sample": f"{n_done} frames of the same workload ({'+'.join(kinds)} on one 1280x720 frame: preprocess + "
                      f"{len(kinds)} full forwards + decode), torch {torch.__version__} CPU fp32, {t_cpu:.1f} s"}


class Camera:
    """One in-flight frame slot: the encoder-owning engine plus its shared-prefix heads, all on one HIP stream."""
    fork = True   # vp_enqueue_multi (default) vs separate vp_enqueue calls (--no-fork)
    fork_all = False  # --fork-all: forked graphs in the throughput legs too (experiment)

    def __init__(self, lib, kinds, blobs, precision, gpu, frame):
        t0 = time.perf_counter()
        self.base = lib.Engine(kinds[0], blobs[0], precision=precision, gpu_id=gpu)
        self.heads = [lib.Engine(k, b, precision=precision, gpu_id=gpu, base=self.base) for k, b in zip(kinds[1:], blobs[1:])]
        self.create_s = time.perf_counter() - t0   # vp_create + vp_create_shared: blob parse, BN fold, prescale + (hi, lo) split, per-kernel packing, uploads
        self.frame = frame
        # no forked graph (hence no side stream) unless a latency leg asks for it: HIP streams are dealt round-robin onto a few
        # hardware queues (4 by default), and a side stream created between two cameras' streams made two cameras share a queue
        self.base.set_multi_fork(False)
        self.base.upload_frame(frame)  # resident in HBM before any timed region
        for _ in range(2):             # first pass is eager (sets kernel attributes), second captures the graphs
            self.enqueue()
        self.sync()

    def enqueue(self):
        if self.heads:
            self.base.enqueue_multi(self.heads)   # one call; forked inside or one engine after the other (set_fork)
        else:
            self.base.enqueue()

    def set_fork(self, on):
        self.base.set_multi_fork((bool(on) or Camera.fork_all) and Camera.fork)

    def sync(self):
        self.base.sync()

    def host_frame(self):
        self.base.infer_multi(self.heads, self.frame)

    def close(self):
        for h in self.heads:
            h.close()
        self.base.close()


def main():
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints its version banner from C stdio when
    # NCCL_DEBUG is set, at process exit, i.e. AFTER our line): keep the real stdout for the JSON line, point fd 1 at stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps K.  Given: EXACTLY K steps per timed region (the contract), repeated over several such "
                    "regions until --min-seconds have been measured, `value` from the MEDIAN region (`timed_regions` in the line).  Absent: 300, raised so "
                    "that one region lasts --min-seconds")
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="seg+3d", choices=sorted(WORKLOADS))
    ap.add_argument("--kind", default=None, help="deprecated alias: --workload <kind>")
    ap.add_argument("--precision", default="fp16x3", choices=["fp16", "fp16x3"],
                    help="precision of `value`; fp16x3 is the parity mode (default), fp16 is reported beside it as fp16_value")
    ap.add_argument("--frame", default="1280x720")
    ap.add_argument("--streams", type=int, default=3,
                    help="frames in flight per GPU (independent engines / HIP streams, round-robin): the latency-bound "
                         "encoder of frame n+1 overlaps the MFMA-bound decoder of frame n")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="floor of every timed region")
    ap.add_argument("--gather", action="store_true", help="all-gather per-camera masks every step (RCCL through the C ABI)")
    ap.add_argument("--no-fork", action="store_true", help="enqueue the base engine and its heads one after the other (vp_enqueue) "
                    "instead of one forked graph per frame (vp_enqueue_multi)")
    ap.add_argument("--fork-all", action="store_true", help="experiment: forked graphs in the several-cameras legs too")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp16 and host-to-host legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = all host cores, max 32)")
    ap.add_argument("--latency-iters", type=int, default=1000, help="synchronised iterations behind p50 / p99 (benchmark.py:17-47: 50 warm-up + 1000)")
    ap.add_argument("--leg", default=None, choices=["three-heads-one-camera"], help="internal: one leg in a child process (see three_heads_note)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="developer knob of the dispatch rules (vp_set_option; repeatable) -- the library does not read the environment; "
                         "whatever is set shows in the line's `library` and `plan_hash` fields")
    args = ap.parse_args()
    exact_steps = args.steps is not None     # the driver's form (--steps K): K is honoured exactly
    if args.steps is None:
        args.steps = 300
    if args.kind:
        args.workload = args.kind
    Camera.fork = not args.no_fork
    Camera.fork_all = args.fork_all

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: one rank per GPU, launched here (the driver's torch.distributed.run form sets the
        # same variables and comes in through the branch below); rank 0 prints the JSON line on the inherited stdout
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.dup2(json_out.fileno(), 1)
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        raise SystemExit(max(p.wait() for p in procs))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))

    from autoware_vision_pilot_amd import lib, synthetic, weights as vw

    for kv in args.option:
        key, _, val = kv.partition("=")
        lib.set_option(key, val)
    kinds = WORKLOADS[args.workload]
    fw, fh = (int(v) for v in args.frame.split("x"))
    sds = [synthetic.make_state_dict(kinds[0], SEEDS[kinds[0]])]
    for k in kinds[1:]:  # heads share the first network's backbone, as the reference's pre-trained wrappers do
        sds.append(synthetic.share_backbone(synthetic.make_state_dict(k, SEEDS[k]), k, sds[0], kinds[0]))
    blobs = [vw.pack_state_dict(sd) for sd in sds]
    frame = synthetic.synthetic_frame(fh, fw, 10 + rank)  # camera r
    nstreams = max(1, args.streams)

    def fence(cams):
        for c in cams:
            c.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    comms = None

    def timed(cams, steps, gather=False):
        """EXACTLY `steps` frames round-robin over the in-flight slots, fenced on both sides; returns seconds (max over ranks)."""
        n = len(cams)
        fence(cams)
        t0 = time.perf_counter()
        for i in range(steps):
            c = cams[i % n]
            c.enqueue()
            if gather:
                comms[i % n].gather(c.base, lib.VP_GATHER_MASK)  # async on the engine's stream, no host sync
        fence(cams)
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def throughput(cams, steps, warmup, gather=False):
        """(K, seconds of one K-step region).  --steps given: K = steps exactly, the region repeated (each one fenced on both sides, max over ranks) until
        --min-seconds are covered, at most 40 times, and the MEDIAN region reported; otherwise K is raised until one region lasts --min-seconds."""
        timed(cams, max(1, warmup), gather)
        probe = timed(cams, min(max(steps, 1), 20), gather) / min(max(steps, 1), 20)
        if exact_steps:
            k = max(1, steps)
            regions = max(1, min(40, int(math.ceil(args.min_seconds / max(probe * k, 1e-6)))))
        else:
            k, regions = max(steps, int(math.ceil(1.15 * args.min_seconds / max(probe, 1e-6)))), 1
        if dist is not None:  # every rank must run the same number of steps and regions
            t = torch.tensor([k, regions], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            k, regions = int(t[0].item()), int(t[1].item())
        els = sorted(timed(cams, k, gather) for _ in range(regions))
        throughput.regions = regions
        return k, els[len(els) // 2]
    throughput.regions = 1

    def latency(cam, iters):
        lat = []
        for i in range(50 + iters):     # benchmark.py:17-47: 50 warm-up iterations, then `iters` timed ones, a device sync per iteration
            t1 = time.perf_counter()
            cam.enqueue()
            cam.sync()
            if i >= 50:
                lat.append((time.perf_counter() - t1) * 1e3)
        return np.array(lat)

    def three_figures(cams, steps, warmup, gather=False):
        # several cameras in flight: engines one after the other (the chip is full; forked heads only add contention);
        # one camera / one frame at a time: the backbone-only heads forked behind the shared encoder (vp_set_multi_fork)
        for c in cams:
            c.set_fork(False)
        k, el = throughput(cams, steps, warmup, gather)
        main_regions = throughput.regions
        cams[0].set_fork(True)
        k1, el1 = throughput(cams[:1], max(steps // 3, 1), 3)
        lat = latency(cams[0], args.latency_iters)
        cams[0].set_fork(False)
        return dict(steps=k, elapsed=el, regions=main_regions, fps=world * k / el, single=k1 / el1, latency_iters=len(lat),
                    p50=float(np.percentile(lat, 50)), p99=float(np.percentile(lat, 99)))

    def ego_engine():
        e = lib.Engine("egolanes", vw.pack_state_dict(synthetic.make_state_dict("egolanes", SEEDS["egolanes"])), precision=args.precision, gpu_id=local_rank)
        e.set_input_format(lib.VP_BGR8, lib.VP_PLANES_RGB)
        e.upload_frame(frame)
        for _ in range(2):
            e.enqueue()
        e.sync()
        return e

    def three_heads_latency(cam, ego, iters):
        cam.set_fork(True)           # one camera, one frame at a time: Scene3D forked behind the encoder, EgoLanes on its own stream
        lat3 = []
        for i in range(50 + iters):
            t1 = time.perf_counter()
            cam.enqueue()
            ego.enqueue()
            cam.sync()
            ego.sync()
            if i >= 50:
                lat3.append((time.perf_counter() - t1) * 1e3)
        cam.set_fork(False)
        return lat3

    if args.leg == "three-heads-one-camera":
        # child process of the default run: ONE camera's engines alone in a process (SceneSeg + Scene3D on a shared encoder, EgoLanes on its own
        # backbone) -- what a one-camera node holds.  The HIP runtime shares its few hardware queues among a process's streams: with the other
        # in-flight cameras' streams alive, this camera's two base engines can land on ONE queue and run one after the other (4.2 ms against 3.6)
        cam = Camera(lib, kinds, blobs, args.precision, local_rank, frame)
        ego = ego_engine()
        lat3 = three_heads_latency(cam, ego, max(20, args.latency_iters))
        json_out.write(json.dumps({"p50_ms": round(float(np.percentile(lat3, 50)), 4), "p99_ms": round(float(np.percentile(lat3, 99)), 4)}) + "\n")
        json_out.flush()     # (fd 1 itself points at stderr for the run: native libraries print there)
        return

    # ---- the reported configuration
    cams = [Camera(lib, kinds, blobs, args.precision, local_rank, frame) for _ in range(nstreams)]

    def make_comms():
        rec = 320 * 640 if kinds[0] != "egolanes" else 80 * 160
        cs = []
        for i in range(nstreams):  # one communicator per engine in flight (RCCL ops of one comm must not overlap)
            ids = [lib.Comm.unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(ids, src=0)
            cs.append(lib.Comm(ids[0], rank, world, local_rank, rec))
        return cs

    if args.gather:
        comms = make_comms()
    main_fig = three_figures(cams, args.steps, args.warmup, args.gather)

    # ---- the N = 1 anchor of the multi-camera exchange: the same throughput leg with the per-frame RCCL all-gather of the
    # class maps enqueued behind every frame (vp_gather, world 1), so the first 8-GPU `--gather` run has its reference point from this command
    gather_fig = None
    if world == 1 and not args.gather and not args.no_secondary and args.leg is None:
        try:
            comms = make_comms()
            for c in cams:
                c.set_fork(False)
            kg, elg = throughput(cams, args.steps, max(3, args.warmup // 3), True)
            gather_fig = {"gather_fps": round(kg / elg, 2),
                          "gather_note": "the `value` leg with vp_gather (ncclAllGather of the 320x640 class map, world 1) enqueued on the engine's stream behind every "
                                         "frame, no host synchronisation: what `--gather` measures at N > 1"}
        except Exception as ex:  # noqa: BLE001 -- RCCL absent / broken on this box: reported, not hidden
            gather_fig = {"gather_fps": None, "gather_note": f"gather leg failed: {ex!r}"}
        finally:
            if comms:
                for cm in comms:
                    cm.close()
            comms = None

    # ---- the `value` leg on OTHER DATA: the matrix pipe is power- and data-toggle-limited (profiles/r03_mfma_power.txt: all-zero
    # operands 2.47 PFLOP/s, random ones 1.74), so the frame rate depends on the bits that go through it.  Same protocol, same kernels; the "trained-like"
    # weight family of the parity sweep (synthetic.make_trained_like_state_dict: BatchNorm scales spread over 10^3 per layer, small variances, sparse
    # large weights) and a DIFFERENT synthetic frame in every in-flight slot (the default leg carries one frame in all of them).
    trained_fig = None
    if world == 1 and not args.no_secondary and args.leg is None and args.workload == "seg+3d":
        sds_t = [synthetic.make_trained_like_state_dict(kinds[0], SEEDS[kinds[0]])]
        for k in kinds[1:]:
            sds_t.append(synthetic.share_backbone(synthetic.make_trained_like_state_dict(k, SEEDS[k]), k, sds_t[0], kinds[0]))
        blobs_t = [vw.pack_state_dict(sd) for sd in sds_t]
        cams_t = [Camera(lib, kinds, blobs_t, args.precision, local_rank, synthetic.synthetic_frame(fh, fw, 40 + 7 * i, smooth=(i != 1))) for i in range(nstreams)]
        for c in cams_t:
            c.set_fork(False)
        kt, elt = throughput(cams_t, args.steps, max(3, args.warmup // 3))
        for c in cams_t:
            c.close()
        # ... and the default weights with three different frames (separates the weights' share from the frames')
        cams_f = [Camera(lib, kinds, blobs, args.precision, local_rank, synthetic.synthetic_frame(fh, fw, 40 + 7 * i, smooth=(i != 1))) for i in range(nstreams)]
        for c in cams_f:
            c.set_fork(False)
        kf, elf = throughput(cams_f, args.steps, max(3, args.warmup // 3))
        for c in cams_f:
            c.close()
        trained_fig = {"value_trained_like": round(kt / elt, 2), "value_three_frames": round(kf / elf, 2),
                       "value_data_note": "the `value` leg on other operands (the matrix pipe's sustained clock depends on the bits it multiplies): value_trained_like = "
                                          "trained-like weight family (BatchNorm scales spread over 10^3 per layer, sparse large weights) + a different frame in each "
                                          "in-flight slot (one of them unsmoothed noise); value_three_frames = the default seeded weights with those three frames; "
                                          "`value` = default weights, one frame in all slots"}

    # ---- FpsTimer-style split of ONE camera's frame (the reference nodes' benchmark: common/benchmark/fps_timer.cpp:37-63, stamps at
    # run_model_node.cpp:66,77,107/180,115/188): wall-clock stamps at the stage boundaries of a synchronous loop, medians.  The stages here
    # are the device-side ones a frame goes through between the host buffers (the nodes' own "preprocess" stamp brackets a cv_bridge copy):
    #   preprocess = pageable frame -> pinned staging -> H2D (vp_upload_frame) + the resize / normalise kernel
    #   inference  = encoder + every network's context / neck / head + fused decode (the graph minus the preprocess launch)
    #   output     = D2H of every network's fp32 logits + u8 mask (vp_fetch_outputs) + the nodes' resizes to the frame size on the device
    #                (mask nearest, run_model_node.cpp:176-177; depth bilinear, :104) with their D2H
    fps_timer = None

    def node_outputs(c, on):
        """on: what the adapters copy per frame since round 5 (HipBackend: the class map; the depth consumer: its fp32 map) -- off: every tensor."""
        c.base.set_outputs(logits=not on, mask=True)
        for h in c.heads:
            h.set_outputs(logits=True, mask=(not on) and h.kind != "scene3d")

    def fps_split(cam, registered):
        """FpsTimer-style medians (us) of one camera's synchronous loop: (upload, graph, output)."""
        st = {"up": [], "net": [], "out": []}
        for i in range(10 + max(30, args.latency_iters // 2)):
            t0 = time.perf_counter()
            cam.base.upload_frame(frame)
            cam.sync()
            t1 = time.perf_counter()
            cam.enqueue()
            cam.sync()
            t2 = time.perf_counter()
            for e in [cam.base] + cam.heads:
                if not registered:
                    e.fetch_outputs()
                if e.kind == "scene3d":
                    e.depth_resized(fh, fw)
                else:
                    e.mask_resized(fh, fw)
            t3 = time.perf_counter()
            if i >= 10:
                st["up"].append(1e6 * (t1 - t0))
                st["net"].append(1e6 * (t2 - t1))
                st["out"].append(1e6 * (t3 - t2))
        return tuple(float(np.median(st[k])) for k in ("up", "net", "out"))

    if rank == 0 and not args.no_secondary:
        cam = cams[0]
        cam.set_fork(True)
        pre_kernel_us = 1e3 * float(cam.base.profile_layers(20)[0])          # launch 0 of the plan = the preprocess kernel (HIP events)
        up0, net0, out0 = fps_split(cam, False)       # as up to round 4: pageable frame staged through the engine's pinned buffer, every tensor to the host
        lib.register_frames(frame)                     # round 5: the node's frame pool page-locked once (vp_register_frames) ...
        try:
            up, net, outp = fps_split(cam, True)       # ... and only the frame-size outputs the node publishes leave the device
        finally:
            lib.unregister_frames(frame)
        cam.set_fork(False)
        fps_timer = {"preprocess_us": round(up + pre_kernel_us, 1), "inference_us": round(net - pre_kernel_us, 1), "output_us": round(outp, 1),
                     "total_us": round(up + net + outp, 1), "upload_us": round(up, 1), "preprocess_kernel_us": round(pre_kernel_us, 1),
                     "all_outputs_pageable": {"preprocess_us": round(up0 + pre_kernel_us, 1), "inference_us": round(net0 - pre_kernel_us, 1),
                                              "output_us": round(out0, 1), "total_us": round(up0 + net0 + out0, 1), "upload_us": round(up0, 1)},
                     "note": "FpsTimer-style (fps_timer.cpp:37-63) medians of one camera, one frame at a time, a host sync at every stage boundary: "
                             "preprocess = frame in a pool registered with vp_register_frames -> ONE DMA + resize / normalise kernel; inference = shared "
                             "encoder + all decoders + fused decode (heads forked); output = what the patched node publishes: the class map resized to the "
                             "frame size (nearest) and the depth map resized (bilinear) on the device, D2H of those two only.  all_outputs_pageable = "
                             "round 4's definition (pageable frame -> pinned staging memcpy -> H2D; D2H of every fp32 logit tensor + mask, then the resizes)"}

    # ---- host-to-host through the synchronous boundary call, one host thread per in-flight engine
    h2h = None
    if not args.no_secondary:
        def host_loop(cam, seconds, out, idx):
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                cam.host_frame()
                n += 1
            out[idx] = (n, time.perf_counter() - t0)

        def host_run(seconds):
            res = [None] * len(cams)
            ts = [threading.Thread(target=host_loop, args=(c, seconds, res, i)) for i, c in enumerate(cams)]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            wall = time.perf_counter() - t0
            return sum(r[0] for r in res) / wall

        # round 5: the adapters' defaults -- the frame pool registered (no staging memcpy), the class map + the depth map to the host, fp32 logits
        # of the segmentation network left in HBM until someone asks (HipBackend::getRawTensorData fetches on demand)
        lib.register_frames(frame)
        try:
            for c in cams:
                c.set_fork(False)
                node_outputs(c, True)
                c.host_frame()
            host_run(0.2)
            h2h = {"fps": host_run(max(args.min_seconds, 1.0))}
            cams[0].set_fork(True)
            lat = []
            for i in range(10 + max(20, args.latency_iters // 2)):
                t1 = time.perf_counter()
                cams[0].host_frame()
                if i >= 10:
                    lat.append((time.perf_counter() - t1) * 1e3)
            h2h["p50"] = float(np.percentile(lat, 50))
            cams[0].set_fork(False)
        finally:
            lib.unregister_frames(frame)
        for c in cams:       # round 4's definition beside it: pageable frame staged through the engine's pinned buffer, every tensor to the host
            node_outputs(c, False)
            c.host_frame()
        h2h["fps_all"] = host_run(max(args.min_seconds, 1.0))
        cams[0].set_fork(True)
        lat = []
        for i in range(10 + max(20, args.latency_iters // 4)):
            t1 = time.perf_counter()
            cams[0].host_frame()
            if i >= 10:
                lat.append((time.perf_counter() - t1) * 1e3)
        h2h["p50_all"] = float(np.percentile(lat, 50))
        cams[0].set_fork(False)
        for c in cams:
            c.base.set_outputs(True, True)
            for h in c.heads:
                h.set_outputs(True, True)

    # ---- north_star's target configuration beside the metric's: SceneSeg + Scene3D (shared encoder) + EgoLanes on one camera.
    # EgoLanes owns its backbone (ego_lanes_network.py:14-15 builds its own Backbone; its checkpoint carries other encoder weights) and
    # is fed RGB planes (onnxruntime_engine.cpp:80-100): a SECOND base engine with its own weights, stream and preprocess, not a head
    # grafted onto SceneSeg's encoder.  957.6 GFLOP per frame.  N = 1 only.
    three = None
    if args.workload == "seg+3d" and not args.no_secondary and world == 1:
        egos = [ego_engine() for _ in range(nstreams)]

        def run3(slots, steps):
            for c, e in slots:
                c.sync()
                e.sync()
            t0 = time.perf_counter()
            for i in range(steps):
                c, e = slots[i % len(slots)]
                c.enqueue()
                e.enqueue()
            for c, e in slots:
                c.sync()
                e.sync()
            return time.perf_counter() - t0

        slots = list(zip(cams, egos))
        lat3 = three_heads_latency(cams[0], egos[0], max(20, args.latency_iters))      # (while every engine still has its own stream)
        crowded_p50 = round(float(np.percentile(lat3, 50)), 4)
        for c in cams:
            c.set_fork(False)
        run3(slots, 30)
        k3 = max(60, int(math.ceil(1.15 * args.min_seconds / (run3(slots, 20) / 20))))
        el3 = run3(slots, k3)
        # (one stream per camera -- the EgoLanes engine moved onto its camera's SceneSeg + Scene3D stream -- was built and measured: 302 / 290 frames/s
        # against 308 / 293 with every base engine on its own stream; not kept)
        alone = None
        try:   # the same measurement in a process that holds ONE camera's engines (see the leg above)
            import subprocess
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", "three-heads-one-camera", "--precision", args.precision, "--frame", args.frame,
                                "--latency-iters", str(args.latency_iters)] + [x for kv in args.option for x in ("--option", kv)],
                               capture_output=True, text=True, timeout=300)
            if r.returncode != 0 or not r.stdout.strip():
                raise RuntimeError(f"rc {r.returncode}: {r.stderr.strip().splitlines()[-3:]}")
            alone = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as ex:  # noqa: BLE001 -- the crowded figure stays
            print(f"bench: the one-camera child leg failed ({ex!r}); three_heads_p50_ms_one_camera_process is null", file=sys.stderr)
            alone = None
        g3 = workload_gflop(kinds) + FRAME_GFLOP["egolanes"]
        three = {"three_heads_fps": round(k3 / el3, 2),
                 # one key, one meaning.  three_heads_p50_ms = measured IN THIS PROCESS, beside the other in-flight cameras' streams
                 # (its meaning up to round 3); _one_camera_process = a child process that holds ONE camera's engines (round 4's headline, null if the
                 # child failed -- never silently substituted)
                 "three_heads_p50_ms": crowded_p50,
                 "three_heads_p50_ms_one_camera_process": alone["p50_ms"] if alone else None,
                 "three_heads_p99_ms_one_camera_process": alone["p99_ms"] if alone else None,
                 "three_heads_p50_source": "in-process beside the other cameras' streams; _one_camera_process: child process" + ("" if alone else " (FAILED: null)"),
                 "three_heads_gflop_per_frame": round(g3, 1),
                 "three_heads_whole_frame_frac": round(g3 * (k3 / el3) / 1e3 / PEAK_FP16_TFLOPS, 4),
                 "three_heads_note": ("BASELINE configs[2] / north_star target: SceneSeg + Scene3D on a shared encoder + EgoLanes on ITS OWN "
                                      "backbone weights and RGB-plane preprocess (a second base engine on its own stream), one 1280x720 "
                                      f"camera, {args.precision}; fps with {nstreams} cameras in flight; p50 one frame at a time IN THIS PROCESS, where "
                                      "the HIP runtime's few hardware queues are shared with the other cameras' streams and the two base engines "
                                      "of a camera can end up on one queue; _one_camera_process = the same in a child process that holds that ONE "
                                      "camera's engines (what a one-camera node sees)")}
        for e in egos:
            e.close()

    # ---- roofline of the dominant kernel family of the REPORTED precision: per-launch HIP events (eager replay, one stream)
    out = None
    if rank == 0:
        engs = [cams[0].base] + cams[0].heads
        rows = []
        for e in engs:
            ms = e.profile_layers(10)
            rows += list(zip(e.layers(), e.layer_kernels(), [float(t) for t in ms], e.layer_flops_executed()))
        fam = {}
        for (name, fl, by), k, t, fx in rows:
            f = fam.setdefault(k, dict(ms=0.0, flops=0.0, flops_exec=0.0, bytes=0.0, n=0, worst=("", 0.0)))
            f["ms"] += t
            f["flops"] += fl
            f["flops_exec"] += fx
            f["bytes"] += by
            f["n"] += 1
            if t > f["worst"][1]:
                f["worst"] = (name, t)
        tot_ms = sum(t for _, _, t, _ in rows)
        # what one frame EXECUTES on the matrix pipe (round 6: the composed up-sampling stages run 0.40-0.51x of the reference formulation's count)
        exec_gflop = sum(fx for _, _, _, fx in rows) / 1e9
        # The path is a dense contraction (bound = MFMA): the roofline kernel is the single instantiation
        # (= one rocprofv3 kernel name) with the largest total time among those carrying >= 5 % of the frame's FLOPs;
        # "+splitk" ops are two launches per timing interval and cannot give a per-kernel duration.
        single = [k for k in fam if "+splitk" not in k]
        tot_fl = sum(f["flops"] for f in fam.values())
        heavy = [k for k in single if fam[k]["flops"] >= 0.05 * tot_fl]
        dom = max(heavy or single, key=lambda k: fam[k]["ms"])
        d = fam[dom]

        def frac_of(k):
            f = fam[k]
            if k.startswith("conv") or k.startswith("upconv"):
                return "mfma", f["flops"] / (f["ms"] * 1e-3) / 1e12 / PEAK_FP16_TFLOPS
            return "hbm", f["bytes"] / (f["ms"] * 1e-3) / 1e9 / 8000.0

        by_time = []
        # every kernel family by its share of the frame's (eager, single-stream) time; "+splitk" families are TWO launches per timing interval (the
        # kernel + the shared finish kernel): their `frac` includes the finish launch, and `launches` counts intervals
        for k in sorted(fam, key=lambda k: -fam[k]["ms"])[:10]:
            trk = pmc_traffic(k)
            by_time.append({"kernel": k, "launches": fam[k]["n"], "time_share": round(fam[k]["ms"] / tot_ms, 3),
                            "bound": frac_of(k)[0], "frac": round(frac_of(k)[1], 4),
                            **({"frac_executed": round(fam[k]["flops_exec"] / (fam[k]["ms"] * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, 4)} if frac_of(k)[0] == "mfma" else {}),
                            "algorithmic_mb_per_launch": round(fam[k]["bytes"] / fam[k]["n"] / 1e6, 2),
                            "traffic_mb_per_launch": round(trk["bytes"] / 1e6, 2) if trk else None,
                            **({"includes_finish_launch": True} if "+splitk" in k else {})})
        gflop = workload_gflop(kinds)
        frame_tflops = gflop * (main_fig["fps"] / world) / 1e3
        if dom.startswith("conv") or dom.startswith("upconv"):
            achieved, peak, unit, bound = d["flops"] / (d["ms"] * 1e-3) / 1e12, PEAK_FP16_TFLOPS, "TFLOP/s", "mfma"
        else:
            achieved, peak, unit, bound = d["bytes"] / (d["ms"] * 1e-3) / 1e9, 8000.0, "GB/s", "hbm"
        tr = pmc_traffic(dom)
        mfma_per_product = 3 if args.precision == "fp16x3" else 1
        roofline = {
            "bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "traffic": (tr or {}).get("bytes"), "traffic_detail": tr,
            "kernel": dom, "launches_per_frame": d["n"], "avg_launch_us": round(1e3 * d["ms"] / d["n"], 2),
            "algorithmic_gflop_per_launch": round(d["flops"] / d["n"] / 1e9, 3),
            "algorithmic_mb_per_launch": round(d["bytes"] / d["n"] / 1e6, 3),
            # executed: what the launches of that instantiation put on the matrix pipe (a composed up-sampling stage: 4 taps of the low-resolution tensor +
            # 9 of the skip tensor per output pixel instead of the reference formulation's ConvTranspose + 1x1 + 3x3); mfma_issue_frac is on EXECUTED work
            "executed_gflop_per_launch": round(d["flops_exec"] / d["n"] / 1e9, 3),
            "achieved_executed": round(d["flops_exec"] / (d["ms"] * 1e-3) / 1e12, 2) if bound == "mfma" else None,
            "mfma_issue_frac": round(mfma_per_product * d["flops_exec"] / (d["ms"] * 1e-3) / 1e12 / peak, 4) if bound == "mfma" else None,
            "slowest_layer": d["worst"][0], "slowest_layer_us": round(1e3 * d["worst"][1], 1),
            "kernel_time_share": round(d["ms"] / tot_ms, 3), "by_time": by_time,
            # every launch of that instantiation: a family average hides that a launch's rate follows its workgroup count (decode_layer_5: 100 workgroups of the
            # 8-wave shape on 100 of 256 CUs -- chosen for its CU-time, DESIGN.md section 3 -- against 200 / 400 for its siblings)
            "layers": [{"layer": name, "us": round(1e3 * t, 1), "tflops": round(fl / (t * 1e-3) / 1e12, 1), "frac": round(fl / (t * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, 4),
                        "tflops_executed": round(fx / (t * 1e-3) / 1e12, 1)}
                       for (name, fl, by), k, t, fx in rows if k == dom],
            "whole_frame": {"achieved": round(frame_tflops, 2), "frac": round(frame_tflops / PEAK_FP16_TFLOPS, 4),
                            # issued MFMA work / peak: on what the frame EXECUTES since round 6 (gflop_per_frame stays SURVEY.md 8d's reference-formulation count,
                            # so `frac` is comparable across rounds)
                            "mfma_issue_frac": round(mfma_per_product * exec_gflop * (main_fig["fps"] / world) / 1e3 / PEAK_FP16_TFLOPS, 4),
                            "gflop_per_frame": round(gflop, 1), "executed_gflop_per_frame": round(exec_gflop, 1), "unit": "TFLOP/s"},
            "note": "per-launch HIP events on the engine stream (eager replay, single stream); `achieved` counts ALGORITHMIC "
                    "FLOPs of the reference formulation (SURVEY.md 8d: one multiply-add per product of the reference's operators); a composed up-sampling "
                    "stage computes its three reference operators with 0.40-0.51x the products (`*_executed`); fp16x3 issues 3 fp16 MFMAs per executed "
                    "product, `mfma_issue_frac` = issued MFMA FLOPs / peak",
        }
        prec_name = {"fp16": "fp16", "fp16x3": "fp16x3 (hi+lo fp16 pairs on the fp16 MFMA pipe, fp32 accumulate; fp32-class: floats within 1e-3 "
                                               "of the fp32 oracle, 0 class flips outside that float tolerance -- tie flips only, counted "
                                               "per pass in profiles/r06_parity_sweep.tsv)"}[args.precision]
        names = {"sceneseg": "SceneSeg", "scene3d": "Scene3D", "egolanes": "EgoLanes", "domainseg": "DomainSeg"}
        wl = "+".join(names[k] for k in kinds)
        out = {
            "metric": f"frames/sec ({wl} {fw}x{fh}, one camera per GPU: preprocess + shared encoder + "
                      f"{len(kinds)} decoder(s) + decode per frame); p50 per-frame latency beside it",
            "value": round(main_fig["fps"], 2), "unit": "frames/s", "n_gpus": world, "steps": main_fig["steps"],
            "steps_requested": args.steps, "timed_regions": main_fig["regions"], "warmup": args.warmup,
            "ms_per_step": round(1e3 * main_fig["elapsed"] / main_fig["steps"], 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": prec_name, "data": "synthetic",
            "config": {"workload": f"BASELINE.json metric configuration: {wl} on one {fw}x{fh} camera per GPU, batch 1, "
                                   f"shared EfficientNet-B0 encoder (vp_create_shared), {args.precision}",
                       "precision": args.precision, "parity_mode": args.precision == "fp16x3",
                       "frames_in_flight_per_gpu": nstreams, "net_input": "1x3x320x640", "gather": bool(args.gather),
                       "gflop_per_frame": round(gflop, 1), "executed_gflop_per_frame": round(exec_gflop, 1), "timed_region_s": round(main_fig["elapsed"], 3)},
            "fps_per_gpu": round(main_fig["fps"] / world, 2),
            "single_stream_fps": round(main_fig["single"], 2),
            "p50_ms": round(main_fig["p50"], 4), "p99_ms": round(main_fig["p99"], 4), "latency_iters": main_fig["latency_iters"],
            "modes_note": ("`value` / host_to_host_fps: several cameras in flight, every camera's engines enqueued one after the "
                           "other (vp_set_multi_fork 0); single_stream_fps / p50_ms / p99_ms / host_to_host_p50_ms: ONE camera, one "
                           "frame at a time, the backbone-only heads forked behind the shared encoder inside one graph "
                           "(vp_enqueue_multi / vp_infer_multi, the default) -- same kernels, bit-identical results"
                           + ("" if Camera.fork else "; --no-fork: forking disabled everywhere")),
            "roofline": roofline,
            "rccl_world": world if (args.gather or world > 1) else 0,
            "rccl_use": ("per-frame all-gather of the class maps (vp_gather) + " if args.gather else "") + ("barrier / max-over-ranks timing" if world > 1 else ("none" if not args.gather else "world 1")),
            "library": lib.version(),
            "engine_create_s": round(float(np.median([c.create_s for c in cams])), 3),
            "engine_create_note": "wall time of vp_create (SceneSeg) + vp_create_shared (Scene3D) from the in-memory VPW1 blobs: the packing a TensorRT-style "
                                  "engine cache would save (tensorrt_backend.cpp:40-54); median over the in-flight slots",
            "plan_hash": {e.kind: f"{e.plan_hash():016x}" for e in engs},
        }
        if fps_timer is not None:
            out["fps_timer"] = fps_timer
        if three is not None:
            out.update(three)
        if gather_fig is not None:
            out.update(gather_fig)
        if trained_fig is not None:
            out.update(trained_fig)
        if h2h is not None:
            out["host_to_host_fps"] = round(h2h["fps"], 2)
            out["host_to_host_p50_ms"] = round(h2h["p50"], 4)
            out["host_to_host_all_outputs_fps"] = round(h2h["fps_all"], 2)
            out["host_to_host_all_outputs_p50_ms"] = round(h2h["p50_all"], 4)
            out["host_to_host_note"] = (f"vp_infer_multi per frame from {nstreams} host threads (one per in-flight engine), as the adapters call it since round 5: "
                                        f"{fw}x{fh}x3 frame in a pool registered with vp_register_frames -> ONE DMA (no staging copy), all networks, D2H of "
                                        "SceneSeg's u8 class map + Scene3D's fp32 depth map, one sync (the segmentation logits stay in HBM until "
                                        "getRawTensorData asks); p50 = one thread alone (heads forked).  _all_outputs_ = round 4's definition: pageable frame "
                                        "-> pinned staging memcpy -> H2D, D2H of every network's fp32 logits + u8 mask")
    for c in cams:
        c.close()
    if comms:
        for cm in comms:
            cm.close()

    # ---- the other precision beside it (fp16 when value is the parity mode, and vice versa)
    if not args.no_secondary:
        other = "fp16" if args.precision == "fp16x3" else "fp16x3"
        cams2 = [Camera(lib, kinds, blobs, other, local_rank, frame) for _ in range(nstreams)]
        fig2 = three_figures(cams2, args.steps, max(3, args.warmup // 3))
        for c in cams2:
            c.close()
        if rank == 0:
            tag = "fp16" if other == "fp16" else "fp16x3"
            out[f"{tag}_value"] = round(fig2["fps"], 2)
            out[f"{tag}_single_stream_fps"] = round(fig2["single"], 2)
            out[f"{tag}_p50_ms"] = round(fig2["p50"], 4)
            out[f"{tag}_note"] = ("plain fp16 tensors / one MFMA per product: the reference's 'fp16' configuration; ~2e-2 max error "
                                  "vs the fp32 oracle, NOT parity-grade" if other == "fp16" else "parity mode beside an fp16 headline")

    # ---- BASELINE configs[1]: SceneSeg ALONE on the 1280x720 camera, batch 1 -- "fp16" as BASELINE.json words it and the parity mode beside it;
    # same three figures, same protocol (N = 1 only: the other configurations are not part of the scaling curve)
    if not args.no_secondary and world == 1 and args.workload == "seg+3d":
        sd1, blob1 = [sds[0]], [blobs[0]]
        for prec, tag in (("fp16", "sceneseg_fp16"), ("fp16x3", "sceneseg_fp16x3")):
            cs = [Camera(lib, ("sceneseg",), blob1, prec, local_rank, frame) for _ in range(nstreams)]
            f1 = three_figures(cs, args.steps, max(3, args.warmup // 3))
            for c in cs:
                c.close()
            out[f"{tag}_fps"] = round(f1["fps"], 2)
            out[f"{tag}_single_stream_fps"] = round(f1["single"], 2)
            out[f"{tag}_p50_ms"] = round(f1["p50"], 4)
            out[f"{tag}_p99_ms"] = round(f1["p99"], 4)
            out[f"{tag}_whole_frame_frac"] = round(FRAME_GFLOP["sceneseg"] * f1["fps"] / 1e3 / PEAK_FP16_TFLOPS, 4)
        out["sceneseg_fps"], out["sceneseg_p50_ms"] = out["sceneseg_fp16_fps"], out["sceneseg_fp16_p50_ms"]
        out["sceneseg_note"] = ("BASELINE configs[1]: SceneSeg alone, 1280x720, batch 1, one MI355X; sceneseg_fps / sceneseg_p50_ms = the fp16 engines (configs[1] says "
                                f"fp16; NOT parity-grade), sceneseg_fp16x3_* = the parity mode; fps with {nstreams} cameras in flight, p50 / p99 one frame at a time over "
                                f"{f1['latency_iters']} synchronised iterations; 367.0 GFLOP per frame")

        # ---- BASELINE configs[4]: AutoDrive (model_library), 1920x1080 frames, fp8 (e4m3) STORED weights, streaming: one backbone pass + head per frame
        # (the previous frame's features are kept on the device).  tools/bench_autodrive.py is the stand-alone form of this leg.
        ad_blob = vw.pack_state_dict(synthetic.make_autodrive_state_dict(5))
        ad_frame = synthetic.synthetic_frame(1080, 1920, 21)
        for prec, tag in (("fp16", "autodrive"), ("fp16x3", "autodrive_fp16x3")):
            engs_ad = [lib.Engine("autodrive", ad_blob, precision=prec, weights_fp8=True, gpu_id=local_rank) for _ in range(nstreams)]
            for e in engs_ad:
                e.upload_frame(ad_frame)
            for i in range(60):
                engs_ad[i % nstreams].enqueue()
            for e in engs_ad:
                e.sync()
            nst = 3000
            t0 = time.perf_counter()
            for i in range(nst):
                engs_ad[i % nstreams].enqueue()
            for e in engs_ad:
                e.sync()
            el_ad = time.perf_counter() - t0
            lat = []
            for i in range(50 + args.latency_iters):
                t1 = time.perf_counter()
                engs_ad[0].enqueue()
                engs_ad[0].sync()
                if i >= 50:
                    lat.append((time.perf_counter() - t1) * 1e3)
            out[f"{tag}_fps"] = round(nst / el_ad, 1)
            out[f"{tag}_p50_ms"] = round(float(np.percentile(lat, 50)), 4)
            out[f"{tag}_p99_ms"] = round(float(np.percentile(lat, 99)), 4)
            if tag == "autodrive":
                out["autodrive_weight_bytes"] = engs_ad[0].weight_bytes()
                out["autodrive_launches_per_frame"] = len(engs_ad[0].layers())
            for e in engs_ad:
                e.close()
        out["autodrive_note"] = ("BASELINE configs[4]: AutoDrive on 1920x1080 frames (resized to 1024x512 on the device), weights RESIDENT AS e4m3 bytes + per-row fp32 "
                                 "scales (autodrive_weight_bytes), converted in the weight-staging path; autodrive_* = fp16 arithmetic, autodrive_fp16x3_* = the parity "
                                 f"mode on the same stored weights; fps with {nstreams} frames in flight, p50 / p99 one frame at a time; ~8.1 GFLOP per frame")

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kinds, sds, frame, args.cpu_seconds, args.cpu_threads)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()


if __name__ == "__main__":
    main()
