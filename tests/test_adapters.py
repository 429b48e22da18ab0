"""The C++ drop-in adapters (adapters/hip_backend.hpp, adapters/egolanes_hip_engine.hpp) compiled against stand-in
OpenCV / reference headers: construction-failure conventions on CPU, and on a GPU one frame driven exactly as
RunModelNode::onImage (run_model_node.cpp:79-104,177) / lateralInferenceThread (main.cpp:505-517) drive the
reference backends, compared with the C-ABI results obtained through the ctypes binding."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "adapters", "test", "adapter_check")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "adapters")], check=True, capture_output=True)


def test_adapter_error_conventions():
    _build()
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "conventions OK" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mtype,kind", [("segmentation", "sceneseg"), ("depth", "scene3d"), ("egolanes", "egolanes")])
def test_adapter_frame_matches_c_abi(tmp_path, state_dicts, vp_opts, mtype, kind):
    from autoware_vision_pilot_amd import lib, weights as vw
    from oracle import pre_post

    _build()
    blob = tmp_path / f"{kind}.vpw"
    blob.write_bytes(vw.pack_state_dict(state_dicts(kind)))
    out = tmp_path / "out.bin"
    r = subprocess.run([BIN, mtype, str(blob), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    frame = raw[:720 * 1280 * 3].reshape(720, 1280, 3)
    rest = raw[720 * 1280 * 3:]
    # the B1 adapter creates its engine with the latency plan (one backend = one network on one camera: hip_backend.hpp; a creation flag since round 6),
    # the B2 adapter with the default plan (the production app runs it beside other engines); the same plan here, so the comparison stays bit for bit
    # (the plans differ in fp32 summation order only)
    eng = lib.Engine(kind, blob.read_bytes(), precision="fp16x3", plan_latency=(kind != "egolanes"))
    eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_RGB if kind == "egolanes" else lib.VP_PLANES_BGR)
    eng.set_norm_form(lib.VP_NORM_OPENCV)   # the adapters compute the C++ front-ends' q * fl(1/255) (hip_backend.hpp, egolanes_hip_engine.hpp)
    eng.infer(frame)
    assert np.array_equal(eng.input_tensor(), pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=(kind == "egolanes"), norm_form="opencv"))
    lg = eng.logits()
    n = lg.size * 4
    assert np.array_equal(rest[:n].view(np.float32).reshape(lg.shape), lg)
    tail = rest[n:]
    if kind == "sceneseg":
        assert np.array_equal(tail[:720 * 1280].reshape(720, 1280), pre_post.resize_nearest_u8(pre_post.seg_mask_u8(lg), 720, 1280))
        # MasksVisualizationKernels::createMaskFromTensorHIP (reference signature) on the backend's own tensor
        assert np.array_equal(tail[720 * 1280:].reshape(320, 640), pre_post.seg_mask_u8(lg))
    elif kind == "scene3d":
        assert np.array_equal(tail.view(np.float32).reshape(720, 1280), pre_post.resize_bilinear_f32(lg[0], 720, 1280))
    else:
        assert np.array_equal(tail.view(np.float32).reshape(80, 160), pre_post.egolanes_planes(lg)[0])
    eng.close()


@pytest.mark.gpu
def test_adapters_constructed_concurrently_keep_their_own_plan(tmp_path, state_dicts):
    """a B1 backend and a B2 engine constructed concurrently from two threads of one process -- each on its own kernel plan
    (creation flag VP_PLAN_LATENCY / default), no process-wide option touched (adapter_check 'threads' checks the plan hashes against plain vp_create)."""
    from autoware_vision_pilot_amd import weights as vw

    _build()
    seg, ego, out = tmp_path / "seg.vpw", tmp_path / "ego.vpw", tmp_path / "out.bin"
    seg.write_bytes(vw.pack_state_dict(state_dicts("sceneseg")))
    ego.write_bytes(vw.pack_state_dict(state_dicts("egolanes")))
    r = subprocess.run([BIN, "threads", str(seg), str(out), str(ego)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    h = np.fromfile(out, dtype=np.uint64)
    assert len(h) >= 2 and h[-1] != h[-2] and h[-1] != 0


@pytest.mark.gpu
def test_multicam_host_cpp(tmp_path, state_dicts):
    """adapters/test/multicam_host.cpp: the C++ multi-camera host (thread per GPU, no Python / torch in the process) -- SceneSeg +
    Scene3D on a shared encoder per camera (vp_create + vp_create_shared + vp_enqueue_multi), per-frame RCCL all-gather of the
    class maps behind the C ABI (vp_comm_*, vp_gather), world = min(visible GPUs, cameras).  Its own checks (every rank's gathered
    buffer identical, each record the rank's own map) plus: every camera's gathered class map equals what the ctypes engine
    computes from the frame the host dumped."""
    import json

    from autoware_vision_pilot_amd import lib, synthetic, weights as vw
    from oracle import weights

    _build()
    sd_seg = state_dicts("sceneseg")
    sd_3d = weights.share_backbone(dict(state_dicts("scene3d")), "scene3d", sd_seg, "sceneseg")
    seg, s3d, dump = tmp_path / "seg.vpw", tmp_path / "s3d.vpw", tmp_path / "dump.bin"
    seg.write_bytes(vw.pack_state_dict(sd_seg))
    s3d.write_bytes(vw.pack_state_dict(sd_3d))
    r = subprocess.run([os.path.join(ROOT, "adapters", "test", "multicam_host"), str(seg), str(s3d), "8", "6", str(dump)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] is True and line["rccl_world"] == min(8, line["gpus_visible"]) and line["frames_per_s"] > 0
    raw = np.fromfile(dump, dtype=np.uint8)
    world, h, w = (int(v) for v in raw[:12].view(np.uint32))
    assert world == line["rccl_world"]
    eng = lib.Engine("sceneseg", seg.read_bytes(), precision="fp16x3")
    eng.set_decode_mode(lib.VP_DECODE_CLASS_INDEX)
    try:
        off = 12
        for cam in range(world):
            frame = raw[off:off + h * w * 3].reshape(h, w, 3)
            off += h * w * 3
            got = raw[off:off + 320 * 640].reshape(320, 640)
            off += 320 * 640
            eng.infer(frame)
            assert np.array_equal(got, eng.mask()), cam
            assert len(np.unique(got)) > 1          # a real class map, not a constant
    finally:
        eng.close()


@pytest.mark.gpu
def test_autospeed_stages_adapter_matches_oracle(tmp_path):
    """adapters/autospeed_hip_stages.hpp driven as AutoSpeedOnnxEngine::inference drives its two CPU stages (preprocessAutoSpeed -> the
    detector -> postProcess, onnxruntime_engine.cpp:115-168): the letterbox tensor and the kept detections, bit for bit against the oracle."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _autospeed_cases as cases
    from oracle import autospeed

    _build()
    raw = cases.raw_tensor(8400, 4, 21)
    rawf, out = tmp_path / "raw.bin", tmp_path / "out.bin"
    raw.tofile(rawf)
    r = subprocess.run([BIN, "autospeed", str(rawf), str(out), "8", "8400"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    blob = np.fromfile(out, dtype=np.uint8)
    frame = blob[:720 * 1280 * 3].reshape(720, 1280, 3)
    rest = blob[720 * 1280 * 3:]
    want_in, (scale, px, py) = autospeed.preprocess(frame)
    n_in = 3 * 640 * 640 * 4
    assert np.array_equal(rest[:n_in].view(np.float32).reshape(3, 640, 640), want_in)
    n = int(rest[n_in:n_in + 4].view(np.int32)[0])
    want = autospeed.postprocess(raw, 0.25, 0.45, scale, px, py, 1280, 720)
    assert n == len(want) and n > 5
    got = rest[n_in + 4:n_in + 4 + 24 * n].view(np.float32).reshape(n, 6).copy()
    got[:, 5] = rest[n_in + 4:n_in + 4 + 24 * n].view(np.int32).reshape(n, 6)[:, 5].astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
