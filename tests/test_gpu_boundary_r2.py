"""Round-2 boundary behaviour on the GPU: the synchronous host-to-host path (strided views, pinned staging, output
selection, one-sync multi-head inference), the stateless createMaskFromTensorHIP twin, per-device kernel attributes from two
threads, and the RCCL all-gather behind the C ABI."""
import threading
import time

import numpy as np
import pytest
import torch

from oracle import pre_post, weights

pytestmark = pytest.mark.gpu


def test_strided_view_at_end_of_allocation(engines):
    """a view with an x offset that reaches the parent's last row guarantees only (h-1)*stride + 3*w readable
    bytes.  The parent ends exactly at the end of an mmap'ed region followed by a PROT_NONE page: an over-read faults."""
    import ctypes
    import mmap

    h, w_parent, x0, w = 360, 700, 60, 640
    nbytes = h * w_parent * 3
    page = mmap.PAGESIZE
    total = (nbytes + page - 1) // page * page + page
    mm = mmap.mmap(-1, total)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    libc = ctypes.CDLL(None)
    assert libc.mprotect(ctypes.c_void_p(base + total - page), ctypes.c_size_t(page), 0) == 0   # guard page
    start = total - page - nbytes                                                               # parent ends at the guard
    big = np.frombuffer(mm, dtype=np.uint8, count=nbytes, offset=start).reshape(h, w_parent, 3)
    big[:] = pre_post.synthetic_frame(h, w_parent, 5)
    view = big[:, x0:x0 + w]
    assert not view.flags["C_CONTIGUOUS"]
    eng = engines("sceneseg", "fp16")
    from autoware_vision_pilot_amd import lib

    for staging in (True, False):
        eng.set_pinned_staging(staging)
        rc = eng._lib.vp_infer(eng._h, view.ctypes.data_as(ctypes.c_void_p), h, w, view.strides[0])
        assert rc == 0
        assert np.array_equal(eng.input_tensor(), pre_post.preprocess(np.ascontiguousarray(view)))
    eng.set_pinned_staging(True)
    del view, big
    libc.mprotect(ctypes.c_void_p(base + total - page), ctypes.c_size_t(page), 3)
    assert lib.VP_OUT_MASK == 2


def test_output_selection_is_lazy_and_identical(engines, frame720):
    eng = engines("sceneseg", "fp16x3")
    eng.infer(frame720)
    want_l, want_m = eng.logits(), eng.mask()
    eng.set_outputs(logits=False, mask=True)
    eng.infer(frame720)
    assert np.array_equal(eng.mask(), want_m)
    assert np.array_equal(eng.logits(), want_l)          # fetched on first use
    eng.set_outputs(logits=False, mask=False)
    eng.infer(frame720)
    assert np.array_equal(eng.logits(), want_l) and np.array_equal(eng.mask(), want_m)
    eng.set_outputs(True, True)
    with pytest.raises(ValueError):
        eng._ck(eng._lib.vp_set_outputs(eng._h, 7))


def test_infer_multi_equals_separate_calls(state_dicts, frame720):
    from autoware_vision_pilot_amd import lib, weights as vw

    sd_seg = state_dicts("sceneseg")
    sd_3d = weights.share_backbone(dict(state_dicts("scene3d")), "scene3d", sd_seg, "sceneseg")
    base = lib.Engine("sceneseg", vw.pack_state_dict(sd_seg), precision="fp16x3")
    head = lib.Engine("scene3d", vw.pack_state_dict(sd_3d), precision="fp16x3", base=base)
    other = lib.Engine("sceneseg", vw.pack_state_dict(sd_seg), precision="fp16x3")
    try:
        base.infer(frame720)
        head.infer_shared()
        a, b, m = base.logits(), head.logits(), base.mask()
        for _ in range(2):
            base.infer_multi([head], frame720)
            assert np.array_equal(base.logits(), a) and np.array_equal(head.logits(), b) and np.array_equal(base.mask(), m)
        with pytest.raises(ValueError):
            base.infer_multi([other], frame720)          # not a shared-prefix engine of this base
    finally:
        head.close()
        base.close()
        other.close()


def test_visualize_rejects_foreign_geometry(engines, frame720):
    eng = engines("sceneseg", "fp16")
    eng.infer(frame720)
    assert eng.frame_hw() == (720, 1280)
    with pytest.raises(ValueError):
        eng.visualize_mask(0, (360, 640))                # smaller buffer than the frame: refused, not overflowed
    assert np.array_equal(eng.visualize_mask(0, (720, 1280)), pre_post.visualize_mask(eng.mask(), frame720, 0))


@pytest.mark.parametrize("c,mode", [(3, 0), (1, 0), (3, 1), (3, 2)])
def test_decode_logits_host_bit_exact(c, mode):
    from autoware_vision_pilot_amd import lib

    rng = np.random.default_rng(7)
    x = rng.standard_normal((c, 96, 200)).astype(np.float32)
    x[:, 3, 5] = 0.25                                   # exact ties: first maximum wins
    got = lib.decode_logits_host(x, mode)
    want = {0: pre_post.seg_mask_u8, 1: pre_post.egolanes_priority_mask,
            2: lambda l: pre_post.argmax_classes(l).astype(np.uint8)}[mode](x)
    assert np.array_equal(got, want)


def test_two_threads_two_devices(state_dicts, frame720):
    """Engines created and first-launched concurrently from two host threads, on two GPUs when the box has them (function
    attributes are per device), else both on gpu 0 (racing first launches must still be clean)."""
    from autoware_vision_pilot_amd import lib, weights as vw

    ndev = torch.cuda.device_count()
    blob = vw.pack_state_dict(state_dicts("egolanes"))
    ref = lib.Engine("egolanes", blob, precision="fp16x3")
    ref.infer(frame720)
    want = ref.logits()
    ref.close()
    got, errs = {}, []

    def work(i):
        # three rounds of (create, first frame = eager warm-up, second frame = graph capture, read back, destroy), the second thread half a
        # round behind: one thread's vp_create / uploads / frees overlap the other's stream capture.  (Round 4: with hipMemcpy / hipMemset on
        # the legacy stream the runtime invalidated the capture -- "would make the legacy stream depend on a capturing ... stream".)
        try:
            if i:
                time.sleep(0.12)
            for _ in range(3):
                e = lib.Engine("egolanes", blob, precision="fp16x3", gpu_id=i % max(1, ndev))
                for _ in range(2):
                    e.infer(frame720)
                got[i] = e.logits()
                e.tensor_read(0)                     # a tensor read-back (device-to-host copy) in the mix
                e.close()
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex)[:600])

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert np.array_equal(got[i], want)


def test_rccl_gather_world_1(engines, frame720):
    """The C-ABI all-gather on a single GPU (world = 1): binds librccl, creates the communicator, runs ncclAllGather on the
    engine's stream behind the graph with no host sync in between."""
    from autoware_vision_pilot_amd import lib

    eng = engines("sceneseg", "fp16")
    eng.upload_frame(frame720)
    eng.enqueue()
    eng.sync()
    comm = lib.Comm(lib.Comm.unique_id(), 0, 1, 0, 3 * 320 * 640 * 4)
    try:
        eng.enqueue()
        comm.gather(eng, lib.VP_GATHER_MASK)             # async on the engine stream
        got = comm.fetch(eng)
        eng.fetch_outputs()
        assert got.shape == (1, 320 * 640) and np.array_equal(got[0].reshape(320, 640), eng.mask())
        eng.enqueue()
        comm.gather(eng, lib.VP_GATHER_LOGITS)
        lg = comm.fetch(eng, np.float32)
        eng.fetch_outputs()
        assert np.array_equal(lg[0].reshape(3, 320, 640), eng.logits())
        with pytest.raises(ValueError):
            lib.Comm(lib.Comm.unique_id(), 2, 1, 0, 16)  # rank >= world
        # one communicator per engine IN FLIGHT, enforced: a second engine (its own stream) may not use the communicator while the first
        # engine's all-gather has not completed (VP_ERR_STATE); once it has, sequential use is legal
        other = engines("scene3d", "fp16")
        other.upload_frame(frame720)
        other.enqueue()
        other.sync()
        for _ in range(20):                               # ~20 frames of work queued ahead of the gather: it cannot have completed
            eng.enqueue()
        comm.gather(eng, lib.VP_GATHER_MASK)
        with pytest.raises(lib.VpError, match="one communicator per"):
            comm.gather(other, lib.VP_GATHER_LOGITS)
        eng.sync()
        comm.gather(other, lib.VP_GATHER_LOGITS)          # the first stream has drained: accepted
        got = comm.fetch(other, np.float32)
        other.fetch_outputs()
        assert np.array_equal(got[0][: 320 * 640].reshape(320, 640), other.logits()[0])
    finally:
        comm.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_rccl_gather_two_ranks(tmp_path):
    """Two processes, camera r on GPU r, the unique id handed over through a file (no torch.distributed involved)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "gather_rank.py")
    idf = tmp_path / "id.bin"
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(idf), str(tmp_path / f"out{r}.npy")]) for r in range(2)]
    assert all(p.wait(timeout=600) == 0 for p in procs)
    a, b = np.load(tmp_path / "out0.npy"), np.load(tmp_path / "out1.npy")
    assert a.shape == (2, 80 * 160) and np.array_equal(a, b) and not np.array_equal(a[0], a[1])
