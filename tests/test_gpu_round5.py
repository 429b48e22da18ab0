"""Round-5 boundary additions on the GPU: frame pools registered for DMA (vp_register_frames), the shape getter that does not fetch
(vp_output_shape), the adapters' lazy-logits default."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_registered_frame_pool_same_results(engines, frame720):
    """A frame inside a registered pool travels by one DMA from the caller's memory (no staging copy): same input tensor, same logits, for a
    packed frame and for a strided view; frames outside the pool keep the staged path; the registry refuses overlaps and unknown pointers."""
    from autoware_vision_pilot_amd import lib

    eng = engines("sceneseg", "fp16")
    eng.infer(frame720)
    ref_in, ref_out = eng.input_tensor().copy(), eng.logits().copy()
    pool = np.zeros((2,) + frame720.shape, dtype=np.uint8)
    pool[1] = frame720
    lib.register_frames(pool)
    try:
        with pytest.raises(lib.VpError):
            lib.register_frames(pool[1])                      # overlaps the registered range
        for _ in range(3):                                   # eager, capture, replay
            eng.infer(pool[1])
            assert np.array_equal(eng.input_tensor(), ref_in) and np.array_equal(eng.logits(), ref_out)
        view = pool[1][100:500, 200:1000]                     # a cv::Mat ROI: strided rows inside the pool
        eng.infer(view)
        a = eng.logits().copy()
        eng.infer(np.ascontiguousarray(view))                 # the same pixels from pageable memory (staged path)
        assert np.array_equal(eng.logits(), a)
        eng.upload_frame(pool[1])                             # the asynchronous pair on a registered frame
        eng.enqueue()
        eng.fetch_outputs()
        assert np.array_equal(eng.logits(), ref_out)
    finally:
        lib.unregister_frames(pool)
    with pytest.raises(lib.VpError):
        lib.unregister_frames(pool)                           # not registered any more
    eng.infer(pool[1])                                        # ... and the frame takes the staged path again
    assert np.array_equal(eng.logits(), ref_out)


def test_output_shape_does_not_fetch(engines, frame720):
    eng = engines("sceneseg", "fp16")
    eng.set_outputs(logits=False, mask=True)
    try:
        eng.infer(frame720)
        assert eng.output_shape() == (1, 3, 320, 640)
        assert not eng.host_logits_current()                  # the shape getter left the tensor in HBM
        lg = eng.logits()                                     # fetched on demand
        assert eng.host_logits_current() and lg.shape == (3, 320, 640)
    finally:
        eng.set_outputs(True, True)
