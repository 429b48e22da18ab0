"""Batched encoder (vp_create_batched + vp_create_shared_frame): `frames` cameras per encoder pass, one shared-prefix head
per camera.  Every camera's logits must meet the same parity bar against the oracle as the single-frame engine (fp16x3:
1e-3), with different frames in the slots, and stay correct through graph replay with the frames swapped."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["sceneseg", "egolanes"])
def test_batched_encoder_heads_parity(state_dicts, kind):
    from autoware_vision_pilot_amd import lib, synthetic, weights as vw
    from oracle import nets, pre_post

    sd = state_dicts(kind)
    blob = vw.pack_state_dict(sd)
    frames = [synthetic.synthetic_frame(720, 1280, s) for s in (41, 42, 43)]
    tsd = nets.to_torch(sd)
    refs = [nets.forward(kind, tsd, torch.from_numpy(pre_post.preprocess(f, input_is_bgr=True, planes_rgb=False)))[0].numpy() for f in frames]
    enc = lib.Engine(kind, blob, precision="fp16x3", frames=3)
    heads = [lib.Engine(kind, blob, precision="fp16x3", base=enc, frame_index=f) for f in range(3)]
    try:
        assert enc.frames() == 3 and all(h.shared_level() == 1 for h in heads)
        for order in ((0, 1, 2), (2, 0, 1), (1, 2, 0)):       # pass 1 eager, pass 2 captures the graphs, pass 3 replays them
            for slot, fi in enumerate(order):
                enc.upload_frame(frames[fi], index=slot)
            enc.enqueue()
            for slot, fi in enumerate(order):
                heads[slot].infer_shared()
                got = heads[slot].logits()
                err = float(np.abs(got - refs[fi]).max() / np.abs(refs[fi]).max())
                assert err <= 1e-3, f"{kind} slot {slot} frame {fi}: {err:.3e}"
    finally:
        for h in heads:
            h.close()
        enc.close()


def test_batched_encoder_errors():
    from autoware_vision_pilot_amd import lib, synthetic, weights as vw

    blob = vw.pack_state_dict(synthetic.make_state_dict("sceneseg", 0))
    with pytest.raises(ValueError):
        lib.Engine("sceneseg", blob, frames=0)
    enc = lib.Engine("sceneseg", blob, frames=2)
    try:
        with pytest.raises(ValueError):
            lib.Engine("sceneseg", blob, base=enc, frame_index=2)                # slot out of range
        with pytest.raises(ValueError):
            enc.upload_frame(synthetic.synthetic_frame(64, 64, 1), index=5)
        with pytest.raises(lib.VpError, match="no outputs"):
            enc._ck(enc._lib.vp_fetch_outputs(enc._h))
    finally:
        enc.close()
