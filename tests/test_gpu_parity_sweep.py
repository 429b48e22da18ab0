"""Parity SWEEP of the parity mode (fp16x3): not one frame and one weight seed but 8 frames (random noise + smooth scenes, three
frame sizes) x 3 weight seeds x the four networks = 96 passes against the CPU oracle.

BASELINE.json's bar: float tensors within 1e-3, class-index maps bit-exact.  A class decision of the oracle itself is only defined
down to the oracle's own fp32 rounding, so the bar on the maps is stated in its strict form:
    * the decode is integer work -- the engine's map is ALWAYS bit-identical to the oracle decode of the engine's own logits;
    * against the oracle's map the flip count must be ZERO, except for pixels whose oracle decision margin (top-1 minus top-2 logit;
      |logit| for the threshold decodes) is at most twice the MEASURED maximum logit error of that pass -- a flip there is a tie
      inside the float tolerance, and each one is reported.
    * Passes that miss the plain bar are judged against an fp64 EVALUATION of the same network, every one of them (round 4; no
      builder-defined scale rule): the bar |error| <= 1e-3 x max(1, |ref|) is stated against the fp32 CPU reference, and on the ill-scaled
      weight seeds of this sweep (the same generator scales the encoder output up by 15-170x: |activations| to 1800, |logits| 59 ... 773)
      that reference is itself 0.5e-3 ... 3.8e-2 away from fp64 -- a distance from it says as much about the reference as about the
      engine.  Such a pass must satisfy   engine_vs_fp64 <= K64 x fp32ref_vs_fp64   with K64 = 3 -- below the ratio of the unit round-offs
      (fp16x3 carries 22 significand bits, two fp16 planes with the lo x lo product dropped, against fp32's 24: 4x).  Measured with the
      weight prescale of round 4: 9 of the 96 passes take this path (19 missed the plain bar in round 3), the engine is CLOSER to fp64
      than the fp32 reference in 6 of them, worst ratio 1.86.  Both distances, their ratio and the error as a fraction of the logits'
      range go into the table; a pass that meets the plain bar never takes this path.
The table goes to gpurun_out/ (copied to profiles/r04_parity_sweep.tsv): per pass max abs / rel logit error, pixels under 1e-3
margin, flips, largest flipped margin, and for re-judged passes the reference's and the engine's distance from fp64."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["sceneseg", "scene3d", "domainseg", "egolanes"]
BASE_SEED = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
FRAMES = [((720, 1280), 101, True), ((720, 1280), 102, False), ((360, 640), 103, True), ((360, 640), 104, False),
          ((1080, 1920), 105, True), ((1080, 1920), 106, False), ((720, 1280), 107, True), ((487, 651), 108, False)]
ROWS = []
JUDGED = []
# Round 6: 3.0 -> 2.5.  Measured worst 1.87 (DomainSeg weight seed 23), most passes < 1.  Where those rows lose their bits is known by layer
# (profiles/r06_where_the_bits_go.tsv: engine / reference stays 0.7-1.2 through encoder stage 5 and grows to ~2 across the MBConv blocks of stages
# 6-7 -- K = 672 / 1152 projections on 10x20 maps -- and does not grow further in the decoder) and by cause (profiles/r06_pair_storage_study.tsv: the
# (hi, lo) storage format ALONE, in exact arithmetic, is 3-4x an fp32 storage's distance; r06_mfma_accum.txt: the matrix pipe's accumulation is as
# good as a scalar fp32 loop's; r06_deep_encoder_bits_ab.txt: the fast SiLU and the depthwise tensor's subnormal low plane are 10-25 % each).
# The ratio of the unit round-offs (2^-23 : 2^-25) is 4.
K64 = 2.5


def _decode(kind, logits):
    from oracle import pre_post

    if kind == "sceneseg":
        srt = np.sort(logits, axis=0)
        return pre_post.argmax_classes(logits), srt[-1] - srt[-2]
    if kind == "egolanes":
        return (logits > 0).astype(np.int64), np.abs(logits)
    return (logits[0] > 0).astype(np.int64), np.abs(logits[0])       # domainseg threshold; scene3d has no class map (sign kept as a probe)


@pytest.mark.parametrize("wseed", [0, 10, 20])
@pytest.mark.parametrize("kind", KINDS)
def test_parity_sweep_fp16x3(kind, wseed):
    from autoware_vision_pilot_amd import lib, weights as vw
    from oracle import nets, pre_post, weights

    sd = weights.make_state_dict(kind, BASE_SEED[kind] + wseed)
    sdt = nets.to_torch(sd)
    eng = lib.Engine(kind, vw.pack_state_dict(sd), precision="fp16x3")
    mode = {"sceneseg": lib.VP_DECODE_CLASS_INDEX, "egolanes": None}.get(kind, lib.VP_DECODE_SEG_MASK)
    if mode is not None:
        eng.set_decode_mode(mode)
    rgb = kind == "egolanes"
    eng.set_input_format(lib.VP_BGR8, lib.VP_PLANES_RGB if rgb else lib.VP_PLANES_BGR)
    fails = []
    try:
        for (h, w), fseed, smooth in FRAMES:
            frame = pre_post.synthetic_frame(h, w, fseed, smooth=smooth)
            x = pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=rgb)
            ref = nets.forward(kind, sdt, torch.from_numpy(x))[0].numpy()
            eng.infer(frame)
            got = eng.logits()
            assert np.array_equal(eng.input_tensor(), x)
            assert np.isfinite(got).all()
            err_abs = float(np.abs(got - ref).max())
            err_rel = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
            ref_cls, margin = _decode(kind, ref)
            got_cls, _ = _decode(kind, got)
            if kind == "sceneseg":       # the engine's own decode: integer work on its own logits, bit-exact
                assert np.array_equal(eng.mask().astype(np.int64), got_cls)
            elif kind == "domainseg":
                assert np.array_equal(eng.mask() > 0, got_cls > 0)
            flips = got_cls != ref_cls
            nflip = int(flips.sum())
            worst = float(margin[flips].max()) if nflip else 0.0
            ref64_note = ""
            if err_rel > 1e-3:           # misses the plain bar: judged against fp64, always (see the module docstring)
                scale = float(np.abs(ref).max())
                sd64 = {k: v.double() for k, v in sdt.items()}
                r64 = nets.forward(kind, sd64, torch.from_numpy(x).double())[0].numpy()
                rel64 = lambda a: float((np.abs(a - r64) / np.maximum(1.0, np.abs(r64))).max())
                e_ref, e_got = rel64(ref.astype(np.float64)), rel64(got.astype(np.float64))
                ref64_note = (f"fp64-judged: fp32 reference {e_ref:.3e} / engine {e_got:.3e} from fp64 = {e_got / max(e_ref, 1e-30):.2f}x "
                              f"(|logits| up to {scale:.0f}: error = {err_abs / scale:.1e} of the range)")
                JUDGED.append((kind, BASE_SEED[kind] + wseed, fseed, e_ref, e_got))
                if not e_got <= K64 * e_ref:   # collected: the remaining frames still go into the table
                    fails.append(f"{kind} seed {wseed} frame {fseed}: logits rel err {err_rel:.3e}; vs fp64: engine {e_got:.3e}, reference {e_ref:.3e}")
            ROWS.append((kind, BASE_SEED[kind] + wseed, f"{h}x{w}", fseed, int(smooth), err_abs, err_rel, int((margin < 1e-3).sum()), nflip, worst, ref64_note))
            if kind != "scene3d":        # Scene3D's output is a depth map: no class decision to flip
                if not (nflip == 0 or worst <= 2.0 * err_abs):
                    fails.append(f"{kind} seed {wseed} frame {fseed}: {nflip} flips, largest oracle margin {worst:.3e} vs max logit error {err_abs:.3e}")
    finally:
        eng.close()
    assert not fails, fails


def test_parity_sweep_report():
    """Runs last in this file: writes the table, and asserts the claim every other test rests on, swept over the frames: with the BASE weight
    seeds (|logits| <= 16) every pass meets the plain bar |error| <= 1e-3 x max(1, |ref|)."""
    if not ROWS:
        pytest.skip("sweep did not run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_sweep_fp16x3.tsv"), "w") as f:
        f.write("# fp16x3 parity sweep against the CPU oracle (tests/test_gpu_parity_sweep.py)\n")
        f.write("# network\tweight_seed\tframe\tframe_seed\tsmooth\tmax_abs_err\tmax_rel_err\tpixels_margin_lt_1e-3\tclass_flips\tlargest_flipped_margin\tnote\n")
        for r in ROWS:
            f.write("\t".join(str(v) if not isinstance(v, float) else f"{v:.3e}" for v in r) + "\n")
        tot = sum(r[8] for r in ROWS)
        base = [r for r in ROWS if r[1] == BASE_SEED[r[0]]]
        assert base and all(r[6] <= 1e-3 and not r[10] for r in base), [r for r in base if r[6] > 1e-3]
        f.write(f"# base weight seeds: {len(base)} passes, all inside the plain 1e-3 bar (worst {max(r[6] for r in base):.3e}), {sum(r[8] for r in base)} ties\n")
        f.write(f"# {len(ROWS)} passes, {tot} class flips in total, worst rel err vs the fp32 reference {max(r[6] for r in ROWS):.3e}, {len(JUDGED)} passes miss the plain bar "
                f"and are judged against fp64 (engine_vs_fp64 <= {K64:g} x fp32ref_vs_fp64): worst ratio {max([j[4] / max(j[3], 1e-30) for j in JUDGED] or [0.0]):.2f}\n")
    print(f"parity sweep: {len(ROWS)} passes, {sum(r[8] for r in ROWS)} flips, worst rel err {max(r[6] for r in ROWS):.3e}")
