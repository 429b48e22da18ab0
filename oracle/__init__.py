"""CPU oracle for the VisionPilot per-frame hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``autoware_vision_pilot_amd`` (the product)
may import, call, link or execute anything in this package.  The only legal
callers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` -- and there only as the checker, never as the thing measured.
(The seeded weight / frame GENERATORS are pure data and live in the product package,
``autoware_vision_pilot_amd/synthetic.py``; this package re-exports them so that checker and
engine see identical tensors.  The dependency points from the oracle to the package, never back.)

What it is: a torch-CPU fp32 *functional* restatement (``torch.nn.functional``
calls over a flat ``state_dict``) of the reference networks

    preprocess -> EfficientNet-B0 features -> context -> neck -> head -> decode

each function citing the reference file:line it follows
(paths relative to /root/reference).

Parity pinning status
---------------------
* context / neck / head / feature-fusion (reference ``Models/model_components``):
  **pinned** -- ``oracle/pin_against_reference.py`` imports the reference's own
  ``nn.Module`` classes in the build container, loads the same seeded
  state-dict, and checks this restatement against them; the resulting vectors
  are committed under ``tests/golden/``.
* EfficientNet-B0 ``.features`` (torchvision, third-party, absent from
  /root/reference and from this image; requirement ``torchvision>=0.22.0``
  un-pinned, Models/requirements.txt:16; call sites Models/model_components/
  backbone.py:9,13-21): restated from the published architecture and **pinned
  against an independent implementation** -- ``oracle/pin_backbone_hf.py`` loads
  the same seeded weights into HuggingFace ``transformers``' EfficientNet (B0
  scale, BatchNorm eps 1e-5; its TensorFlow-style one-sided stride-2 pads replaced
  by torchvision's symmetric (k-1)//2) and compares the stem, all 16 MBConv blocks
  and the top conv: max difference 0.0.  Fixture ``tests/golden/backbone_hf_pin.npz``;
  the live comparison also runs in the CPU suite.  What stays unpinned is only
  "torchvision itself was never executed here" (its published definition is what
  both implementations follow).
* resize (OpenCV ``cv::resize`` INTER_LINEAR u8 / Pillow): **parity unpinned** --
  third-party arithmetic, no reference test fixes it.  We define our own
  integer bilinear (modelled on OpenCV's 11-bit fixed-point scheme) and pin the
  engine to *that* definition bit-exactly.  OpenCV's silent INTER_LINEAR ->
  INTER_AREA switch at an exact 2x downscale (e.g. 1280x640 frames) needs no
  branch: with half-pixel centres the exact-2x bilinear IS the 2x2 box average
  with the same rounding (tests/test_oracle_golden.py).
* decode (argmax / threshold / lane priority mask): pinned by construction --
  restated line by line from the reference's C++ loops and checked against
  ``torch.max`` here.
"""
