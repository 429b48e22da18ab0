"""Parameter inventory + seeded state-dict generator (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The generator itself lives in autoware_vision_pilot_amd/synthetic.py (pure data generation, shared so that checker and
engine see identical tensors; bench.py and the tools import it from there, never from oracle/); this module re-exports it.

The key layout is the reference's ``state_dict`` key layout, so the same dict can be
``load_state_dict``-ed into the reference's own modules (oracle/pin_against_reference.py)
and exported to the engine's weight blob (autoware_vision_pilot_amd/weights.py).

Key prefixes:
  SceneSeg  : Backbone.encoder.* SceneContext.* SceneNeck.* SceneSegHead.*
              (Models/model_components/scene_seg_network.py:11-21)
  Scene3D   : PreTrainedBackbone.pretrainedBackBone.encoder.* DepthContext.* DepthNeck.* SuperDepthHead.*
              (scene_3d_network.py:13-22, pre_trained_backbone.py:10)
  DomainSeg : DomainSegUpstream.{pretrainedBackBone.encoder,pretrainedContext,pretrainedNeck}.* DomainSegHead.*
              (domain_seg_network.py:11-14, domain_seg_upstream.py:10-20)
  EgoLanes  : BEVBackbone.encoder.* AutoSteerContext.* EgopathNeck.* EgoLanesHead.*
              (ego_lanes_network.py:14-26)

Init is NOT PyTorch's default: with default init the decoder contracts the signal
~0.6x/layer and argmax collapses to one class (SURVEY.md 8(d) init note), which
would make every parity check vacuous.  We use a variance-preserving init.
"""
from autoware_vision_pilot_amd.synthetic import (  # noqa: F401
    B0_LAST_OUT, B0_STAGES, B0_STEM_OUT, BN_EPS, MODEL_KINDS, PREFIX, _init, backbone_spec, context_channels, context_spec,
    head_spec, make_state_dict, model_spec, neck_spec, param_count, share_backbone)
