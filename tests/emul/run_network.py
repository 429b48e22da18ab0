#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: one whole network through the emulated engine (tests/emul) on the CPU, against the oracle.
    python tests/emul/run_network.py [sceneseg|scene3d|domainseg|egolanes] [fp16x3|fp16]
The 360-GFLOP scene networks take a few minutes on 8 cores; AutoDrive (~8 s) and EgoLanes (~1 min) are in
tests/test_engine_emulated.py."""
import sys, time, os, ctypes as ct, numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)
import build as eb
from autoware_vision_pilot_amd import lib, synthetic, weights as vw
from oracle import nets, pre_post
so = ct.CDLL(eb.build(), mode=os.RTLD_LOCAL | os.RTLD_NOW)
for name,(res,args) in lib._SIGS.items():
    fn=getattr(so,name); fn.restype, fn.argtypes = res, args
lib._lib = so
kind = sys.argv[1] if len(sys.argv)>1 else "sceneseg"
prec = sys.argv[2] if len(sys.argv)>2 else "fp16x3"
sd = synthetic.make_state_dict(kind, 0)
frame = synthetic.synthetic_frame(720,1280,9)
eng = lib.Engine(kind, vw.pack_state_dict(sd), precision=prec)
so.vp_use_graph(eng._h, 0)
t0=time.time(); eng.infer(frame); print("infer", time.time()-t0, flush=True)
got = eng.logits()
x = torch.from_numpy(pre_post.preprocess(frame, input_is_bgr=True, planes_rgb=False))
assert np.array_equal(eng.input_tensor(), x.numpy())
ref = nets.forward(kind, nets.to_torch(sd), x)[0].numpy()
print("rel err", float(np.abs(got-ref).max()/np.abs(ref).max()), "class flips", int((got.argmax(0)!=ref.argmax(0)).sum()) if ref.shape[0]>1 else -1)
