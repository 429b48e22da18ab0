#!/usr/bin/env python3
"""Developer tool: (re)write tests/golden/kernel_plan.json -- the kernel tag the engine's dispatch rules give EVERY layer of every
network in both precisions (vp_layer_kernel), taken from the CPU-emulated engine (tests/emul; the rules are host code and see the
MI355X's 256 CUs there too).  tests/test_engine_emulated.py::test_kernel_plan_is_the_committed_one compares against it, so a predicate
regression that silently picks another kernel fails the CPU suite.  Re-run after a DELIBERATE change of a rule and review the diff."""
import ctypes as ct
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))


def kernel_plan(lib):
    from autoware_vision_pilot_amd import synthetic, weights as vw

    out = {}
    for kind, seed in (("sceneseg", 0), ("scene3d", 1), ("domainseg", 3), ("egolanes", 2)):
        blob = vw.pack_state_dict(synthetic.make_state_dict(kind, seed))
        for prec in ("fp16x3", "fp16"):
            eng = lib.Engine(kind, blob, precision=prec)
            out[f"{kind}/{prec}"] = [[n, k] for (n, _, _), k in zip(eng.layers(), eng.layer_kernels())]
            eng.close()
    return out


if __name__ == "__main__":
    import build as eb

    from autoware_vision_pilot_amd import lib

    so = ct.CDLL(eb.build(), mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name, (res, args) in lib._SIGS.items():
        fn = getattr(so, name)
        fn.restype, fn.argtypes = res, args
    lib._lib = so
    path = os.path.join(ROOT, "tests", "golden", "kernel_plan.json")
    with open(path, "w") as f:
        json.dump(kernel_plan(lib), f, indent=0)
    print("written", path)
