import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402,F401  -- before libvp_hip so one HIP runtime is shared (see autoware_vision_pilot_amd/lib.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a CPU-only box: skip them explicitly there.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def vp_opts():
    """Developer options of the library under test (vp_set_option; the library does not read the environment): `setenv` / `delenv` like
    monkeypatch, everything cleared afterwards.  Acts on whichever library autoware_vision_pilot_amd.lib is bound to (the emulated one
    inside an `emu_lib` module)."""
    from autoware_vision_pilot_amd import lib

    class _Opts:
        def setenv(self, key, value):
            lib.set_option(key, value)

        def delenv(self, key):
            lib.set_option(key, None)

    yield _Opts()
    lib.clear_options()


SEEDS = {"sceneseg": 0, "scene3d": 1, "egolanes": 2, "domainseg": 3}
FRAME_SEED = 1


@pytest.fixture(scope="session")
def state_dicts():
    """Seeded numpy state-dicts (same seeds as tests/golden), built lazily per kind."""
    from oracle import weights

    cache = {}

    def get(kind):
        if kind not in cache:
            cache[kind] = weights.make_state_dict(kind, SEEDS[kind])
        return cache[kind]

    return get


@pytest.fixture(scope="session")
def frame720():
    from oracle import pre_post

    return pre_post.synthetic_frame(720, 1280, FRAME_SEED)


@pytest.fixture(scope="session")
def oracle_runs(state_dicts, frame720):
    """Oracle forward (torch CPU fp32) with intermediates for `kind` on the 720p frame, cached."""
    from oracle import nets, pre_post

    cache = {}

    def get(kind):
        if kind not in cache:
            sd = nets.to_torch(state_dicts(kind))
            x = torch.from_numpy(pre_post.preprocess(frame720, input_is_bgr=True, planes_rgb=False))
            out, inter = nets.forward(kind, sd, x, return_intermediates=True)
            cache[kind] = (out[0].numpy(), inter, x.numpy())
        return cache[kind]

    return get


@pytest.fixture(scope="session")
def engines(state_dicts):
    """libvp_hip engines per (kind, precision), cached for the session (GPU only)."""
    from autoware_vision_pilot_amd import lib, weights as vw

    cache = {}

    def get(kind, precision):
        key = (kind, precision)
        if key not in cache:
            cache[key] = lib.Engine(kind, vw.pack_state_dict(state_dicts(kind)), precision=precision)
        return cache[key]

    yield get
    for e in cache.values():
        e.close()
