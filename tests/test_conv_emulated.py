"""The MFMA convolution kernels (csrc/kernels_conv.hip, kernels_conv3x3.hip, kernels_conv3x3_x3.hip,
kernels_convt_rs.hip) and the engine code that packs weights and plans them, executed on the CPU: every csrc/ source
is compiled for the host on the HIP-on-CPU shim of tests/emul, whose `v_mfma_f32_32x32x16_f16` is a wave-level rendezvous
with the ISA's register layout.  Same entry point (vp_op_conv2d), reference (PyTorch) and tolerances as the GPU op tests
(tests/test_gpu_conv_op.py), on maps small enough for thread-per-work-item emulation: tile shapes, K blocks, split-K,
ragged channels / odd images, residual modes, activations, ConvTranspose pixel shuffle, both precisions."""
import ctypes as ct
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))

ACT = {0: lambda t: t, 1: F.gelu, 2: F.silu}


@pytest.fixture(scope="module")
def emu_lib():
    """autoware_vision_pilot_amd.lib bound to the emulated library for the duration of this module."""
    import build as emul_build

    from autoware_vision_pilot_amd import lib

    if not os.path.exists(emul_build.CLANG):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    so = ct.CDLL(emul_build.build(), mode=os.RTLD_LOCAL | os.RTLD_NOW)     # never RTLD_GLOBAL: same symbol names as libvp_hip.so
    for name, (res, args) in lib._SIGS.items():
        fn = getattr(so, name)
        fn.restype, fn.argtypes = res, args
    saved = lib._lib
    lib._lib = so
    yield lib
    lib._lib = saved


def _h(a):
    return a.astype(np.float16).astype(np.float32)


def _case(lib, cin, cout, h, w, ks, mode, act, res_mode, precision, cfgs, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((cin, h, w), dtype=np.float32)
    k = 2 if mode == 1 else ks
    wshape = (cin, cout, 2, 2) if mode == 1 else (cout, cin, k, k)
    wt = rng.standard_normal(wshape, dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * (1 if mode == 1 else k * k))))
    b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
    oh, ow = (2 * h, 2 * w) if mode == 1 else (h, w)
    res = rng.standard_normal((cout, oh, ow), dtype=np.float32) if res_mode else None
    xr, wr, rr = (x, wt, res) if precision == 1 else (_h(x), _h(wt), _h(res) if res is not None else None)
    xt, wtt, bt = (torch.from_numpy(v).double() for v in (xr, wr, b))
    y = F.conv_transpose2d(xt[None], wtt, bt, stride=2) if mode == 1 else F.conv2d(xt[None], wtt, bt, padding=ks // 2)
    y = ACT[act](y)
    if res_mode == 1:
        y = y + torch.from_numpy(rr).double()[None]
    elif res_mode == 2:
        r = torch.from_numpy(rr).double()[None]
        y = y * r + r
    ref = y[0].float().numpy()
    tol = 1.5e-3 if precision == 0 else 2e-5
    for tile, bk, nsplit in cfgs:
        got = lib.op_conv2d(x, wt, b, ks=ks, mode=mode, act=act, res=res, res_mode=res_mode, precision=precision, tile=tile, bk=bk, nsplit=nsplit)
        assert got.shape == ref.shape
        err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
        assert err <= tol, f"tile {tile} bk {bk} nsplit {nsplit}: err {err:.3e}"


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_halo_3x3_kernel_every_tile(emu_lib, precision):
    """kernels_conv3x3.hip: the five workgroup tiles (128/64/32 output channels x 16x16 / 8x16 pixels), split-K, ragged
    channel counts, an image that is not a multiple of the pixel tile."""
    cfgs = [(100, -1, 1), (101, -1, 2), (102, -1, 1), (103, -1, 3), (104, -1, 1)]
    _case(emu_lib, 32, 72, 11, 21, 3, 0, 1, 0, precision, cfgs, seed=1)
    _case(emu_lib, 40, 24, 16, 16, 3, 0, 0, 1, precision, [(103, -1, 1), (104, -1, 2)], seed=2)


def test_x3w8_kernel(emu_lib):
    """kernels_conv3x3_x3.hip (halo tiles 6 - 8): the pipelined fp16x3 kernels -- fragment prefetch across tap (and, tile 6,
    chunk) boundaries, three weight buffers, single-buffered halo with the two-barrier chunk hand-over (tile 7), register
    epilogue with both planes; one and several 32-channel chunks, GELU and no activation, an image that is not a multiple of
    the patch, two output-channel tiles; bit-identical to halo tile 1 (same K order)."""
    _case(emu_lib, 32, 128, 16, 32, 3, 0, 1, 0, 1, [(106, -1, 1), (107, -1, 1), (108, -1, 1)], seed=21)
    _case(emu_lib, 96, 256, 19, 21, 3, 0, 0, 0, 1, [(106, -1, 1), (107, -1, 1), (108, -1, 1)], seed=22)
    _case(emu_lib, 64, 64, 9, 17, 3, 0, 1, 0, 1, [(108, -1, 1), (108, -1, 2)], seed=26)         # 64-channel shape: one channel tile, ragged map; two K slices
    rng = np.random.default_rng(23)
    x = rng.standard_normal((64, 18, 33), dtype=np.float32)
    wt = rng.standard_normal((128, 64, 3, 3), dtype=np.float32) * np.float32(0.06)
    b = rng.standard_normal((128,), dtype=np.float32) * np.float32(0.1)
    a = emu_lib.op_conv2d(x, wt, b, ks=3, act=1, precision=1, tile=106, nsplit=1)
    c = emu_lib.op_conv2d(x, wt, b, ks=3, act=1, precision=1, tile=101, nsplit=1)
    assert np.array_equal(a, c)
    assert np.array_equal(emu_lib.op_conv2d(x, wt, b, ks=3, act=1, precision=1, tile=107, nsplit=1), c)
    assert np.array_equal(emu_lib.op_conv2d(x, wt, b, ks=3, act=1, precision=1, tile=108, nsplit=1), c)      # 64-channel shape: same K order too
    assert np.array_equal(emu_lib.op_conv2d(x, wt, b, ks=3, act=1, precision=1, tile=109, nsplit=1), c)      # round 5: 64 channels x 16x16 pixels, waves side by side
    # halo tile 9 (round 5): four waves side by side along the pixels (WCO = 1), each 64 channels x 64 pixels; six halo pieces per thread wait in registers
    # across the chunk hand-over; one and several chunks, ragged maps, GELU / none, split-K slices, and the fp16 engines' 64-channel chunks
    _case(emu_lib, 32, 64, 16, 32, 3, 0, 1, 0, 1, [(109, -1, 1)], seed=91)
    _case(emu_lib, 96, 128, 19, 21, 3, 0, 0, 0, 1, [(109, -1, 1), (109, -1, 2)], seed=92)
    _case(emu_lib, 128, 64, 35, 17, 3, 0, 1, 0, 0, [(109, -1, 1)], seed=93)
    # split-K slices of the 4-wave shape (fp32 partials + finish kernel), incl. a mul-add residual (context_layer_6's epilogue)
    _case(emu_lib, 160, 128, 10, 20, 3, 0, 1, 0, 1, [(107, -1, 2), (107, -1, 5)], seed=24)
    _case(emu_lib, 96, 256, 12, 18, 3, 0, 1, 2, 1, [(107, -1, 3), (108, -1, 3)], seed=25)


def test_x1_k64_kernel(emu_lib):
    """The VP_FP16 engines' form of the pipelined kernels (kernels_conv3x3_x3.hip, template parameter X1; halo tiles 6 - 8 with one fp16 plane
    per tensor): a step covers a 64-channel chunk, its two 32-channel halves travel as the two LDS planes (weights packed likewise by the
    engine), two MFMAs per fragment pair.  One and several chunks, GELU and none, ragged maps, two output-channel tiles, split-K slices with
    a mul-add residual behind the finish kernel; input channels that are not a multiple of 64 are refused."""
    _case(emu_lib, 64, 128, 16, 32, 3, 0, 1, 0, 0, [(106, -1, 1), (107, -1, 1), (108, -1, 1)], seed=27)     # one chunk
    _case(emu_lib, 192, 256, 19, 21, 3, 0, 0, 0, 0, [(106, -1, 1), (107, -1, 1), (108, -1, 1)], seed=28)    # three chunks, ragged map, no activation
    _case(emu_lib, 100, 64, 9, 17, 3, 0, 1, 0, 0, [(108, -1, 1), (108, -1, 2)], seed=29)                    # 100 -> 128 padded channels: two chunks; two K slices
    _case(emu_lib, 256, 128, 10, 20, 3, 0, 1, 2, 0, [(107, -1, 2), (107, -1, 4)], seed=30)                  # split-K + mul-add residual (finish kernel)
    rng = np.random.default_rng(31)
    x = rng.standard_normal((96, 16, 16), dtype=np.float32)
    wt = rng.standard_normal((128, 96, 3, 3), dtype=np.float32) * np.float32(0.05)
    with pytest.raises(emu_lib.VpError):
        emu_lib.op_conv2d(x, wt, np.zeros(128, np.float32), ks=3, act=1, precision=0, tile=107, nsplit=1)     # 96 input channels: not a multiple of 64


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_generic_gemm_kernel(emu_lib, precision):
    """kernels_conv.hip: implicit GEMM, all four tiles, both K blocks, split-K, 1x1 (K1 fast path incl. the register
    epilogues) and 3x3, residual add / mul-add, fewer input channels than one K block."""
    _case(emu_lib, 96, 24, 7, 13, 1, 0, 2, 0, precision, [(-1, -1, -1), (0, 32, 1), (1, 32, 2), (2, 64, 1), (3, 32, 1)], seed=3)
    _case(emu_lib, 48, 80, 6, 10, 1, 0, 0, 1, precision, [(-1, -1, -1), (2, 32, 2)], seed=4)
    # one, two and five 32-channel K blocks on ragged pixel counts / channel tiles (the encoder's small 1x1 GEMMs)
    _case(emu_lib, 16, 96, 9, 15, 1, 0, 2, 0, precision, [(-1, -1, -1)], seed=31)
    _case(emu_lib, 160, 40, 10, 20, 1, 0, 0, 1, precision, [(-1, -1, -1), (2, 32, 3)], seed=32)
    _case(emu_lib, 64, 200, 3, 43, 1, 0, 2, 0, precision, [(-1, -1, -1)], seed=33)
    _case(emu_lib, 32, 40, 5, 8, 3, 0, 1, 2, precision, [(1, 32, 1), (2, 64, 2)], seed=5)
    _case(emu_lib, 3, 32, 8, 8, 3, 0, 0, 0, precision, [(-1, -1, -1)], seed=6)


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_conv_transpose_pixel_shuffle(emu_lib, precision):
    """ConvTranspose2d(k2, s2) as a GEMM with the pixel-shuffle store; K = 128 on >= 2048 pixels takes the register-stationary kernel."""
    _case(emu_lib, 64, 48, 5, 6, 2, 1, 0, 0, precision, [(-1, -1, -1), (1, 32, 1), (2, 64, 1)], seed=7)
    _case(emu_lib, 128, 64, 4, 8, 2, 1, 1, 0, precision, [(-1, -1, -1)], seed=8)
    # >= 2048 pixels, K = 128, map width a multiple of 32: both engines take kernels_convt_rs.hip on their own
    _case(emu_lib, 128, 128, 40, 64, 2, 1, 0, 0, precision, [(-1, -1, -1)], seed=11)


def _skip_case(lib, cin, cs, cout, h, w, precision, seed):
    """ConvTranspose2d(k2, s2)(x) + Conv2d 1x1(skip) through vp_op_conv2d mode 2 against torch (fp64)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((cin, h, w), dtype=np.float32)
    sk = rng.standard_normal((cs, 2 * h, 2 * w), dtype=np.float32)
    wt = rng.standard_normal((cin, cout, 2, 2), dtype=np.float32) * np.float32(np.sqrt(2.0 / cin))
    ws = rng.standard_normal((cout, cs), dtype=np.float32) * np.float32(np.sqrt(2.0 / cs))
    bt, bs = (rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1) for _ in range(2))
    q = (lambda a: a) if precision == 1 else _h
    y = F.conv_transpose2d(torch.from_numpy(q(x)).double()[None], torch.from_numpy(q(wt)).double(), torch.from_numpy(bt).double(), stride=2)
    y = y + F.conv2d(torch.from_numpy(q(sk)).double()[None], torch.from_numpy(q(ws)).double()[:, :, None, None], torch.from_numpy(bs).double())
    ref = y[0].float().numpy()
    got = lib.op_conv2d(x, np.concatenate([wt.ravel(), ws.ravel()]), np.concatenate([bt, bs]), mode=2, res=sk, res_mode=cs, precision=precision)
    err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
    assert got.shape == ref.shape and err <= (1.5e-3 if precision == 0 else 2e-5), err
    return got


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_convt_register_stationary_kernel(emu_lib, precision, vp_opts):
    """kernels_convt_rs.hip (tile 5): weights stationary in registers, pixel tiles by LDS-DMA three deep, permuted weight rows +
    wave-private patch epilogue.  K = 128 (every workgroup covers the four quadrants) with one and with many tiles per workgroup
    (the prologue / steady-state / tail wait counts), K = 256 + 32 with the fused skip link (one quadrant per workgroup; ragged
    real channel counts below the padded ones), and the refusal of shapes it does not cover."""
    _case(emu_lib, 128, 128, 32, 64, 2, 1, 0, 0, precision, [(5, -1, 1)], seed=41)                  # 64 tiles, one per workgroup
    vp_opts.setenv("VP_CONVT_RS_GROUPS", "5")
    _case(emu_lib, 128, 128, 32, 64, 2, 1, 0, 0, precision, [(5, -1, 1)], seed=42)                  # 12-13 tiles per workgroup
    _case(emu_lib, 100, 120, 34, 64, 2, 1, 0, 0, precision, [(5, -1, 1)], seed=43)                  # channels padded to 128 / 128
    vp_opts.setenv("VP_CONVT_RS_GROUPS_K288", "3")        # (round 5: one key per shape case)
    a = _skip_case(emu_lib, 256, 24, 256, 16, 128, precision, seed=44)                              # 64 tiles x 4 quadrant slices, 21-22 per workgroup
    vp_opts.setenv("VP_CONVT_RS", "0")
    b = _skip_case(emu_lib, 256, 24, 256, 16, 128, precision, seed=44)                              # same layer through the GEMM kernel
    assert float(np.abs(a - b).max()) <= (4e-3 if precision == 0 else 2e-5)
    vp_opts.delenv("VP_CONVT_RS")
    with pytest.raises(emu_lib.VpError):
        _case(emu_lib, 64, 128, 32, 64, 2, 1, 0, 0, precision, [(5, -1, 1)], seed=45)               # K = 64: not covered


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_head_logits_conv_kernel(emu_lib, precision, vp_opts):
    """kernels_head.hip through vp_op_conv2d mode 3 (fp32 NCHW logits written by the kernel): 16x16x32 MFMA with the weights
    stationary in registers, LDS-DMA halo with the zero page for the border and the pad slots, persistent workgroups; 64 channels
    (one slab, 8x16 tiles) and 128 channels (two slabs meeting in LDS, 4x16 tiles), 1 and 3 logit channels, maps that are not a
    multiple of the tile; the same layers through the halo kernel's 32-channel tile."""
    tol = 1.5e-3 if precision == 0 else 2e-5
    for seed, (cin, cout, h, w) in enumerate([(64, 3, 19, 37), (128, 1, 10, 33), (128, 3, 16, 48), (64, 1, 8, 16)]):
        rng = np.random.default_rng(60 + seed)
        x = rng.standard_normal((cin, h, w), dtype=np.float32)
        wt = rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * 9)))
        b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
        q = (lambda a: a) if precision == 1 else _h
        ref = F.conv2d(torch.from_numpy(q(x)).double()[None], torch.from_numpy(q(wt)).double(), torch.from_numpy(b).double(), padding=1)[0].float().numpy()
        got = emu_lib.op_conv2d(x, wt, b, ks=3, mode=3, precision=precision)
        err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
        assert got.shape == ref.shape and err <= tol, (cin, cout, h, w, err)
        vp_opts.setenv("VP_HEAD_CONV", "0")                          # same layer through the halo kernel's 32-channel tile
        halo = emu_lib.op_conv2d(x, wt, b, ks=3, mode=3, precision=precision)
        vp_opts.delenv("VP_HEAD_CONV")
        assert float((np.abs(halo - ref) / np.maximum(1.0, np.abs(ref))).max()) <= tol


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_gemm_dma_kernel(emu_lib, precision, vp_opts):
    """kernels_gemm_dma.hip (tile 6): both operands by LDS-DMA three K steps deep, slot swizzle and weight-row permutation applied
    on the DMA's global side, wave-private patch epilogue; ragged pixel count (last tile re-reads the last pixel), K of 8 and of
    more steps, the fused skip link (K extension from the 2H x 2W tensor, quadrant of the workgroup), two weight-row tiles per
    quadrant; against torch and against the implicit-GEMM kernel on the same layer."""
    vp_opts.setenv("VP_GEMM_DMA", "1")          # fp16 engines take it only on request
    _case(emu_lib, 256, 256, 9, 15, 2, 1, 0, 0, precision, [(6, -1, 1)], seed=51)                   # 135 pixels: one full tile + 7 pixels
    _case(emu_lib, 320, 512, 8, 16, 2, 1, 0, 0, precision, [(6, -1, 1), (6, -1, 3)], seed=52)       # 10 K steps, 8 weight tiles; 3 K slices (3 + 3 + 4 steps)
    a = _skip_case(emu_lib, 256, 24, 256, 10, 16, precision, seed=53)                               # 8 + 1 K steps, 160 pixels
    b = _skip_case(emu_lib, 512, 40, 512, 12, 20, precision, seed=54)                               # 16 + 2 K steps, two tiles per quadrant
    vp_opts.setenv("VP_GEMM_DMA", "0")
    assert float(np.abs(a - _skip_case(emu_lib, 256, 24, 256, 10, 16, precision, seed=53)).max()) <= (4e-3 if precision == 0 else 2e-5)
    assert float(np.abs(b - _skip_case(emu_lib, 512, 40, 512, 12, 20, precision, seed=54)).max()) <= (4e-3 if precision == 0 else 2e-5)
    vp_opts.setenv("VP_GEMM_DMA", "1")
    with pytest.raises(emu_lib.VpError):
        _case(emu_lib, 256, 96, 9, 15, 2, 1, 0, 0, precision, [(6, -1, 1)], seed=55)                # 4 * 96 rows: not a multiple of 256


def test_map_kernel(emu_lib):
    """kernels_conv3x3_map.hip (halo tile 11): a workgroup holds a 32-channel weight slab x a K slice against ALL 800 pixels of a 20x40
    region; 16-channel K steps double-buffered by LDS-DMA with the slot swizzle on the global side, zero page at the map border, host
    packing in LDS image order, fp32 slabs -> finish kernel.  One region and four, one K slice and several (incl. slices of unequal
    length), ragged channel counts, GELU / none / the context's mul-add residual."""
    _case(emu_lib, 48, 40, 20, 40, 3, 0, 1, 0, 1, [(111, -1, 1), (111, -1, 2)], seed=71)        # 64 padded input channels = 4 steps; 2 channel tiles
    _case(emu_lib, 96, 72, 40, 80, 3, 0, 0, 0, 1, [(111, -1, 3)], seed=72)                      # four regions, 6 steps in 3 slices, ragged channel tile
    _case(emu_lib, 80, 32, 20, 40, 3, 0, 1, 2, 1, [(111, -1, 5)], seed=73)                      # 96 padded channels = 6 steps in 5 slices (1, 1, 1, 1, 2)
    # round 4, the context block's geometry: 10x20 regions, four waves carrying 2 / 2 / 2 / 1 pixel tiles, the seventh tile ragged (8 pixels)
    _case(emu_lib, 48, 72, 10, 20, 3, 0, 1, 0, 1, [(111, -1, 1), (111, -1, 2)], seed=74)        # one region; ragged channel tile
    _case(emu_lib, 96, 40, 10, 20, 3, 0, 1, 2, 1, [(111, -1, 3)], seed=75)                      # the context's mul-add residual behind 3 K slices
    _case(emu_lib, 32, 32, 30, 20, 3, 0, 0, 0, 1, [(111, -1, 1)], seed=76)                      # three 10x20 regions stacked (30 rows: not a 20x40 multiple)
    with pytest.raises(emu_lib.VpError):
        x = np.zeros((32, 16, 40), np.float32)
        emu_lib.op_conv2d(x, np.zeros((32, 32, 3, 3), np.float32), np.zeros(32, np.float32), ks=3, precision=1, tile=111, nsplit=1)   # 16 rows: not a region multiple
    # round 4, the VP_FP16 engines' form (X1): steps of 32 input channels, the step's two 16-channel halves in the two LDS planes, two MFMAs per fragment pair
    _case(emu_lib, 48, 40, 20, 40, 3, 0, 1, 0, 0, [(111, -1, 1), (111, -1, 2)], seed=77)        # 64 padded channels = 2 steps; one and two K slices
    _case(emu_lib, 160, 72, 40, 80, 3, 0, 0, 0, 0, [(111, -1, 3)], seed=78)                     # four regions, 5 steps in 3 slices (1, 2, 2), ragged channel tile
    _case(emu_lib, 96, 40, 10, 20, 3, 0, 1, 2, 0, [(111, -1, 3)], seed=79)                      # the context geometry, mul-add residual behind 3 slices


def test_map2_kernel(emu_lib):
    """kernels_conv3x3_map.hip, halo tile 12 (round 5): the map kernel on 64-channel weight slabs -- two M tiles per wave, rolling fragment
    prefetch, single-buffered weights in two tap groups behind two barriers per step, pixel tile 24 split by M tile between waves 0 and 1.  One
    region and four, one K slice and several (unequal lengths, a one-step slice: no A refill / no halo refill paths), one and two channel slabs,
    ragged channel counts (padded rows), GELU / none / mul-add residual; bit-identical to tile 11 for equal K slices (same summation order)."""
    _case(emu_lib, 48, 40, 20, 40, 3, 0, 1, 0, 1, [(112, -1, 1), (112, -1, 2)], seed=71)        # 64 padded input channels = 4 steps; one 64-channel slab (40 real rows)
    _case(emu_lib, 96, 72, 40, 80, 3, 0, 0, 0, 1, [(112, -1, 3)], seed=72)                      # four regions, 6 steps in 3 slices, two slabs (72 -> 128 rows)
    _case(emu_lib, 80, 32, 20, 40, 3, 0, 1, 2, 1, [(112, -1, 5)], seed=73)                      # 96 padded channels = 6 steps in 5 slices (1, 1, 1, 1, 2): one-step slices
    rng = np.random.default_rng(81)
    x = rng.standard_normal((64, 20, 40), dtype=np.float32)
    wt = rng.standard_normal((128, 64, 3, 3), dtype=np.float32) * np.float32(0.06)
    b = rng.standard_normal((128,), dtype=np.float32) * np.float32(0.1)
    for ns in (1, 2):
        a = emu_lib.op_conv2d(x, wt, b, ks=3, act=1, precision=1, tile=111, nsplit=ns)
        c = emu_lib.op_conv2d(x, wt, b, ks=3, act=1, precision=1, tile=112, nsplit=ns)
        assert np.array_equal(a, c), f"tile 12 differs from tile 11 at nsplit {ns}"
    with pytest.raises(emu_lib.VpError):
        emu_lib.op_conv2d(np.zeros((32, 10, 20), np.float32), np.zeros((64, 32, 3, 3), np.float32), np.zeros(64, np.float32), ks=3, precision=1, tile=112, nsplit=1)   # 10x20: tile 11's geometry only
    with pytest.raises(emu_lib.VpError):
        emu_lib.op_conv2d(np.zeros((32, 20, 40), np.float32), np.zeros((64, 32, 3, 3), np.float32), np.zeros(64, np.float32), ks=3, precision=0, tile=112, nsplit=1)   # parity mode only


def _q_e4m3(w):
    """oracle/autodrive.py quantize_fp8_e4m3 on one weight tensor: per-output-row symmetric OCP e4m3, returned de-quantised (fp32)."""
    from oracle import autodrive

    return autodrive.quantize_fp8_e4m3({"x.weight": w})["x.weight"]


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_fp8_weight_storage(emu_lib, precision):
    """VP_WEIGHTS_FP8 as STORAGE (round 4): on the kernels that stage weights through registers (generic GEMM kernel, halo kernel) the engine keeps
    one e4m3 BYTE per weight + a row scale in HBM and converts on the way to LDS.  The codes and scales are recovered from the de-quantised blob
    values, so the result must equal a convolution with exactly those values: against torch on the quantised weights, at the tolerance of
    un-quantised weights (fp16x3: 2e-5 -- a single wrong code would show at 6e-2 of a weight) -- 3x3 (halo tiles, split-K), 1x1, strided 3x3,
    ragged channels, rows scaled after quantisation as a BatchNorm fold does, and a row of tiny values (subnormal codes)."""
    rng = np.random.default_rng(700 + precision)
    tol = 2e-5 if precision == 1 else 3e-3
    q = (lambda a: a) if precision == 1 else _h
    for ks, cin, cout, h, w, tile, nsplit in ((3, 48, 72, 11, 21, -1, -1), (3, 96, 40, 16, 16, 103, 2), (1, 80, 200, 9, 15, -1, -1), (3, 32, 32, 8, 16, 104, 1)):
        wt = (rng.standard_normal((cout, cin, ks, ks)) * np.sqrt(2.0 / (cin * ks * ks))).astype(np.float32)
        wt[1] *= np.float32(1e-4)                                                 # a row whose codes are mostly e4m3 subnormals after scaling? no: scaled per row -- kept as a plain small row
        wq = _q_e4m3(wt)
        fold = (0.5 + rng.random(cout)).astype(np.float32) * np.where(rng.random(cout) < 0.3, -1.0, 1.0).astype(np.float32)
        wq = (wq * fold[:, None, None, None]).astype(np.float32)                   # what BatchNorm folding does to a quantised row (incl. a sign flip)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        x = rng.standard_normal((cin, h, w)).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(q(x)).double()[None], torch.from_numpy(wq).double(), torch.from_numpy(b).double(), padding=ks // 2)[0].float().numpy()
        got = emu_lib.op_conv2d(x, wq, b, ks=ks, precision=precision | 16, tile=tile, nsplit=nsplit)
        err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
        assert got.shape == ref.shape and err <= tol, (ks, cin, cout, err)
        plain = emu_lib.op_conv2d(x, wq, b, ks=ks, precision=precision, tile=tile, nsplit=nsplit)   # the same values as (hi, lo) fp16 planes
        assert float((np.abs(got - plain) / np.maximum(1.0, np.abs(plain))).max()) <= (2e-5 if precision == 1 else 2e-3)
