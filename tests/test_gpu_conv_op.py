"""Unit parity of the MFMA implicit-GEMM conv kernel (csrc/kernels_conv.hip) through the C ABI (vp_op_conv2d)
against plain PyTorch fp32 ops on the same seeded inputs: ragged sizes, every tile shape, both K-block sizes,
split-K, pixel-shuffle (ConvTranspose) store, fused bias/GELU/SiLU/residual epilogues, both precisions."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACT = {0: lambda t: t, 1: F.gelu, 2: F.silu}


def _h(a):
    """what an fp16 tensor in HBM holds"""
    return a.astype(np.float16).astype(np.float32)


def _reference(x, w, b, ks, mode, act, res, res_mode, fp16):
    if fp16:
        x, w = _h(x), _h(w)
        res = _h(res) if res is not None else None
    xt, wt, bt = (torch.from_numpy(v).double() for v in (x, w, b))
    y = F.conv_transpose2d(xt[None], wt, bt, stride=2) if mode == 1 else F.conv2d(xt[None], wt, bt, padding=ks // 2)
    y = ACT[act](y)
    if res_mode == 1:
        y = y + torch.from_numpy(res).double()[None]
    elif res_mode == 2:
        r = torch.from_numpy(res).double()[None]
        y = y * r + r
    return y[0].float().numpy()


CASES = [
    # cin, cout, h, w, ks, mode, act, res_mode
    (32, 64, 16, 24, 3, 0, 1, 0),
    (40, 24, 13, 9, 3, 0, 0, 0),       # ragged channels and odd image, narrower than any tile
    (128, 128, 32, 40, 3, 0, 1, 0),
    (256, 96, 10, 20, 3, 0, 1, 2),     # context-like: 200 pixels, mul-add residual (scene_context.py:56)
    (80, 160, 20, 40, 1, 0, 0, 1),     # skip-link like 1x1 with residual add
    (96, 24, 17, 33, 1, 0, 2, 0),      # MBConv-like 1x1 + SiLU
    (64, 48, 10, 12, 2, 1, 0, 0),      # ConvTranspose2d k2 s2 -> pixel shuffle
    (64, 3, 24, 40, 3, 0, 0, 0),       # final-layer-like, 3 output channels
    (3, 32, 8, 8, 3, 0, 0, 0),         # fewer input channels than one K block
    (96, 72, 20, 40, 3, 0, 1, 0),      # neck-like small maps
    (64, 40, 40, 80, 3, 0, 0, 0),      # ragged channel tile
    (160, 33, 16, 32, 3, 0, 1, 0),     # AutoDrive P5 maps
    (64, 32, 32, 64, 3, 0, 0, 0),
    (16, 96, 40, 64, 1, 0, 2, 0),      # MBConv expand, one 32-channel K block (pointwise kernel)
    (672, 112, 20, 40, 1, 0, 0, 1),    # MBConv project with residual, 21 K blocks (odd), ragged channel tile
    (320, 1280, 10, 20, 1, 0, 2, 0),   # features[8]
]


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
@pytest.mark.parametrize("case", CASES, ids=[f"c{c[0]}-{c[1]}_{c[2]}x{c[3]}_k{c[4]}m{c[5]}a{c[6]}r{c[7]}" for c in CASES])
def test_conv_op_matches_torch(case, precision):
    from autoware_vision_pilot_amd import lib

    cin, cout, h, w, ks, mode, act, res_mode = case
    rng = np.random.default_rng(hash(case) % (2**31))
    x = rng.standard_normal((cin, h, w), dtype=np.float32)
    k = 2 if mode == 1 else ks
    wshape = (cin, cout, 2, 2) if mode == 1 else (cout, cin, k, k)
    wt = (rng.standard_normal(wshape, dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * (1 if mode == 1 else k * k)))))
    b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
    oh, ow = (2 * h, 2 * w) if mode == 1 else (h, w)
    res = rng.standard_normal((cout, oh, ow), dtype=np.float32) if res_mode else None
    ref = _reference(x, wt, b, ks, mode, act, res, res_mode, fp16=(precision == 0))
    tol = 1.5e-3 if precision == 0 else 2e-5  # fp16: one output rounding (2^-11) + accumulation order
    cfgs = [(-1, -1, -1), (0, 32, 1), (1, 32, 2), (2, 32, 3), (3, 32, 1), (0, 64, 1), (2, 64, 2)]
    if ks == 3 and mode == 0 and h >= 8 and w >= 16:  # LDS-halo 3x3 kernel tiles (kernels_conv3x3.hip)
        cfgs += [(100, -1, 1), (101, -1, 2), (102, -1, 1), (103, -1, 3), (104, -1, 1)]
    for tile, bk, nsplit in cfgs:
        got = lib.op_conv2d(x, wt, b, ks=ks, mode=mode, act=act, res=res, res_mode=res_mode, precision=precision, tile=tile, bk=bk,
                            nsplit=nsplit)
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        assert got.shape == ref.shape
        assert err.max() <= tol, f"tile={tile} bk={bk} nsplit={nsplit}: max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("cin,cout,h,w,act", [(128, 128, 64, 96, 1), (96, 256, 35, 50, 0), (512, 128, 32, 32, 1), (32, 384, 16, 16, 1)])
def test_x3w8_kernel_matches_torch_and_halo_tile(cin, cout, h, w, act):
    """kernels_conv3x3_x3.hip (halo tiles 6 - 8; fp16x3, and since round 4 the fp16 engines on 64-channel chunks): fragments prefetched across tap / chunk boundaries, three
    weight buffers, register epilogue; 8-wave 16x16 shape and 4-wave 8x16 shape.  Same K order as halo tile 1 => bit-identical."""
    from autoware_vision_pilot_amd import lib

    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.standard_normal((cin, h, w), dtype=np.float32)
    wt = rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
    ref = _reference(x, wt, b, 3, 0, act, None, 0, fp16=False)
    got = lib.op_conv2d(x, wt, b, ks=3, act=act, precision=1, tile=106, nsplit=1)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= 2e-5, err.max()
    assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, act=act, precision=1, tile=101, nsplit=1))
    assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, act=act, precision=1, tile=107, nsplit=1))   # 4-wave shape, single halo buffer
    assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, act=act, precision=1, tile=108, nsplit=1))   # 64-channel shape, three workgroups per CU
    assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, act=act, precision=1, tile=109, nsplit=1))   # round 5: 64 channels x 16x16 pixels, waves side by side
    for ns in (2, 3):                                                                                      # split-K slices + finish kernel
        if cin >= 32 * ns:
            sk = lib.op_conv2d(x, wt, b, ks=3, act=act, precision=1, tile=107, nsplit=ns)
            assert (np.abs(sk - ref) / np.maximum(1.0, np.abs(ref))).max() <= 2e-5
            assert np.array_equal(sk, lib.op_conv2d(x, wt, b, ks=3, act=act, precision=1, tile=108, nsplit=ns))   # same slices, same finish kernel
    # VP_FP16 engines: the same shapes on 64-channel chunks (X1: the chunk's halves in the two LDS planes); other channel counts are refused
    ref16 = _reference(x, wt, b, 3, 0, act, None, 0, fp16=True)
    for tile in (106, 107, 108, 109):
        if ((cin + 31) // 32 * 32) % 64 == 0:
            g16 = lib.op_conv2d(x, wt, b, ks=3, act=act, precision=0, tile=tile, nsplit=1)
            assert (np.abs(g16 - ref16) / np.maximum(1.0, np.abs(ref16))).max() <= 1.5e-3, tile
        else:
            with pytest.raises(lib.VpError):
                lib.op_conv2d(x, wt, b, ks=3, act=act, precision=0, tile=tile, nsplit=1)


def test_conv_op_transpose_detecting():
    """Identity weights with an asymmetric input: catches a swapped output-channel/pixel mapping of the MFMA result."""
    from autoware_vision_pilot_amd import lib

    cin = cout = 64
    x = np.arange(cin * 8 * 40, dtype=np.float32).reshape(cin, 8, 40) % 97 - 48.0
    w = np.zeros((cout, cin, 1, 1), dtype=np.float32)
    for c in range(cout):
        w[c, (c * 7 + 3) % cin, 0, 0] = 1.0  # a permutation, not symmetric
    got = lib.op_conv2d(x, w, np.zeros(cout, np.float32), ks=1, precision=1)
    ref = x[[(c * 7 + 3) % cin for c in range(cout)]]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_convt_register_stationary_kernel(precision, vp_opts):
    """kernels_convt_rs.hip (tile 5; the engine's choice for the two large-map up-sampling stages of every head): weights
    stationary in registers, pixel tiles by LDS-DMA, LDS-only barriers with explicit vmcnt waits.  K = 128 at the head's real
    size (160x320 -> 320x640: 6-7 tiles per workgroup) and with one tile per workgroup; K = 256 + 32 with the fused skip link
    against torch and against the GEMM kernel on the same layer."""
    from autoware_vision_pilot_amd import lib

    tol = 1.5e-3 if precision == 0 else 2e-5
    rng = np.random.default_rng(77 + precision)
    for cin, cout, h, w in ((128, 128, 160, 320), (128, 128, 32, 64), (100, 120, 34, 64)):
        x = rng.standard_normal((cin, h, w), dtype=np.float32)
        wt = rng.standard_normal((cin, cout, 2, 2), dtype=np.float32) * np.float32(np.sqrt(2.0 / cin))
        b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
        ref = _reference(x, wt, b, 2, 1, 0, None, 0, fp16=(precision == 0))
        got = lib.op_conv2d(x, wt, b, ks=2, mode=1, precision=precision, tile=5)
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        assert err.max() <= tol, (cin, cout, h, w, err.max())
        for _ in range(3):  # the tile stream is asynchronous (DMA three deep, stores in flight): same bits every run
            assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=2, mode=1, precision=precision, tile=5))
    with pytest.raises(lib.VpError):
        lib.op_conv2d(x[:64], wt[:64], b, ks=2, mode=1, precision=precision, tile=5)   # K = 64: not covered
    # fused skip link, the head's real size (80x160 -> 160x320, 256 + 32 channels)
    cin, cs, cout, h, w = 256, 24, 256, 80, 160
    x = rng.standard_normal((cin, h, w), dtype=np.float32)
    sk = rng.standard_normal((cs, 2 * h, 2 * w), dtype=np.float32)
    wt = rng.standard_normal((cin, cout, 2, 2), dtype=np.float32) * np.float32(np.sqrt(2.0 / cin))
    ws = rng.standard_normal((cout, cs), dtype=np.float32) * np.float32(np.sqrt(2.0 / cs))
    bt, bs = (rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1) for _ in range(2))
    q = (lambda a: a) if precision == 1 else _h
    y = F.conv_transpose2d(torch.from_numpy(q(x)).double()[None], torch.from_numpy(q(wt)).double(), torch.from_numpy(bt).double(), stride=2)
    y = y + F.conv2d(torch.from_numpy(q(sk)).double()[None], torch.from_numpy(q(ws)).double()[:, :, None, None], torch.from_numpy(bs).double())
    ref = y[0].float().numpy()
    wcat, bcat = np.concatenate([wt.ravel(), ws.ravel()]), np.concatenate([bt, bs])
    got = lib.op_conv2d(x, wcat, bcat, mode=2, res=sk, res_mode=cs, precision=precision)
    assert (np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max() <= tol
    for _ in range(3):
        assert np.array_equal(got, lib.op_conv2d(x, wcat, bcat, mode=2, res=sk, res_mode=cs, precision=precision))
    vp_opts.setenv("VP_CONVT_RS", "0")
    gemm = lib.op_conv2d(x, wcat, bcat, mode=2, res=sk, res_mode=cs, precision=precision)
    assert (np.abs(gemm - ref) / np.maximum(1.0, np.abs(ref))).max() <= tol


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_head_logits_conv_kernel(precision, vp_opts):
    """kernels_head.hip (vp_op_conv2d mode 3: the kernel writes fp32 NCHW logits): 16x16x32 MFMA, weights stationary in registers,
    LDS-DMA halo with the zero page; the heads' real shapes (64 -> 3 and 128 -> 1 on 320x640: 12 / 25 tiles per persistent
    workgroup) and ragged maps, against torch and against the halo kernel's 32-channel tile; same bits run to run."""
    from autoware_vision_pilot_amd import lib

    tol = 1.5e-3 if precision == 0 else 2e-5
    for seed, (cin, cout, h, w) in enumerate([(64, 3, 320, 640), (128, 1, 320, 640), (128, 3, 80, 160), (64, 2, 19, 37), (128, 1, 10, 33)]):
        rng = np.random.default_rng(90 + seed)
        x = rng.standard_normal((cin, h, w), dtype=np.float32)
        wt = rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * 9)))
        b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
        ref = _reference(x, wt, b, 3, 0, 0, None, 0, fp16=(precision == 0))
        got = lib.op_conv2d(x, wt, b, ks=3, mode=3, precision=precision)
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        assert got.shape == ref.shape and err.max() <= tol, (cin, cout, h, w, err.max(), np.unravel_index(err.argmax(), err.shape))
        assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, mode=3, precision=precision))
        vp_opts.setenv("VP_HEAD_CONV", "0")
        halo = lib.op_conv2d(x, wt, b, ks=3, mode=3, precision=precision)
        vp_opts.delenv("VP_HEAD_CONV")
        assert (np.abs(halo - ref) / np.maximum(1.0, np.abs(ref))).max() <= tol


@pytest.mark.parametrize("precision", [0, 1], ids=["fp16", "fp16x3"])
def test_gemm_dma_kernel(precision, vp_opts):
    """kernels_gemm_dma.hip (tile 6; the engine's choice for the three small-map up-sampling stages in the parity mode): the neck's
    real shapes incl. the fused skip link, against torch and the implicit-GEMM kernel; same bits run to run (asynchronous DMA ring)."""
    from autoware_vision_pilot_amd import lib

    vp_opts.setenv("VP_GEMM_DMA", "1")
    tol = 1.5e-3 if precision == 0 else 2e-5
    rng = np.random.default_rng(123 + precision)
    for cin, cs, cout, h, w in ((1280, 80, 1280, 10, 20), (768, 40, 768, 20, 40), (512, 24, 512, 40, 80), (256, 0, 256, 9, 15)):
        x = rng.standard_normal((cin, h, w), dtype=np.float32)
        wt = rng.standard_normal((cin, cout, 2, 2), dtype=np.float32) * np.float32(np.sqrt(2.0 / cin))
        bt = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
        q = (lambda a: a) if precision == 1 else _h
        y = F.conv_transpose2d(torch.from_numpy(q(x)).double()[None], torch.from_numpy(q(wt)).double(), torch.from_numpy(bt).double(), stride=2)
        if cs:
            sk = rng.standard_normal((cs, 2 * h, 2 * w), dtype=np.float32)
            ws = rng.standard_normal((cout, cs), dtype=np.float32) * np.float32(np.sqrt(2.0 / cs))
            bs = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
            y = y + F.conv2d(torch.from_numpy(q(sk)).double()[None], torch.from_numpy(q(ws)).double()[:, :, None, None], torch.from_numpy(bs).double())
            run = lambda: lib.op_conv2d(x, np.concatenate([wt.ravel(), ws.ravel()]), np.concatenate([bt, bs]), mode=2, res=sk, res_mode=cs, precision=precision)
        else:
            run = lambda: lib.op_conv2d(x, wt, bt, ks=2, mode=1, precision=precision, tile=6)
        ref = y[0].float().numpy()
        got = run()
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        assert got.shape == ref.shape and err.max() <= tol, (cin, cs, cout, h, w, err.max())
        for _ in range(3):
            assert np.array_equal(got, run())
        if cs:
            vp_opts.setenv("VP_GEMM_DMA", "0")
            assert (np.abs(run() - ref) / np.maximum(1.0, np.abs(ref))).max() <= tol
            vp_opts.setenv("VP_GEMM_DMA", "1")


@pytest.mark.parametrize("cin,cout,h,w,act,res_mode,nsplit", [(1280, 768, 20, 40, 1, 0, -1), (512, 512, 40, 80, 1, 0, -1), (96, 40, 40, 80, 0, 0, 3), (256, 64, 20, 40, 1, 2, 5)])
def test_map_kernel_matches_torch(cin, cout, h, w, act, res_mode, nsplit):
    """kernels_conv3x3_map.hip (halo tile 11; the engine's choice for the neck's 20x40 / 40x80 layers in the parity mode): a workgroup holds
    a 32-channel weight slab x a K slice against all 800 pixels of a 20x40 region (LDS-DMA double buffering, swizzled 32-byte rows, zero
    page at the border).  decode_layer_0 / 3 at their real sizes with the engine's own split factor, a ragged channel tile on four
    regions, the context block's mul-add residual; against torch and run to run."""
    from autoware_vision_pilot_amd import lib

    rng = np.random.default_rng(cin + cout + h)
    x = rng.standard_normal((cin, h, w), dtype=np.float32)
    wt = rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
    res = rng.standard_normal((cout, h, w), dtype=np.float32) if res_mode else None
    ref = _reference(x, wt, b, 3, 0, act, res, res_mode, fp16=False)
    got = lib.op_conv2d(x, wt, b, ks=3, act=act, res=res, res_mode=res_mode, precision=1, tile=111, nsplit=nsplit)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= 2e-5, err.max()
    assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, act=act, res=res, res_mode=res_mode, precision=1, tile=111, nsplit=nsplit))
    auto = lib.op_conv2d(x, wt, b, ks=3, act=act, res=res, res_mode=res_mode, precision=1)     # the engine's own choice
    assert (np.abs(auto - ref) / np.maximum(1.0, np.abs(ref))).max() <= 2e-5


@pytest.mark.parametrize("cin,cout,h,w,act,res_mode,nsplit", [(1280, 768, 20, 40, 1, 0, -1), (768, 512, 40, 80, 1, 0, -1), (512, 512, 40, 80, 1, 0, 4), (96, 40, 40, 80, 0, 0, 3),
                                                             (256, 64, 20, 40, 1, 2, 5), (48, 128, 20, 40, 0, 0, 4)])
def test_map2_kernel_matches_torch_and_tile11(cin, cout, h, w, act, res_mode, nsplit):
    """kernels_conv3x3_map.hip, halo tile 12 (round 5; the engine's choice for the neck's 20x40 / 40x80 layers in the parity mode): 64-channel weight slabs,
    two M tiles per wave, single-buffered weights in two tap groups behind two barriers per step, pixel tile 24 split between waves 0 and 1.
    decode_layer_0 / 2 / 3 at their real sizes, ragged channel counts, one-step K slices, the mul-add residual; against torch, run to run, and bit
    for bit against tile 11 with the same K slices (same summation order)."""
    from autoware_vision_pilot_amd import lib

    rng = np.random.default_rng(cin + cout + h)
    x = rng.standard_normal((cin, h, w), dtype=np.float32)
    wt = rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(np.sqrt(2.0 / (cin * 9)))
    b = rng.standard_normal((cout,), dtype=np.float32) * np.float32(0.1)
    res = rng.standard_normal((cout, h, w), dtype=np.float32) if res_mode else None
    ref = _reference(x, wt, b, 3, 0, act, res, res_mode, fp16=False)
    got = lib.op_conv2d(x, wt, b, ks=3, act=act, res=res, res_mode=res_mode, precision=1, tile=112, nsplit=nsplit)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= 2e-5, err.max()
    for _ in range(3):
        assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, act=act, res=res, res_mode=res_mode, precision=1, tile=112, nsplit=nsplit))
    if nsplit > 0:
        assert np.array_equal(got, lib.op_conv2d(x, wt, b, ks=3, act=act, res=res, res_mode=res_mode, precision=1, tile=111, nsplit=nsplit))
