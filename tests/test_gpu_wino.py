"""Parity of the 1-D Winograd form of the wide split-fp16 layers (bsvd_amd/csrc/conv3x3_wino.hip, BsvdConvArgs.w_wino_packed)
against the CPU oracle (double-accumulating conv) -- F(2,3) and F(4,3), every operand form the direct kernel's tests cover:
zero / compact / full-frame halos (bsvd_arch.py:94,104,112-113), PixelShuffle + skip add (:263-267, :402), ragged sizes,
sub-tiles below / right of the image, an odd number of sub-tiles per frame, the zero-chunk skip of a clip's end frames."""
import numpy as np
import pytest
import torch

from helpers import maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state
from test_gpu_f16x3 import _Net, from_split, to_split

pytestmark = pytest.mark.gpu
TIGHT = 2e-4


def _dev():
    return torch.device("cuda", 0)


def _exec(net, st, wide_conv):
    from bsvd_amd.engine import HipExecutor, PackedNet
    return HipExecutor(PackedNet(net, {k: torch.as_tensor(v) for k, v in st.items()}, _dev(), "f16x3", wide_conv))


CASES = [
    # cin, cout, tsm, act, epi, T, H, W
    (128, 128, True, "relu6", 0, 3, 10, 19),       # ragged in x and y, three temporal sources
    (128, 128, True, "relu6", 0, 1, 16, 16),       # exactly one sub-tile (second one of the pair dead); single frame: both sources halos
    (256, 256, True, "relu", 0, 2, 9, 17),
    (128, 128, False, "none", 0, 2, 35, 48),       # three sub-tile rows of F(4,3), five of F(2,3); odd sub-tile count per frame
    (256, 512, False, "none", 1, 2, 9, 13),        # UpBlock: PixelShuffle + skip add, four channel tiles
    (128, 256, False, "none", 1, 1, 12, 20),
    (128, 128, True, "relu6", 0, 4, 40, 64),       # several workgroups per frame
]


# the forms of the product library; the measurement variants (F(4,3), forced tiles, 4-wave workgroups, the persistent form, the
# all-positions-per-wave kernel) run the same cases from tests/measure_driver.py on a -DBSVD_MEASURE build (test_gpu_measure.py)
PRODUCT_FORMS = ["wino2", "wino6"]


@pytest.mark.parametrize("wide_conv", PRODUCT_FORMS)
@pytest.mark.parametrize("cin,cout,tsm,act,epi,T,H,W", CASES)
def test_wino_layer_vs_oracle(wide_conv, cin, cout, tsm, act, epi, T, H, W):
    from bsvd_amd.netspec import ConvSpec
    from bsvd_amd.schedule import Halo
    rs = np.random.RandomState(cin + cout + H)
    sp = ConvSpec("l", "l", cin, cout, 1, tsm, act, epi)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                       ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    gex, oex = _exec(_Net(sp), st, wide_conv), OracleExecutor(st, double=True)
    assert "l" in gex.packed.wino
    x = torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32))
    extra = extra_dev = None
    eps = 0
    if epi == 1:
        extra = torch.from_numpy(rs.standard_normal((T, 2 * H, 2 * W, cout // 4)).astype(np.float32))
        extra = from_split(to_split(extra))
        extra_dev, eps = to_split(extra).to(_dev()), cout // 4
    xq = from_split(to_split(x))
    halos = [(None, None)]
    if tsm:
        fold = sp.fold
        hp = from_split(to_split(torch.from_numpy(rs.standard_normal((H, W, fold)).astype(np.float32))))
        hn = from_split(to_split(torch.from_numpy(rs.standard_normal((H, W, fold)).astype(np.float32))))
        halos.append((Halo(hp, fold, 0), Halo(hn, fold, 0)))
        full = from_split(to_split(torch.from_numpy(rs.standard_normal((1, H, W, cin)).astype(np.float32))))
        halos.append((Halo(full, cin, fold), Halo(full, cin, 0)))
        halos.append((None, Halo(hn, fold, 0)))
    gex.record_variants = True
    for hp, hn in halos:
        want = oex.conv(sp, xq, hp, hn, extra, eps, 1)
        d = lambda h: None if h is None else Halo(to_split(h.t).to(_dev()), h.pstride, h.coff)
        got = from_split(gex.conv(sp, to_split(x).to(_dev()), d(hp), d(hn), extra_dev, eps, 1).cpu())
        assert "_kernel<F(%s,3)" % wide_conv[4] in gex.last_variant and gex.last_variant.startswith("wino_" if wide_conv.endswith("b") else "winox_"), gex.last_variant
        err = maxabs(got.numpy(), want.numpy())
        print("%s layer %s max-abs %.3e (|y| max %.1f)" % (wide_conv, (cin, cout, tsm, epi, T, H, W), err, float(want.abs().max())))
        assert err < TIGHT


def _conv_with_code(gex, sp, x, code, hp=None, hn=None):
    """one launch with an explicit BsvdConvArgs.wino_m (2 = the kernel picks its tile, 42 = never the half-height tile)"""
    import ctypes
    from bsvd_amd import _lib
    a, y = gex.build_args(sp, x, hp, hn)
    a.wino_m = code
    buf = ctypes.create_string_buffer(96)
    _lib.check(gex.lib.bsvd_conv3x3_variant(ctypes.byref(a), buf, 96), "variant")
    _lib.check(gex.lib.bsvd_conv3x3(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv")
    return y, buf.value.decode()


def test_half_height_tile_is_bit_identical_and_taken_by_grids_that_do_not_fill_the_chip():
    """wino_m = 2 launches the 8-row tile when the 16-row grid would leave CUs idle (single-frame launches of the stream schedules:
    256 -> 256 at 135 x 240 is 270 workgroups on 256 CUs); both tiles run the same instruction sequence per output, so the choice --
    which depends on the launch's size -- cannot break stream == clip (bsvd_arch.py:485-552 vs :555-569).  wino_m = 42 (launches that
    share the chip with another graph branch) never takes it: same bits."""
    from bsvd_amd.netspec import ConvSpec
    from bsvd_amd.schedule import Halo
    rs = np.random.RandomState(5)
    # (256 -> 256 at 135 x 240 was the case this tile was built for -- 270 workgroups; since the short last row band folds, that launch is
    #  240 + 16 = 256 workgroups of the full tile, one round: see the folded-band test below.  128 -> 128 on 135 x 240: 128 workgroups, half the chip)
    for cin, H, W, want_half in ((128, 270, 480, False), (128, 135, 240, True)):
        sp = ConvSpec("l", "l", cin, cin, 1, True, "relu6", 0)
        st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cin, cin, 3, 3)),
                           ("l.bias", (cin,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
        x = to_split(torch.from_numpy(rs.standard_normal((1, H, W, cin)).astype(np.float32))).to(_dev())
        gex = _exec(_Net(sp), st, "wino2")
        y_auto, v_auto = _conv_with_code(gex, sp, x, 2)
        y_full, v_full = _conv_with_code(gex, sp, x, 42)
        assert v_auto.endswith("[8 rows]") == want_half and not v_full.endswith("[8 rows]"), (cin, v_auto, v_full)
        assert torch.equal(y_auto, y_full)
        if want_half:       # ... and a 10-frame clip of the same layer takes the full tile and produces the same bits per frame
            gex.record_variants = True
            x10 = x.expand(10, -1, -1, -1).contiguous()
            y10 = gex.conv(sp, x10)
            assert not gex.last_variant.endswith("[8 rows]"), gex.last_variant
            y1 = gex.conv(sp, x, Halo(x, cin, sp.fold), Halo(x, cin, 0))     # frame 5's neighbours = the same frame
            assert gex.last_variant.endswith("[8 rows]")
            assert torch.equal(y10[5:6], y1)


FOLD_CASES = [
    # cin, cout, tsm, epi, T, H, W: 16-row grids whose last row band has <= 8 live rows
    (128, 128, True, 0, 1, 24, 16),       # one tile column: the folded tile's second half lies outside the image
    (128, 128, True, 0, 3, 23, 40),       # three tile columns (odd: the last folded tile is half dead), 7 live rows, three temporal sources
    (256, 256, True, 0, 2, 17, 64),       # one live row in the band, two channel tiles
    (256, 512, False, 1, 2, 20, 33),      # PixelShuffle + skip through the folded epilogue, four channel tiles
    (128, 256, False, 1, 1, 40, 50),      # two full bands above the folded one
]


@pytest.mark.parametrize("xf32", [False, True])
@pytest.mark.parametrize("cin,cout,tsm,epi,T,H,W", FOLD_CASES)
def test_folded_last_row_band_is_bit_identical_to_the_half_height_tile(cin, cout, tsm, epi, T, H, W, xf32):
    """F(2,3) on a 16-row tile grid whose last row band has <= 8 live rows: that band's tiles are walked two per workgroup, as the two 8-row
    halves of one 16-row tile (conv3x3_winox.hip: XCfg::FOLD, wx_grid) -- 256 -> 256 on one 135 x 240 frame is then 256 workgroups, one round on
    256 CUs.  Same instruction sequence per output as every other tile: bit-identical to the 8-row tile (which these small grids take with
    wino_m = 2; 42 = never the 8-row tile = the folded grid), with fp16-pair and plain-fp32 tensors, and within the oracle tolerance."""
    import ctypes
    from bsvd_amd import _lib
    from bsvd_amd.netspec import ConvSpec
    from bsvd_amd.schedule import Halo
    rs = np.random.RandomState(cin + H + W)
    sp = ConvSpec("l", "l", cin, cout, 1, tsm, "relu6" if epi == 0 else "none", epi)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                       ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    gex, oex = _exec(_Net(sp), st, "wino2"), OracleExecutor(st, double=True)
    gex.force_x_f32 = gex.force_y_f32 = xf32
    x = torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32))
    if not xf32:
        x = from_split(to_split(x))
    enc = (lambda t: t.to(_dev())) if xf32 else (lambda t: to_split(t).to(_dev()))
    dec = (lambda t: t.cpu()) if xf32 else (lambda t: from_split(t.cpu()))
    extra = extra_dev = None
    eps = 0
    if epi == 1:
        extra = from_split(to_split(torch.from_numpy(rs.standard_normal((T, 2 * H, 2 * W, cout // 4)).astype(np.float32))))
        extra_dev, eps = to_split(extra).to(_dev()), cout // 4
    hp = hn = hpd = hnd = None
    if tsm:
        q = (lambda t: t) if xf32 else (lambda t: from_split(to_split(t)))
        hp = Halo(q(torch.from_numpy(rs.standard_normal((H, W, sp.fold)).astype(np.float32))), sp.fold, 0)
        hn = Halo(q(torch.from_numpy(rs.standard_normal((H, W, sp.fold)).astype(np.float32))), sp.fold, 0)
        hpd, hnd = Halo(enc(hp.t), sp.fold, 0), Halo(enc(hn.t), sp.fold, 0)
    outs = {}
    for code in (2, 42):
        a, y = gex.build_args(sp, enc(x), hpd, hnd, extra_dev, eps, 1)
        a.wino_m = code
        buf = ctypes.create_string_buffer(96)
        _lib.check(gex.lib.bsvd_conv3x3_variant(ctypes.byref(a), buf, 96), "variant")
        _lib.check(gex.lib.bsvd_conv3x3(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv")
        torch.cuda.synchronize()
        outs[code] = (y, buf.value.decode())
    assert outs[2][1].endswith("[8 rows]" + ("[f32 in]" if xf32 else "")) and "[8 rows]" not in outs[42][1], (outs[2][1], outs[42][1])
    assert torch.equal(outs[2][0], outs[42][0])
    err = maxabs(dec(outs[42][0]).numpy(), oex.conv(sp, x, hp, hn, extra, eps, 1).numpy())
    print("folded band, layer %s f32 %s: max-abs %.3e" % ((cin, cout, tsm, epi, T, H, W), xf32, err))
    assert err < TIGHT


@pytest.mark.parametrize("seed", range(8))
def test_folded_band_random_shapes_bitwise(seed):
    """seeded random layers on 16-row grids with a short last band (any column count, 1-3 frames, both tensor formats, with and without the
    reverse tile walk the engine alternates between layers): folded grid (wino_m 42) == 8-row tile (wino_m 2) bit for bit."""
    import ctypes
    from bsvd_amd import _lib
    from bsvd_amd.netspec import ConvSpec
    from bsvd_amd.schedule import Halo
    rs = np.random.RandomState(1000 + seed)
    cin = int(rs.choice([128, 256]))
    cout = int(rs.choice([128, 256]))
    H = 16 * int(rs.randint(1, 4)) + int(rs.randint(1, 9))
    W = int(rs.randint(8, 90))
    T = int(rs.randint(1, 4))
    xf32 = bool(rs.randint(0, 2))
    sp = ConvSpec("l", "l", cin, cout, 1, True, "relu6", 0)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                       ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    gex = _exec(_Net(sp), st, "wino2")
    gex.force_x_f32 = gex.force_y_f32 = xf32
    enc = (lambda t: t.to(_dev())) if xf32 else (lambda t: to_split(t).to(_dev()))
    x = enc(torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32)))
    hp = Halo(enc(torch.from_numpy(rs.standard_normal((H, W, sp.fold)).astype(np.float32))), sp.fold, 0)
    hn = Halo(enc(torch.from_numpy(rs.standard_normal((H, W, sp.fold)).astype(np.float32))), sp.fold, 0)
    outs = []
    for code in (2, 42):
        for flip in (0, 1):
            a, y = gex.build_args(sp, x, hp, hn)
            a.wino_m = code
            a.tile_order = flip
            _lib.check(gex.lib.bsvd_conv3x3(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv")
            torch.cuda.synchronize()
            outs.append(y)
    print("fold fuzz %d: %d->%d %dx%dx%d f32 %s" % (seed, cin, cout, T, H, W, xf32))
    for y in outs[1:]:
        assert torch.equal(outs[0], y)


def test_one_135_row_frame_of_the_256_channel_layers_is_one_round_of_full_tiles():
    """the launch the folded band was built for: 256 -> 256 on ONE 135 x 240 frame (the quarter-resolution temporal layers of a 540 x 960 stream
    step) -- 8 full bands x 15 x 2 + 8 x 2 folded = 256 workgroups, so wino_m = 2 keeps the full tile; same bits as that frame inside a clip."""
    from bsvd_amd.netspec import ConvSpec
    from bsvd_amd.schedule import Halo
    rs = np.random.RandomState(11)
    cin = 256
    sp = ConvSpec("l", "l", cin, cin, 1, True, "relu6", 0)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cin, cin, 3, 3)),
                       ("l.bias", (cin,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    x = to_split(torch.from_numpy(rs.standard_normal((1, 135, 240, cin)).astype(np.float32))).to(_dev())
    gex = _exec(_Net(sp), st, "wino2")
    gex.record_variants = True
    y1 = gex.conv(sp, x, Halo(x, cin, sp.fold), Halo(x, cin, 0))
    assert "[8 rows]" not in gex.last_variant, gex.last_variant
    y10 = gex.conv(sp, x.expand(10, -1, -1, -1).contiguous())
    assert torch.equal(y10[5:6], y1)


def test_product_library_has_no_measurement_variants():
    """the kernel variants DESIGN 4.1d records as slower live in measurement builds only.  The in-tree library
    reports a product build, the engine refuses their names, and the ABI answers their wino_m codes with -19."""
    import ctypes
    import os
    from bsvd_amd import _lib, engine
    from bsvd_amd.netspec import ConvSpec
    lib = _lib.load()
    if os.environ.get("BSVD_HIP_LIB"):
        pytest.skip("a library override is active")
    assert lib.bsvd_build_info() & _lib.BUILD_MEASURE == 0
    assert engine.WIDE_CONV == ("direct", "wino2", "wino6", "wino26")
    sp = ConvSpec("l", "l", 128, 128, 1, True, "relu6", 0)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (128, 128, 3, 3)),
                       ("l.bias", (128,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    for name in engine.MEASURE_WIDE_CONV:
        with pytest.raises(ValueError, match="measurement"):
            _exec(_Net(sp), st, name)
    gex = _exec(_Net(sp), st, "wino2")
    x = to_split(torch.zeros(1, 8, 16, 128)).to(_dev())
    a, _ = gex.build_args(sp, x)
    for code in (4, 12, 14, 22, 32, 36, 52, 62):
        a.wino_m = code
        assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -19 and b"measurement" in lib.bsvd_last_error(), code


@pytest.mark.parametrize("wide_conv", ["wino2", "wino6"])
def test_wino_refuses_what_it_cannot_run(wide_conv):
    """an explicit w_wino_packed never falls back silently: stride 2 / RESID / odd fold return -19 with the reason"""
    import ctypes
    from bsvd_amd import _lib
    from bsvd_amd.netspec import ConvSpec
    sp = ConvSpec("l", "l", 128, 128, 1, True, "relu6", 0)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (128, 128, 3, 3)),
                       ("l.bias", (128,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    gex = _exec(_Net(sp), st, wide_conv)
    x = to_split(torch.zeros(1, 8, 16, 128)).to(_dev())
    a, _ = gex.build_args(sp, x)
    lib = _lib.load()
    a.stride = 2
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -19 and b"stride" in lib.bsvd_last_error()
    a.stride, a.fold = 1, 8
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -19 and b"fold" in lib.bsvd_last_error()
    a.fold, a.wino_m = 16, 3
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -19
