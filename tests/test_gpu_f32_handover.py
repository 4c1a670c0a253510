"""Plain-fp32 hand-over between split-mode layers (BsvdConvArgs.y_f32 / x_f32, engine.F32_HANDOVER_DEFAULT): a tensor whose only reader is a
Winograd-form layer is stored as fp32 channels instead of fp16 pairs, and the reader's input transform starts from the value.  Tested: the
Winograd kernel with fp32 input (and fp32 output) on every operand form of the temporal gather (bsvd_arch.py:21-50, 94, 104, 112-113) and the
PixelShuffle + skip epilogue (:263-267) against the double-accumulating oracle, the direct stride-2 kernel's fp32 store (:229-255), the ABI's
refusals, which tensors the engine hands over, and the whole network: on == off inside the error class, stream == clip bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import bsvd_keys, maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state
from test_gpu_f16x3 import _Net, from_split, to_split
from test_gpu_wino import CASES, PRODUCT_FORMS, _exec

pytestmark = pytest.mark.gpu
TIGHT = 2e-4


def _dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("wide_conv", PRODUCT_FORMS)
@pytest.mark.parametrize("cin,cout,tsm,act,epi,T,H,W", CASES + [(128, 128, True, "relu6", 0, 1, 135, 50), (256, 256, True, "relu", 0, 2, 20, 33)])
def test_wino_layer_with_fp32_input_and_output_vs_oracle(wide_conv, cin, cout, tsm, act, epi, T, H, W):
    from bsvd_amd.netspec import ConvSpec
    from bsvd_amd.schedule import Halo
    rs = np.random.RandomState(cin + cout + H + 1)
    sp = ConvSpec("l", "l", cin, cout, 1, tsm, act, epi)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                       ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    gex, oex = _exec(_Net(sp), st, wide_conv), OracleExecutor(st, double=True)
    gex.force_x_f32 = True
    x = torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32))          # exact fp32 values: nothing to re-encode
    extra = extra_dev = None
    eps = 0
    if epi == 1:
        extra = from_split(to_split(torch.from_numpy(rs.standard_normal((T, 2 * H, 2 * W, cout // 4)).astype(np.float32))))
        extra_dev, eps = to_split(extra).to(_dev()), cout // 4          # the skip tensor stays an fp16-pair tensor
    halos = [(None, None)]
    if tsm:
        fold = sp.fold
        hp = torch.from_numpy(rs.standard_normal((H, W, fold)).astype(np.float32))
        hn = torch.from_numpy(rs.standard_normal((H, W, fold)).astype(np.float32))
        full = torch.from_numpy(rs.standard_normal((1, H, W, cin)).astype(np.float32))
        halos += [(Halo(hp, fold, 0), Halo(hn, fold, 0)), (Halo(full, cin, fold), Halo(full, cin, 0)), (None, Halo(hn, fold, 0))]
    gex.record_variants = True
    d = lambda h: None if h is None else Halo(h.t.to(_dev()), h.pstride, h.coff)      # noqa: E731
    for hp, hn in halos:
        want = oex.conv(sp, x, hp, hn, extra, eps, 1)
        outs = {}
        for yf in (False, True):
            gex.force_y_f32 = yf
            y = gex.conv(sp, x.to(_dev()), d(hp), d(hn), extra_dev, eps, 1).cpu()
            assert "[f32 in]" in gex.last_variant, gex.last_variant
            outs[yf] = y if yf else from_split(y)
            err = maxabs(outs[yf].numpy(), want.numpy())
            print("%s f32 in, %s out, layer %s: max-abs %.3e (|y| max %.1f)" % (wide_conv, "f32" if yf else "pairs", (cin, cout, tsm, epi, T, H, W), err, float(want.abs().max())))
            assert err < TIGHT
        # the pair store only re-encodes the fp32 result
        assert torch.equal(from_split(to_split(outs[True])), outs[False])


@pytest.mark.parametrize("cin,cout,T,H,W", [(64, 128, 2, 21, 38), (128, 256, 1, 10, 19), (64, 128, 1, 1, 1)])
def test_stride2_direct_layer_stores_plain_fp32(cin, cout, T, H, W):
    """DownBlock's conv (bsvd_arch.py:229-255) feeds a temporal-fusion layer only: with y_f32 it stores the values the pair store encodes"""
    from bsvd_amd.engine import HipExecutor, PackedNet
    from bsvd_amd.netspec import ConvSpec
    rs = np.random.RandomState(H)
    sp = ConvSpec("l", "l", cin, cout, 2, False, "relu6", 0)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                       ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    gex = HipExecutor(PackedNet(_Net(sp), {k: torch.as_tensor(v) for k, v in st.items()}, _dev(), "f16x3"))
    x = to_split(torch.from_numpy((rs.rand(T, H, W, cin) * 3).astype(np.float32))).to(_dev())
    y_pairs = from_split(gex.conv(sp, x).cpu())
    gex.force_y_f32 = True
    y_f32 = gex.conv(sp, x).cpu()
    assert torch.equal(from_split(to_split(y_f32)), y_pairs)
    want = OracleExecutor(st, double=True).conv(sp, from_split(x.cpu()))
    assert maxabs(y_f32.numpy(), want.numpy()) < TIGHT


def test_which_tensors_are_handed_over_and_the_abi_refusals():
    import bsvd_amd
    from bsvd_amd import _lib
    kw = dict(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None, precision="f16x3")
    m = bsvd_amd.BSVD(**kw).to(_dev()).eval()
    pk = m._executor(_dev()).packed
    names = sorted(k.split(".", 1)[1] for k in pk.f32_out if k.startswith("temp1."))
    # the producers whose only reader is a Winograd-form layer: both stride-2 convs, 6 of the 8 temporal-fusion convs (not downc0.c2 = x1:
    # read by downc1's direct stride-2 conv and as a skip; not upc1.c2? it feeds upc1's conv: handed over) and upc2's PixelShuffle conv
    assert names == sorted(["downc0.convblock.0", "downc0.memconv.c1.op.conv", "downc1.convblock.0", "downc1.memconv.c1.op.conv",
                            "downc1.memconv.c2.op.conv", "upc2.memconv.c1.op.conv", "upc2.memconv.c2.op.conv", "upc2.convblock.0",
                            "upc1.memconv.c1.op.conv", "upc1.memconv.c2.op.conv"])
    assert len(pk.f32_in) == len(pk.f32_out) == 20
    assert not bsvd_amd.BSVD(wide_conv="direct", **kw).to(_dev())._executor(_dev()).packed.f32_out      # no Winograd reader, nothing handed over
    assert not bsvd_amd.BSVD(f32_handover=False, **kw).to(_dev())._executor(_dev()).packed.f32_out
    # ABI: x_f32 without the Winograd form, y_f32 on a RESID / planar layer
    from bsvd_amd.netspec import ConvSpec
    sp = ConvSpec("l", "l", 128, 128, 1, True, "relu6", 0)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (128, 128, 3, 3)),
                       ("l.bias", (128,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    gex = _exec(_Net(sp), st, "direct")
    a, _ = gex.build_args(sp, torch.zeros(1, 8, 16, 128, device=_dev()))
    lib = _lib.load()
    a.x_f32 = 1
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -21 and b"x_f32" in lib.bsvd_last_error()
    a.x_f32, a.y_f32, a.epilogue, a.extra, a.resid_ch = 0, 1, 2, a.x, 3
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -21 and b"y_f32" in lib.bsvd_last_error()


@pytest.mark.parametrize("blind", [False, True])
def test_whole_network_handover_on_vs_off_and_stream_equals_clip(blind):
    import bsvd_amd
    from oracle import bsvd_oracle as O
    dev = _dev()
    kw = dict(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", pretrain_ckpt=None, precision="f16x3")
    kw.update(dict(act="relu", interm_ch=30, blind=True) if blind else dict(act="relu6", interm_ch=64))
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 30 if blind else 64, blind=blind), 9)
    rs = np.random.RandomState(4)
    x = torch.from_numpy(rs.rand(1, 6, 3 if blind else 4, 40, 56).astype(np.float32))
    cfg = O.default_cfg(act="relu", interm_ch=30, blind=True) if blind else O.default_cfg()
    want = O.bsvd_clip(x, O.to_torch_state(st), cfg)
    ys = {}
    for on in (False, True):
        m = bsvd_amd.BSVD(f32_handover=on, **kw)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        m = m.to(dev).eval()
        ys[on] = m(x.to(dev))
        err = maxabs(ys[on].cpu().numpy(), want.numpy())
        print("blind=%s handover=%s: max-abs vs CPU oracle %.2e (|y| max %.1f)" % (blind, on, err, float(want.abs().max())))
        assert err < 1.5e-4
        m.engine_mode = "stream"
        assert torch.equal(m(x.to(dev)), ys[on])
        m.release_stream_buffers()
    assert maxabs(ys[True].cpu().numpy(), ys[False].cpu().numpy()) < 1.5e-4
