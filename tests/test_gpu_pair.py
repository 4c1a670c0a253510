"""The fused 64-channel conv pair (BsvdConvArgs.pre_w_packed, conv3x3_mfma.hip `pre_pair`): OutputCvBlock's /
InputCvBlock's two plain convs (bsvd_arch.py:194-226, 287-306) as ONE launch whose tiles compute the first conv on their own 18 x 18 patch.
Per output both convs run the stand-alone kernels' arithmetic, so the claim under test is BIT equality with the two launches -- for the
PLAIN pair, the RESID pair (DenBlock 1's exit, planar residual base), the planar-output pair (network exit, split16 residual base), on ragged
sizes -- plus the usual tolerance against the double-accumulating CPU oracle, the whole network with the knob on (clip == off, stream ==
clip) and the ABI's refusals."""
import ctypes
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import bsvd_keys, maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state
from test_gpu_f16x3 import from_split, to_split

pytestmark = pytest.mark.gpu
TIGHT = 2e-4


class _PairNet:
    """edge layer, the pair a -> b, edge layer: PackedNet packs a and b as ordinary split layers and finds the pair in `temp1`"""

    def __init__(self, a, b):
        from bsvd_amd.netspec import ConvSpec
        self.layers = [ConvSpec("e0", "e0", 4, 16, 1, False, "none", 0), a, b, ConvSpec("e1", "e1", 16, 3, 1, False, "none", 2)]
        self.temp1 = OrderedDict(out0=a, out3=b)
        self.temp2 = None


def _setup(ca, cm, cb, act_a, act_b, epi_b, seed=5):
    from bsvd_amd.engine import HipExecutor, PackedNet
    from bsvd_amd.netspec import ConvSpec
    a = ConvSpec("out0", "a", ca, cm, 1, False, act_a, 0)
    b = ConvSpec("out3", "b", cm, cb, 1, False, act_b, epi_b)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("a.weight", (cm, ca, 3, 3)), ("a.bias", (cm,)),
                       ("b.weight", (cb, cm, 3, 3)), ("b.bias", (cb,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], seed)
    tst = {k: torch.as_tensor(v) for k, v in st.items()}
    net = _PairNet(a, b)
    dev = torch.device("cuda", 0)
    fused = HipExecutor(PackedNet(net, tst, dev, "f16x3", "direct", fuse_pairs=True))
    plain = HipExecutor(PackedNet(net, tst, dev, "f16x3", "direct", fuse_pairs=False))
    assert fused.fuse_pair(net.temp1, "out0", "out3") and not plain.fuse_pair(net.temp1, "out0", "out3")
    return a, b, st, fused, plain


SIZES = [(1, 16, 16), (2, 10, 19), (1, 33, 50), (3, 5, 7), (1, 48, 64), (2, 1, 1)]


@pytest.mark.parametrize("T,H,W", SIZES)
@pytest.mark.parametrize("ca,cm,cb,act", [(64, 64, 64, "relu6"), (64, 32, 64, "relu"), (32, 64, 48, "relu6")])
def test_plain_pair_equals_two_launches_bitwise_and_the_oracle(ca, cm, cb, act, T, H, W):
    a, b, st, fused, plain = _setup(ca, cm, cb, act, act, 0)
    rs = np.random.RandomState(T * 100 + H)
    x = torch.from_numpy((rs.rand(T, H, W, ca) * 3 - 0.5).astype(np.float32))
    xs = to_split(x).cuda()
    fused.record_variants = True
    y1 = fused.conv_pair_fused(a, b, xs)
    assert "[fused pair]" in fused.last_variant, fused.last_variant
    y2 = plain.conv(b, plain.conv(a, xs))
    assert torch.equal(y1, y2)
    oex = OracleExecutor(st, double=True)
    xq = from_split(to_split(x))
    mid = from_split(to_split(oex.conv(a, xq).float()))          # the tensor between the convs is carried as fp16 pairs in both forms
    want = oex.conv(b, mid)
    err = maxabs(from_split(y1.cpu()).numpy(), want.numpy())
    print("pair %d->%d->%d %s %s: max-abs %.2e (|y| max %.1f)" % (ca, cm, cb, act, (T, H, W), err, float(want.abs().max())))
    assert err < TIGHT


@pytest.mark.parametrize("T,H,W", [(2, 10, 19), (1, 33, 50), (1, 16, 16)])
def test_resid_pair_with_a_planar_base_equals_two_launches(T, H, W):
    """DenBlock 1's OutputCvBlock in clip mode: out3 subtracts from the block's planar 4-channel input (none_minus, bsvd_arch.py:408-414)"""
    a, b, st, fused, plain = _setup(64, 64, 64, "relu6", "none", 2)
    rs = np.random.RandomState(H)
    xs = to_split(torch.from_numpy((rs.rand(T, H, W, 64) * 3).astype(np.float32))).cuda()
    base = torch.from_numpy(rs.rand(T, 4, H, W).astype(np.float32)).cuda()
    kw = dict(extra=base, extra_pstride=1, extra_cstride=H * W)
    y1 = fused.conv_pair_fused(a, b, xs, **kw)
    y2 = plain.conv(b, plain.conv(a, xs), **kw)
    assert torch.equal(y1, y2)
    assert float((from_split(y1.cpu())[..., :3] - from_split(y2.cpu())[..., :3]).abs().max()) == 0.0


@pytest.mark.parametrize("T,H,W", [(2, 10, 19), (1, 33, 50), (1, 16, 16), (1, 4, 4)])
def test_exit_pair_planar_output_equals_two_launches(T, H, W):
    """the network exit: out0 -> out3 (64 -> 3, planar fp32 output, residual against the split16 input of DenBlock 2, clamp)"""
    a, b, st, fused, plain = _setup(64, 64, 3, "relu6", "none", 2)
    rs = np.random.RandomState(W)
    xs = to_split(torch.from_numpy((rs.rand(T, H, W, 64) * 3).astype(np.float32))).cuda()
    base = to_split(torch.from_numpy(rs.rand(T, H, W, 64).astype(np.float32))).cuda()
    for clamp in (None, (0.0, 1.0)):
        kw = dict(extra=base, extra_pstride=64, extra_cstride=1, y_planar=(3, clamp))
        fused.record_variants = True
        y1 = fused.conv_pair_fused(a, b, xs, **kw)
        assert "[fused pair]" in fused.last_variant and "[planar out]" in fused.last_variant, fused.last_variant
        y2 = plain.conv(b, plain.conv(a, xs), **kw)
        assert y1.shape == (T, 3, H, W) and torch.equal(y1, y2)


@pytest.mark.parametrize("blind", [False, True])
def test_whole_network_with_fused_pairs_is_bit_identical_in_every_schedule(blind):
    import bsvd_amd
    dev = torch.device("cuda", 0)
    kw = dict(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", pretrain_ckpt=None, precision="f16x3")
    kw.update(dict(act="relu", interm_ch=30, blind=True) if blind else dict(act="relu6", interm_ch=64))
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 30 if blind else 64, blind=blind), 9)
    models = {}
    for fp in (False, True):
        m = bsvd_amd.BSVD(fuse_pairs=fp, **kw)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        models[fp] = m.to(dev).eval()
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.rand(1, 5, 3 if blind else 4, 36, 52).astype(np.float32)).to(dev)
    y0 = models[False](x)
    y1 = models[True](x)
    ex = models[True]._executor(dev)
    assert sorted(k.split(".", 1)[1] for k in ex.packed.pairs) == ["inc.convblock.3", "outc.convblock.3", "outc.convblock.3"]
    assert torch.equal(y0, y1)
    models[True].engine_mode = "stream"
    assert torch.equal(models[True](x), y1)
    for _ in range(3):                                  # the per-frame API on rings + graphs (third pass: graph replays)
        outs = [models[True].feedin_one_element(x[0, i:i + 1]) for i in range(x.shape[1])]
        outs += [models[True].feedin_one_element(None) for _ in range(models[True].shift_num)]
        assert models[True].feedin_one_element(None) is None
        models[True].reset()
        assert torch.equal(torch.cat([o for o in outs if o is not None]), y1[0])
    models[True].release_stream_buffers()


def test_fused_pair_refuses_what_it_cannot_run():
    from bsvd_amd import _lib
    a, b, st, fused, plain = _setup(64, 64, 64, "relu6", "relu6", 0)
    xs = to_split(torch.zeros(1, 8, 16, 64)).cuda()
    args, _ = fused.build_args(b, xs, pre=a)
    lib = _lib.load()
    for field, val, word in (("stride", 2, b"stride"), ("fold", 16, b"fold"), ("pre_cin", 24, b"pre_cin"), ("dtype", _lib.BSVD_F32, b"BSVD_F16X3")):
        old = getattr(args, field)
        setattr(args, field, val)
        rc = lib.bsvd_conv3x3(ctypes.byref(args), None)
        assert rc == -20 and word in lib.bsvd_last_error(), (field, rc, lib.bsvd_last_error())
        setattr(args, field, old)
    assert lib.bsvd_conv3x3(ctypes.byref(args), None) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("cm", [96, 128])
def test_fused_pair_refuses_a_middle_tensor_wider_than_two_channel_pairs(cm):
    """the kernel carries TWO 32-channel pairs of the first conv's output; with cm = 96 / 128 chunks 4.. of the
    second conv's K would be refilled with pair 1's data.  The engine does not pair such layers, and the ABI refuses them with -20."""
    from bsvd_amd import _lib
    from bsvd_amd.engine import HipExecutor, PackedNet, pair_fusable
    from bsvd_amd.netspec import ConvSpec
    a = ConvSpec("out0", "a", 64, cm, 1, False, "relu6", 0)
    b = ConvSpec("out3", "b", cm, 64, 1, False, "relu6", 0)
    assert not pair_fusable(a, b, "f16x3")
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("a.weight", (cm, 64, 3, 3)), ("a.bias", (cm,)),
                       ("b.weight", (64, cm, 3, 3)), ("b.bias", (64,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 5)
    net = _PairNet(a, b)
    ex = HipExecutor(PackedNet(net, {k: torch.as_tensor(v) for k, v in st.items()}, torch.device("cuda", 0), "f16x3", "direct", fuse_pairs=True))
    assert not ex.fuse_pair(net.temp1, "out0", "out3")
    # the ABI asked explicitly: a valid 64 -> 64 -> 64 request with the middle width raised (validation precedes any launch)
    a64, b64, _, fused, _ = _setup(64, 64, 64, "relu6", "relu6", 0)
    xs = to_split(torch.zeros(1, 8, 16, 64)).cuda()
    args, _ = fused.build_args(b64, xs, pre=a64)
    args.Cin = cm
    lib = _lib.load()
    rc = lib.bsvd_conv3x3(ctypes.byref(args), None)
    assert rc == -20 and b"Cin = 32 or 64" in lib.bsvd_last_error(), (rc, lib.bsvd_last_error())
