"""Host logic added in round 5, on CPU: the `precision` property, the constructor-only `wide_conv`, which layer
pairs the engine may fuse (engine.pair_fusable) and the fused-pair plumbing of both schedules -- clip and the ring / plan engine -- through the
oracle-backed executor (one executor call per pair, the first conv's ring never allocated, same results as the unfused schedules)."""
import numpy as np
import pytest
import torch

from helpers import bsvd_keys, load_golden, maxabs, state_for
from oracle_exec import OracleExecutor


def _model(**kw):
    import bsvd_amd
    return bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None, **kw)


def test_precision_is_a_validated_property():
    m = _model()
    assert m.precision == "f16x3" and m.precision_requested == "auto"
    m.precision = "fp32"
    assert m.precision == "fp32" and m.precision_requested == "fp32" and m._precision_init == "fp32"
    m.precision = "auto"
    assert m.precision == "f16x3"
    with pytest.raises(ValueError):
        m.precision = "bf16"
    assert m.precision == "f16x3"                  # a refused assignment changes nothing
    import bsvd_amd
    odd = bsvd_amd.BSVD(chns=[32, 40, 128], mid_ch=32, in_ch=4, out_ch=3, norm="none", act="relu", interm_ch=30, pretrain_ckpt=None)
    assert odd.precision == "fp32"                 # fold 5: 'auto' resolves to exact fp32 ...
    with pytest.raises(ValueError, match="fold"):
        odd.precision = "f16x3"                    # ... and the explicit request is refused like in the constructor
    assert odd.precision == "fp32"


def test_wide_conv_is_a_constructor_keyword_only(monkeypatch):
    from bsvd_amd import arch, engine
    monkeypatch.setenv("BSVD_WIDE_CONV", "direct")          # round 4 read this; an env leak changed the arithmetic form behind a model's back
    assert _model().wide_conv == arch.WIDE_CONV_DEFAULT == "wino2"
    assert _model(wide_conv="wino6").wide_conv == "wino6"
    assert engine.WIDE_CONV == ("direct", "wino2", "wino6", "wino26")
    with pytest.raises(ValueError):
        _model(wide_conv="wino3")
    assert _model().fuse_pairs is arch.FUSE_PAIRS_DEFAULT and _model(fuse_pairs=True).fuse_pairs is True


def test_which_pairs_are_fusable():
    from bsvd_amd.engine import pair_fusable
    from bsvd_amd.netspec import make_netspec
    net = make_netspec([64, 128, 256], 64, 4, 3, "relu6", 64)
    f = lambda S, a, b, prec="f16x3": pair_fusable(S[a], S[b], prec)       # noqa: E731
    assert not f(net.temp1, "inc0", "inc3")                  # planar 4-channel entry: that pair is the fused ENTRY (head_fusable)
    assert f(net.temp1, "out0", "out3") and f(net.temp2, "inc0", "inc3") and f(net.temp2, "out0", "out3")
    assert not f(net.temp1, "out0", "out3", "fp32")          # a split-mode kernel
    assert not f(net.temp1, "down0", "d0c1") and not f(net.temp1, "up1", "out0")      # stride 2 / temporal shift / PixelShuffle are not plain convs
    blind = make_netspec([64, 128, 256], 64, 4, 3, "relu", 30, blind=True)
    assert f(blind.temp2, "inc0", "inc3")                    # 64 -> 30 (padded 32) -> 64: one 32-channel pair
    odd = make_netspec([48, 128, 256], 48, 4, 3, "relu6", 48)
    assert not f(odd.temp1, "out0", "out3")                  # 48 mid channels: not whole 32-channel pairs


@pytest.mark.parametrize("T", [3, 7])
def test_both_schedules_with_fused_pairs_equal_the_unfused_ones(T):
    from bsvd_amd.netspec import make_netspec
    from bsvd_amd.schedule import bsvd_clip
    from bsvd_amd.stream_plan import StreamEngine
    g = load_golden("g4_bsvd_small_T%d" % T)
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    x = torch.from_numpy(g["x"][0])
    plain, fused = OracleExecutor(st), OracleExecutor(st, fuse_pairs=True)
    y0 = bsvd_clip(plain, net, x, None, x_planar=True, y_planar=(3, None))
    y1 = bsvd_clip(fused, net, x, None, x_planar=True, y_planar=(3, None))
    assert maxabs(y0.numpy(), g["out"][0]) < 1e-4 and torch.equal(y0, y1)
    assert plain.launches == 32 and fused.launches == 29     # three pairs, one call each
    assert sum("+" in k for k in fused.log) == 3
    # the ring / plan engine: the first conv of a fused pair has no ring, the pair is one recorded launch
    eng = StreamEngine(net, OracleExecutor(st, fuse_pairs=True), x.shape[2], x.shape[3], 4,
                       alloc=lambda shape: torch.zeros(shape, dtype=torch.float32), poison=True)
    for key in (net.temp1["out0"].key, net.temp2["inc0"].key, net.temp2["out0"].key):
        assert key not in eng.rings
    outs = []
    for t in range(T):
        y = eng.feed(x[t:t + 1], (3, None))
        outs.append(None if y is None else y.clone())
    for _ in range(net.shift_num):
        y = eng.feed(None, (3, None))
        outs.append(None if y is None else y.clone())
    assert eng.feed(None, (3, None)) is None
    got = torch.cat([o for o in outs if o is not None])
    assert got.shape == y0.shape and maxabs(got.numpy(), y0.numpy()) < 1e-4 and not bool(torch.isnan(got).any())      # (CPU conv2d: batch-dependent last bits)
