"""fp16 range of the split mode's Winograd form: the kernel converts TRANSFORMED values --
d1 + d2, d0 - d2 reach 2x (F(2,3)) to 4.7x (F(6,3)) the activation range, U = G g 1.5x to 15x the weight range -- and an fp16
conversion that overflows turns a (hi, lo) pair into (inf, NaN).  What guards it (bsvd_internal.h: MODE.FP16_OVFL, saturating
conversions at no instruction cost; bsvd_pack_weights_wino clamps U; engine.PackedNet keeps a layer whose max |G g| leaves fp16's range
on the direct form) is tested here on the MI355X: an unbounded-ReLU layer (the blind network's activation, bsvd_arch.py:185-192) with
activations at 4e4, and BN-folded-sized weights at 5e4, through `wino2` / `wino6` against the double-accumulating oracle and `direct`."""
import warnings

import numpy as np
import pytest
import torch

from helpers import maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state
from test_gpu_f16x3 import _Net, from_split, to_split
from test_gpu_wino import _exec

pytestmark = pytest.mark.gpu


def _layer(cin, cout, act="relu", wscale=1.0, seed=7):
    from bsvd_amd.netspec import ConvSpec
    sp = ConvSpec("l", "l", cin, cout, 1, False, act, 0)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                       ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], seed)
    st["l.weight"] = (st["l.weight"] * wscale).astype(np.float32)
    return sp, st


@pytest.mark.parametrize("form,amp,tol", [("direct", 4.0e4, 5e-5), ("wino2", 3.0e4, 5e-5), ("wino2", 4.0e4, 3e-4), ("wino2", 6.5e4, 1e-3),
                                          ("wino6", 1.3e4, 5e-5), ("wino6", 6.5e4, None)])
def test_relu_layer_with_activations_far_up_fp16s_range(form, amp, tol):
    """Activations of amp on every channel: F(2,3)'s transformed values reach 2 amp -- beyond fp16's 65504 from amp = 32752 on, where an
    unsaturated conversion gives (inf, NaN) and NaN frames.  With saturating conversions hi stops at 65504 and lo = fp16(v - hi) carries the
    rest (up to another 65504, at lo's own 11-bit precision: relative error <= 2^-13 there instead of 2^-22), so
      * while the transformed values stay below 65504 (F(2,3): amp <= 3.2e4, F(6,3): amp <= 1.4e4) nothing changes: the error class of
        every other layer test;
      * F(2,3) over the REST of the activation range (transformed values <= 131008) degrades gracefully -- 1e-4-class relative error;
      * beyond (F(6,3) at amp > 2.8e4) the transformed value saturates: finite, never NaN."""
    rs = np.random.RandomState(1)
    cin = cout = 128
    sp, st = _layer(cin, cout, "relu", wscale=0.02)        # |y| ~ 1e3 .. 3e3: the OUTPUT stays far inside the range
    T, H, W = 2, 12, 20
    x = torch.from_numpy((amp * (0.75 + 0.25 * rs.rand(T, H, W, cin))).astype(np.float32))
    x[..., ::3] *= -1.0
    xq = from_split(to_split(x))
    want = OracleExecutor(st, double=True).conv(sp, xq)
    gex = _exec(_Net(sp), st, form)
    assert ("l" in gex.packed.wino) == (form != "direct")
    got = from_split(gex.conv(sp, to_split(x).cuda()).cpu())
    assert bool(torch.isfinite(got).all()), "%s at %.3g: inf / NaN in the output" % (form, amp)
    rel = maxabs(got.numpy(), want.numpy()) / float(want.abs().max())
    print("%s, activations %.3g: max-abs / max|y| = %.2e (|y| max %.1f)" % (form, amp, rel, float(want.abs().max())))
    if tol is not None:
        assert rel < tol, rel


def test_weights_whose_transform_leaves_fp16s_range_keep_the_direct_form():
    """max |w| = 5e4 passes the raw-weight guard (arch.F16X3_WEIGHT_LIMIT 6e4) but U = G g reaches 1.5 x 5e4 = 7.5e4 for F(2,3) (7.5e5 for
    F(6,3)): the engine keeps such a layer on the direct form (same answer in every schedule: the decision reads the layer's weights only),
    says so once, and the result matches the oracle.  A smaller layer of the same network still takes the Winograd form."""
    from bsvd_amd.engine import HipExecutor, PackedNet
    rs = np.random.RandomState(2)
    sp, st = _layer(128, 128, "relu")
    w = st["l.weight"]
    w[5, 7, 1, :] = (5.0e4, -4.0e4, 3.0e4)
    x = torch.from_numpy((1e-3 * rs.standard_normal((1, 9, 17, 128))).astype(np.float32))
    xq = from_split(to_split(x))
    want = OracleExecutor(st, double=True).conv(sp, xq)
    for form in ("wino2", "wino6"):
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            gex = HipExecutor(PackedNet(_Net(sp), {k: torch.as_tensor(v) for k, v in st.items()}, torch.device("cuda", 0), "f16x3", form))
        assert "l" not in gex.packed.wino and gex.packed.wino_range_fallback and gex.packed.wino_range_fallback[0][0] == "l"
        assert any("direct form" in str(r.message) for r in rec)
        got = from_split(gex.conv(sp, to_split(x).cuda()).cpu())
        assert bool(torch.isfinite(got).all())
        rel = maxabs(got.numpy(), want.numpy()) / float(want.abs().max())
        print("%s -> direct fallback: rel %.2e" % (form, rel))
        assert rel < 5e-5
    # a weight of 3.9e4 x 1.5 = 5.85e4 stays inside: F(2,3) takes the layer and is exact; F(6,3) (x 15) does not take it
    w[5, 7, 1, :] = (3.9e4, -3.0e4, 2.0e4)
    want = OracleExecutor(st, double=True).conv(sp, xq)
    gex = _exec(_Net(sp), st, "wino2")
    assert "l" in gex.packed.wino
    got = from_split(gex.conv(sp, to_split(x).cuda()).cpu())
    assert bool(torch.isfinite(got).all()) and maxabs(got.numpy(), want.numpy()) / float(want.abs().max()) < 5e-5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert "l" not in _exec(_Net(sp), st, "wino6").packed.wino


@pytest.mark.parametrize("m", [2, 6])
def test_wino_pack_saturates_instead_of_packing_inf_nan_pairs(m):
    """the C ABI below the engine's guard: bsvd_pack_weights_wino on a weight whose transform overflows stores +-65504 pairs, never
    (inf, NaN) -- a host without the guard gets a saturated layer, not NaN frames"""
    from bsvd_amd import _lib
    lib = _lib.load()
    cin = cout = 32
    w = torch.zeros(cout, cin, 3, 3)
    w[3, 4, 0, :] = torch.tensor([6.0e4, 6.0e4, 6.0e4])
    w[9, 1, 2, :] = torch.tensor([-6.0e4, 5.0e4, -6.0e4])
    wd = w.cuda()
    n = lib.bsvd_packed_wino_weight_elems(cin, cout, m)
    wp = torch.empty(n, dtype=torch.float32, device="cuda")
    bp = torch.empty(cout, dtype=torch.float32, device="cuda")
    _lib.check(lib.bsvd_pack_weights_wino(wd.data_ptr(), None, cin, cout, cin, cout, 0, m, wp.data_ptr(), bp.data_ptr(), None), "pack")
    torch.cuda.synchronize()
    h = wp.view(torch.float16).float()
    assert bool(torch.isfinite(h).all()) and float(h.abs().max()) == 65504.0
