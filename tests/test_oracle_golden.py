"""Pins the CPU oracle (oracle/) to golden vectors computed by the real reference
(tests/golden/make_golden.py ran /root/reference/Experimental_root/archs/bsvd_arch.py on CPU).

Tolerances: the oracle calls the same oneDNN conv as the reference, in a different association
order for the clip formulation -> rounding-level differences only (<= 1e-4 at output magnitude ~10).
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, bsvd_keys, state_for, maxabs
from seeded import seeded_state, state_digest
from oracle import bsvd_oracle as O
from oracle import conv_ref as C

TOL = 1e-4


@pytest.mark.parametrize("C_", [32, 128])
def test_g1_shiftconv(C_):
    g = load_golden("g1_shiftconv_c%d" % C_)
    st = state_for(g, [("conv.weight", (C_, C_, 3, 3)), ("conv.bias", (C_,))])
    fold = C_ // 8
    # C oracle
    y = C.conv3x3(g["center"][0], st["conv.weight"], st["conv.bias"], g["left"][0], g["right"][0][:fold], fold)
    assert maxabs(y, g["out"][0]) < TOL
    # torch oracle: one-frame clip with explicit halos
    P = O.to_torch_state({"k.weight": st["conv.weight"], "k.bias": st["conv.bias"]})
    yt = O.tsm_conv_clip(torch.from_numpy(g["center"]), P, "k",
                         {"prev": torch.from_numpy(g["left"][0]), "next": torch.from_numpy(g["right"][0][:fold])})
    assert maxabs(yt.numpy(), g["out"]) < TOL


@pytest.mark.parametrize("T", [1, 2, 3, 5])
def test_g2_bibuffer(T):
    g = load_golden("g2_bibuffer_T%d" % T)
    st = state_for(g, [("op.conv.weight", (32, 32, 3, 3)), ("op.conv.bias", (32,))])
    P = O.to_torch_state(st)
    buf = O._BiBuffer(P, "op.conv")
    outs, nones = [], []
    for x in [torch.from_numpy(v) for v in g["x"]] + [None, None]:
        y = buf.feed(x)
        nones.append(y is None)
        if y is not None:
            outs.append(y.numpy())
    assert nones == list(g["is_none"])
    assert maxabs(np.stack(outs), g["out"]) < TOL
    # clip formulation gives the same frames
    yc = O.tsm_conv_clip(torch.from_numpy(g["x"][:, 0]), P, "op.conv")
    assert maxabs(yc.numpy(), g["out"][:, 0]) < TOL


@pytest.mark.parametrize("tag,in_ch,out_ch", [("a", 4, 32), ("b", 32, 3)])
def test_g3_denblock_taps(tag, in_ch, out_ch):
    g = load_golden("g3_denblock_" + tag)
    keys = [(k[len("temp1."):], s) for k, s in bsvd_keys([32, 64, 128], out_ch, in_ch, 3, 32)
            if k.startswith("temp1.")]
    st = state_for(g, keys)
    P = O.to_torch_state({"temp1." + k: v for k, v in st.items()})
    taps = {}
    y = O.denblock_clip(torch.from_numpy(g["x"]), P, "temp1.", "relu6", taps=taps)
    for name in ("x0", "x1", "x2", "u2", "u1", "o"):
        assert maxabs(taps[name].numpy(), g[name]) < TOL, name
    assert maxabs(y.numpy(), g["out"]) < TOL
    # streaming DenBlock agrees too
    blk = O._DenBlockStream(P, "temp1.", "relu6")
    outs = [blk.feed(v) for v in [torch.from_numpy(g["x"][i:i + 1]) for i in range(g["x"].shape[0])] + [None] * 9]
    outs = [o for o in outs if o is not None]
    assert maxabs(torch.cat(outs).numpy(), g["out"]) < TOL


def test_g4d_reference_reset_quirk_is_recorded():
    """The oracle reproduces the clean result; the golden documents that the REAL reference, after reset() in the middle
    of an un-flushed stream, does not (stale MemSkip entries) -- the product deliberately deviates (bsvd_amd.BSVD.reset)."""
    g = load_golden("g4d_reset_mid_stream")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    cfg = O.default_cfg(chns=[32, 64, 128], mid_ch=32, interm_ch=32)
    y = O.bsvd_clip(torch.from_numpy(g["x"]), O.to_torch_state(st), cfg)
    assert maxabs(y.numpy(), g["clean"]) < TOL
    assert bool(g["differs"]) and maxabs(g["dirty"], g["clean"]) > 1.0 and not bool(g["again_clean"])


def test_g4c_batch_is_one_long_clip():
    """N = 2: the reference streams the N*F frames as ONE clip (bsvd_arch.py:494-499); recorded from the real forward."""
    g = load_golden("g4c_batch_is_one_clip")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    P = O.to_torch_state(st)
    cfg = O.default_cfg(chns=[32, 64, 128], mid_ch=32, interm_ch=32)
    x = torch.from_numpy(g["x"])
    assert x.shape[0] == 2
    assert maxabs(O.bsvd_clip(x, P, cfg).numpy(), g["out"]) < TOL
    assert maxabs(O.stream_forward(x, P, cfg).numpy(), g["out"]) < TOL


@pytest.mark.parametrize("T", [1, 2, 3, 7])
def test_g4_bsvd_small(T):
    g = load_golden("g4_bsvd_small_T%d" % T)
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    P = O.to_torch_state(st)
    cfg = O.default_cfg(chns=[32, 64, 128], mid_ch=32, interm_ch=32)
    x = torch.from_numpy(g["x"])
    sched = []
    ys = O.stream_forward(x, P, cfg, schedule=sched)
    yc = O.bsvd_clip(x, P, cfg)
    assert int(g["shift_num"]) == 16
    assert [list(map(bool, s)) for s in g["schedule"]] == [list(s) for s in sched]
    assert maxabs(ys.numpy(), g["out"]) < TOL
    assert maxabs(yc.numpy(), g["out"]) < TOL


def test_g4b_defaults_odd_channels():
    g = load_golden("g4b_bsvd_defaults")
    st = state_for(g, bsvd_keys([32, 64, 128], 3, 4, 3, 30))
    P = O.to_torch_state(st)
    cfg = O.default_cfg(chns=[32, 64, 128], mid_ch=3, interm_ch=30, act="relu")
    x = torch.from_numpy(g["x"])
    y = O.bsvd_clip(x[:, :, :3], P, cfg, noise_map=x[:, :, 3:4])
    assert maxabs(y.numpy(), g["out"]) < TOL
    y = O.stream_forward(x, P, cfg)
    assert maxabs(y.numpy(), g["out"]) < TOL


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])     # d: 20 frames > the 16-step latency (pipeline steady state)
def test_g5_bsvd_c64(tag):
    g = load_golden("g5_bsvd_c64_" + tag)
    st = state_for(g, bsvd_keys([64, 128, 256], 64, 4, 3, 64))
    P = O.to_torch_state(st)
    x = torch.from_numpy(g["x"])
    taps = {}
    yc = O.bsvd_clip(x, P, taps=taps)
    assert maxabs(yc.numpy(), g["out"]) < TOL
    if tag == "c":
        for name in ("t1_x0", "t1_x2", "t1_out"):
            assert maxabs(taps[name].numpy(), g[name]) < TOL, name
    if tag in ("c", "d"):
        ys = O.stream_forward(x, P)
        assert maxabs(ys.numpy(), g["out"]) < TOL


def test_g5c_c_oracle_independent_arithmetic():
    """The double-accumulating plain-C conv agrees with the reference (oneDNN fp32) on the real c64 net."""
    g = load_golden("g5_bsvd_c64_c")
    st = state_for(g, bsvd_keys([64, 128, 256], 64, 4, 3, 64))
    y = C.bsvd_clip_c(g["x"], st)
    assert maxabs(y, g["out"]) < TOL


def test_g6_blind_wnet_semantics():
    g = load_golden("g6_blind_c64")
    tsn = seeded_state([(k, tuple(int(v) for v in s.split(","))) for k, s in zip(g["tsn_keys"], g["tsn_shapes"])],
                       int(g["seed"]))
    assert state_digest(tsn) == str(g["digest"])
    P = O.to_torch_state(O.tsn_to_bsvd_keys(tsn))
    cfg = O.default_cfg(interm_ch=30, act="relu", blind=True)
    x = torch.from_numpy(g["x"])
    assert maxabs(O.bsvd_clip(x, P, cfg).numpy(), g["out"]) < TOL
    assert maxabs(O.stream_forward(x, P, cfg).numpy(), g["out"]) < TOL


def test_g7_ckpt_keymap():
    g = load_golden("g7_ckpt_keymap")
    want = {"module." + str(k): str(v) for k, v in zip(g["tsn_keys"], g["bsvd_keys"])}
    got = {}
    for src in want:
        one = O.tsn_to_bsvd_keys({src: 0})
        assert len(one) == 1
        got[src] = next(iter(one))
    assert got == want
    assert len(want) == 64
    # and without the DataParallel prefix
    k0 = str(g["tsn_keys"][6])
    assert next(iter(O.tsn_to_bsvd_keys({k0: 0}))) == str(g["bsvd_keys"][6])


def test_g13_batchnorm_defaults():
    """The reference constructor's true defaults (norm='bn', eval mode): oracle with explicit eval-mode batch_norm."""
    g = load_golden("g13_batchnorm_defaults")
    st = state_for(g, bsvd_keys([32, 64, 128], 3, 4, 3, 30, norm="bn"))
    assert list(st.keys()) == list(g["keys"])
    P = O.to_torch_state(st)
    cfg = O.default_cfg(chns=[32, 64, 128], mid_ch=3, act="relu", interm_ch=30)
    scale = float(np.abs(g["out"]).max())
    y = O.bsvd_clip(torch.from_numpy(g["x"]), P, cfg)
    assert maxabs(y.numpy(), g["out"]) < 1e-5 * scale
    y = O.stream_forward(torch.from_numpy(g["x"]), P, cfg)
    assert maxabs(y.numpy(), g["out"]) < 1e-5 * scale
