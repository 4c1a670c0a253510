"""Concurrency fence.  Round 3 found a packed-fp32 head kernel that was bit-exact alone and wrong in single lanes next
to a split-fp16 MFMA kernel on another stream (never shipped, never root-caused).  The kernels that DO ship were
only guarded by single-shot bitwise tests, while `streaming_forward` runs two graph branches concurrently by default.  Here every shipped
kernel family runs >= 200 times beside a loop of the wide split-fp16 MFMA layer -- in both of its forms, the direct 128-accumulator
tile and the Winograd kernel -- on a second HIP stream, and every output must equal the kernel's own solo result bit for bit; and the
two-branch `streaming_forward` must equal the clip schedule 50 times in a row (bsvd_arch.py:485-552 is the reference's sequential loop:
it has no concurrency to get wrong)."""
import numpy as np
import pytest
import torch

from helpers import bsvd_keys
from seeded import seeded_clip, seeded_state

pytestmark = pytest.mark.gpu
REPS = 200


def _dev():
    return torch.device("cuda", 0)


def _model(precision, wide_conv="auto"):
    import bsvd_amd
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 17)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None,
                      precision=precision, wide_conv=wide_conv)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return m.to(_dev())


def _families(m, ex):
    """(name, thunk) per shipped kernel family of model `m`'s arithmetic mode, each on inputs a real network would hand it"""
    net = m.net
    S = net.temp1
    T, H, W = 2, 128, 192
    x = torch.from_numpy(seeded_clip((1, T, 4, H, W), 3, kind="sigma30"))[0].to(_dev())
    fams = []
    if ex.fuse_head(S):
        a0 = ex.conv_head_fused(S["inc0"], S["inc3"], x)
        fams.append(("fused entry", lambda: ex.conv_head_fused(S["inc0"], S["inc3"], x)))
    else:
        t0 = ex.conv(S["inc0"], x, x_planar=True)
        a0 = ex.conv(S["inc3"], t0)
        fams.append(("head (planar in)", lambda: ex.conv(S["inc0"], x, x_planar=True)))
        fams.append(("64-channel tile", lambda: ex.conv(S["inc3"], t0)))
    a1 = ex.conv(S["down0"], a0)                                  # stride 2, 64 -> 128
    fams.append(("stride-2 tile", lambda: ex.conv(S["down0"], a0)))
    a2 = ex.conv(S["d0c1"], a1)                                   # temporal fusion 128 -> 128 (the wide form of the mode)
    fams.append(("wide temporal-fusion layer", lambda: ex.conv(S["d0c1"], a1)))
    # (every tensor goes where the network sends it: since round 5 a tensor only Winograd-form layers read is plain fp32, the skip
    #  tensor x1 = downc0.c2's output stays an fp16-pair tensor)
    x1 = ex.conv(S["d0c2"], a2)
    a3 = ex.conv(S["down1"], x1)
    a4 = ex.conv(S["u2c2"], ex.conv(S["u2c1"], ex.conv(S["d1c2"], ex.conv(S["d1c1"], a3))))
    fams.append(("up-conv + PixelShuffle + skip", lambda: ex.conv(S["up2"], a4, extra=x1, extra_pstride=x1.shape[-1])))
    S2 = net.temp2                                                # second DenBlock: NHWC 64-channel input, planar exit
    o0 = ex.conv(S2["out0"], a0)
    fams.append(("64-channel tile (out0)", lambda: ex.conv(S2["out0"], a0)))
    fams.append(("exit (planar out, residual)", lambda: ex.conv(S2["out3"], o0, extra=a0, extra_pstride=a0.shape[-1], extra_cstride=1,
                                                                 y_planar=(net.out_ch, None))))
    return fams


@pytest.mark.parametrize("precision,neighbour", [("f16x3", "direct"), ("f16x3", "wino2"), ("fp32", "direct")])
def test_every_kernel_family_beside_a_split_mfma_neighbour(precision, neighbour):
    m = _model(precision)
    ex = m._executor(_dev())
    # the neighbour: the wide temporal-fusion layer of a split-fp16 model in the given form, many workgroups, on its own stream
    mn = _model("f16x3", neighbour)
    exn = mn._executor(_dev())
    big = torch.rand((4, 135, 240, 128), device=_dev())
    spb = mn.net.temp1["d0c1"]
    exn.record_variants = True
    exn.conv(spb, big)
    assert ("winox_kernel" in exn.last_variant) == (neighbour != "direct"), exn.last_variant
    exn.record_variants = False
    side = torch.cuda.Stream()
    fams = _families(m, ex)
    assert len(fams) >= 6
    torch.cuda.synchronize()
    for name, run in fams:
        ref = run().clone()
        torch.cuda.synchronize()
        bad = 0
        for it in range(REPS):
            with torch.cuda.stream(side):
                exn.conv(spb, big)
            y = run()
            if it % 4 == 3:
                torch.cuda.synchronize()
            if not torch.equal(y, ref):
                bad += 1
        torch.cuda.synchronize()
        print("%-5s beside %-6s  %-34s %d runs, %d mismatching" % (precision, neighbour, name, REPS, bad))
        assert bad == 0, (precision, neighbour, name, bad)


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_two_branch_streaming_forward_equals_clip_50_times(precision):
    m = _model(precision)
    assert m.stream_overlap
    x = torch.from_numpy(seeded_clip((1, 21, 4, 96, 128), 5, kind="sigma30"))[0].to(_dev())
    want = m.clip_forward(x)
    for chunk in (1, 2):
        m.stream_chunk = chunk
        bad = 0
        for _ in range(25):
            got = m.streaming_forward([x[i:i + 1] for i in range(x.shape[0])])
            got = torch.cat(list(got)) if isinstance(got, (list, tuple)) else got
            bad += 0 if torch.equal(got.reshape(want.shape), want) else 1
        print("%s two-branch streaming_forward, %d frame(s) per step: 25 runs, %d mismatching" % (precision, chunk, bad))
        assert bad == 0
    m.release_stream_buffers()
