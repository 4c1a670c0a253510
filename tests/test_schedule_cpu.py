"""Host logic of the product (netspec layer list, channel padding, clip and stream schedules, checkpoint
key map) driven on CPU through the oracle-backed executor and checked against the reference goldens.
The HIP kernels are not involved here (see test_gpu_*.py for those)."""
import numpy as np
import pytest
import torch

from helpers import load_golden, bsvd_keys, state_for, maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state, state_digest
from bsvd_amd import checkpoint
from bsvd_amd.netspec import make_netspec, pad16
from bsvd_amd.schedule import StreamPipeline, bsvd_clip, Halo, planar_ok

TOL = 1e-4


def run_clip(net, st, x5, planar=True):
    ex = OracleExecutor(st)
    ex.planar_io = planar
    x = torch.from_numpy(x5).reshape(-1, *x5.shape[2:])
    pin, pout = planar_ok(ex, net)
    assert (pin, pout) == (planar, planar)
    xin = x if pin else ex.to_nhwc(x, net.temp1["inc0"].cin_pad)
    y = bsvd_clip(ex, net, xin, x_planar=pin, y_planar=(net.out_ch, None) if pout else None)
    if not pout:
        y = ex.to_nchw(y, net.out_ch)
    return y.numpy().reshape(x5.shape[0], x5.shape[1], net.out_ch, *x5.shape[3:]), ex


def run_stream(net, st, x5, schedule=None, planar=True):
    ex = OracleExecutor(st)
    ex.planar_io = planar
    pin, pout = planar_ok(ex, net)
    pipe = StreamPipeline(net)
    x = torch.from_numpy(x5).reshape(-1, *x5.shape[2:])
    T = x.shape[0]
    outs = []

    def feed(v):
        xin = None if v is None else (v if pin else ex.to_nhwc(v, net.temp1["inc0"].cin_pad))
        y = pipe.feed(ex, xin, x_planar=pin, y_planar=(net.out_ch, None) if pout else None)
        if schedule is not None:
            schedule.append([v is None, y is None])
        return None if y is None else (y if pout else ex.to_nchw(y, net.out_ch))

    for t in range(T):
        outs.append(feed(x[t:t + 1]))
    while len(outs) < T + pipe.shift_num:
        outs.append(feed(None))
    feed(None)
    y = torch.cat(outs[pipe.shift_num:]).numpy()
    return y.reshape(x5.shape[0], x5.shape[1], net.out_ch, *x5.shape[3:]), pipe


def test_macs_and_shift_num_c64():
    net = make_netspec([64, 128, 256], 64, 4, 3, "relu6", 64)
    assert net.macs_per_frame(540, 960) == 613_619_712_000
    assert net.shift_num == 16
    assert len(net.layers) == 32 and sum(l.tsm for l in net.layers) == 16
    blind = make_netspec([64, 128, 256], 64, 4, 3, "relu", 30, blind=True)
    assert abs(blind.macs_per_frame(540, 960) / 1e9 - 582.4) < 0.1


def test_batch_of_clips_is_one_long_clip():
    g = load_golden("g4c_batch_is_one_clip")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    y, _ = run_clip(net, st, g["x"])
    assert y.shape == g["out"].shape and maxabs(y, g["out"]) < TOL
    y, _ = run_stream(net, st, g["x"])
    assert maxabs(y.reshape(g["out"].shape), g["out"]) < TOL


@pytest.mark.parametrize("T", [1, 2, 3, 7])
def test_small_net_clip_and_stream(T):
    g = load_golden("g4_bsvd_small_T%d" % T)
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    y, ex = run_clip(net, st, g["x"])
    assert ex.launches == 32
    assert maxabs(y, g["out"]) < TOL
    y, ex = run_clip(net, st, g["x"], planar=False)          # generic entry/exit (layout kernels) path
    assert maxabs(y, g["out"]) < TOL
    y, _ = run_stream(net, st, g["x"], planar=False)
    assert maxabs(y, g["out"]) < TOL
    sched = []
    y, pipe = run_stream(net, st, g["x"], sched)
    assert sched == [list(map(bool, s)) for s in g["schedule"]]
    assert maxabs(y, g["out"]) < TOL
    # a completed flush leaves the skip FIFOs empty
    for blk in (pipe.t1, pipe.t2):
        assert len(blk.skip_in) == len(blk.skip_x0) == len(blk.skip_x1) == 0


def test_default_ctor_odd_channels():
    """mid_ch=3, interm_ch=30, fold=4: exercises channel padding (3->16, 30->32) in every role."""
    g = load_golden("g4b_bsvd_defaults")
    st = state_for(g, bsvd_keys([32, 64, 128], 3, 4, 3, 30))
    net = make_netspec()        # reference constructor defaults
    assert (net.temp1["out3"].cout_pad, net.temp2["inc0"].cin_pad, net.temp1["inc0"].cout_pad) == (16, 16, 32)
    y, _ = run_clip(net, st, g["x"])
    assert maxabs(y, g["out"]) < TOL
    y, _ = run_stream(net, st, g["x"])
    assert maxabs(y, g["out"]) < TOL


@pytest.mark.parametrize("tag", ["a", "c", "d"])
def test_c64_clip(tag):
    g = load_golden("g5_bsvd_c64_" + tag)
    st = state_for(g, bsvd_keys([64, 128, 256], 64, 4, 3, 64))
    net = make_netspec([64, 128, 256], 64, 4, 3, "relu6", 64)
    y, _ = run_clip(net, st, g["x"])
    assert maxabs(y, g["out"]) < TOL
    if tag == "d":          # 20 frames: the stream schedule reaches its steady state (results while data still arrives)
        ys, _ = run_stream(net, st, g["x"])
        assert maxabs(ys, g["out"]) < TOL


def test_blind_wnet_semantics_and_tsn_checkpoint_keys():
    g = load_golden("g6_blind_c64")
    tsn = seeded_state([(k, tuple(int(v) for v in s.split(","))) for k, s in zip(g["tsn_keys"], g["tsn_shapes"])],
                       int(g["seed"]))
    assert state_digest(tsn) == str(g["digest"])
    assert checkpoint.is_tsn_schema(tsn)
    st = checkpoint.to_bsvd_state(tsn)
    net = make_netspec([64, 128, 256], 64, 4, 3, "relu", 30, blind=True)
    assert net.net_in_ch == 3 and net.temp2["inc0"].cin == 64
    y, _ = run_clip(net, st, g["x"])
    assert maxabs(y, g["out"]) < TOL


def test_checkpoint_keymap_matches_reference_load():
    g = load_golden("g7_ckpt_keymap")
    for src, dst in zip(g["tsn_keys"], g["bsvd_keys"]):
        assert checkpoint.tsn_key_to_bsvd("module." + str(src)) == str(dst)
        assert checkpoint.tsn_key_to_bsvd(str(src)) == str(dst)
    assert checkpoint.tsn_key_to_bsvd("something.else") is None
    plain = {"module.temp1.inc.convblock.0.weight": 1}
    assert list(checkpoint.to_bsvd_state(plain)) == ["temp1.inc.convblock.0.weight"]


def test_sharded_clip_with_halos_equals_whole_clip():
    """Frame-window sharding: two shards exchanging per-layer 1-frame halos == one clip."""
    g = load_golden("g4_bsvd_small_T7")
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    x = torch.from_numpy(g["x"][0])
    cut = 3
    # run both shards in lock-step, layer by layer, using python generators as "ranks"
    import threading
    boxes = {}
    barrier = threading.Barrier(2)
    results = [None, None]

    def rank(r, frames):
        ex = OracleExecutor(st)

        def halo_fn(sp, v):
            fold = sp.fold
            mine = {"first": ex.halo_pack(v[0], 0, fold), "last": ex.halo_pack(v[-1], fold, fold)}
            boxes[(r, sp.key)] = mine
            barrier.wait()
            other = boxes[(1 - r, sp.key)]
            barrier.wait()
            if r == 0:
                return None, Halo(other["first"], fold, 0)
            return Halo(other["last"], fold, 0), None

        xin = ex.to_nhwc(frames, net.temp1["inc0"].cin_pad)
        results[r] = ex.to_nchw(bsvd_clip(ex, net, xin, halo_fn), net.out_ch)

    th = [threading.Thread(target=rank, args=(0, x[:cut])), threading.Thread(target=rank, args=(1, x[cut:]))]
    [t.start() for t in th]
    [t.join() for t in th]
    y = torch.cat(results).numpy()
    assert maxabs(y, g["out"][0]) < TOL


def test_clip_peak_bytes_bound():
    """engine_mode='auto' sizes the clip schedule from this bound: 85 frames of 1080p fit one MI355X (288 GB), 4K does not."""
    from bsvd_amd.netspec import make_netspec, clip_peak_bytes
    net = make_netspec([64, 128, 256], 64, 4, 3, "relu6", 64)
    b1080 = clip_peak_bytes(net, 85, 1080, 1920)
    assert 150e9 < b1080 < 0.9 * 288e9
    assert clip_peak_bytes(net, 85, 2160, 3840) > 288e9
    assert clip_peak_bytes(net, 10, 540, 960) < 7e9
    # the bound really is an upper bound of what the schedule keeps live (count NHWC floats through an executor spy)
    from oracle_exec import OracleExecutor
    import torch
    from helpers import bsvd_keys
    from seeded import seeded_state
    from bsvd_amd.schedule import bsvd_clip
    small = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    st = seeded_state(bsvd_keys([32, 64, 128], 32, 4, 3, 32), 3)
    ex = OracleExecutor(st)
    live, peak = {}, [0]
    conv = ex.conv

    def spy(sp, x, *a, **k):
        y = conv(sp, x, *a, **k)
        live[id(y)] = (y.untyped_storage().nbytes(), __import__("weakref").ref(y))
        for key in [k_ for k_, (_, r) in live.items() if r() is None]:
            del live[key]
        peak[0] = max(peak[0], sum(n for n, _ in live.values()) + x0_bytes)
        return y

    ex.conv = spy
    T, H, W = 3, 16, 24
    x = torch.randn(T, H, W, 16)
    x[..., 4:] = 0
    x0_bytes = x.numel() * 4
    bsvd_clip(ex, small, x)
    assert 0 < peak[0] <= clip_peak_bytes(small, T, H, W)
    print('live peak %d B, bound %d B' % (peak[0], clip_peak_bytes(small, T, H, W)))


def test_batchnorm_fold_reproduces_the_reference_default_constructor():
    """norm='bn' (the reference's default ctor, bsvd_arch.py:446): the product folds the eval-mode BatchNorm layers into
    the packed conv weights (checkpoint.fold_batchnorm); the folded network through the schedules equals the reference."""
    import bsvd_amd
    from bsvd_amd.netspec import norm_key_after
    g = load_golden("g13_batchnorm_defaults")
    st = state_for(g, bsvd_keys([32, 64, 128], 3, 4, 3, 30, norm="bn"))
    m = bsvd_amd.BSVD(pretrain_ckpt=None)                       # all defaults, like the reference's BSVD()
    assert m.norm == "bn" and sorted(m.state_dict().keys()) == sorted(st.keys())
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    with pytest.raises(RuntimeError, match="eval"):
        m._executor(torch.device("cpu"))                         # train mode: batch statistics cannot be folded
    folded = {k: v.numpy() for k, v in m.eval()._engine_state().items()}
    assert len(folded) == 64 and not any(".b1." in k or "running" in k for k in folded)
    assert norm_key_after("temp1.upc2.convblock.0") is None and norm_key_after("temp2.outc.convblock.3") is None
    assert norm_key_after("temp2.downc1.memconv.c2.op.conv") == "temp2.downc1.memconv.b2"
    scale = float(np.abs(g["out"]).max())
    y, _ = run_clip(m.net, folded, g["x"])
    assert maxabs(y, g["out"]) < 2e-5 * scale
    y, _ = run_stream(m.net, folded, g["x"])
    assert maxabs(y, g["out"]) < 2e-5 * scale
    # an in-place update of a running statistic re-packs (signature covers buffers)
    sig = m._signature()
    m.temp1.inc.convblock["1"].running_mean.add_(0.5)
    assert m._signature() != sig
    with pytest.raises(NotImplementedError):
        bsvd_amd.BSVD(norm="in", pretrain_ckpt=None)


def test_precision_auto_picks_split_when_the_network_admits_it():
    import bsvd_amd
    assert bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None).precision == "f16x3"
    assert bsvd_amd.BSVD(norm="none", pretrain_ckpt=None).precision == "f16x3"          # c32-sized: fold 8 / 16
    odd = bsvd_amd.BSVD(chns=[16, 32, 64], mid_ch=16, norm="none", interm_ch=16, pretrain_ckpt=None)
    assert odd.precision == "fp32" and odd.precision_requested == "auto"                # fold 4 / 8 on 32 / 64 channels... not admitted
    with pytest.raises(ValueError):
        bsvd_amd.BSVD(chns=[16, 32, 64], mid_ch=16, norm="none", interm_ch=16, pretrain_ckpt=None, precision="f16x3")


def test_parameter_swap_invalidates_the_cached_signature_and_bn_eps_is_read_from_the_module():
    """the engine caches its parameter list; replacing a Parameter OBJECT (module.weight = nn.Parameter(...)),
    a buffer or a sub-module must still be noticed -> the cache is keyed by torch's global registration hooks.  And the
    BatchNorm fold takes each module's own eps."""
    import bsvd_amd
    import torch.nn as nn
    m = bsvd_amd.BSVD(chns=[16, 32, 64], mid_ch=16, interm_ch=16, norm="bn", act="relu", pretrain_ckpt=None,
                      engine_mode="clip", precision="fp32").eval()
    s0 = m._signature()
    assert m._signature() == s0
    conv = m.temp1.inc.convblock["0"]
    conv.weight = nn.Parameter(conv.weight.detach().clone() * 2)          # a NEW Parameter object
    s1 = m._signature()
    assert s1 != s0
    bn = m.temp1.inc.convblock["1"]
    bn.register_buffer("running_var", bn.running_var.clone() + 1.0)         # a NEW buffer object
    s2 = m._signature()
    assert s2 != s1
    m.temp2.outc.convblock["0"] = nn.Conv2d(16, 16, 3, padding=1)           # a swapped sub-module
    assert m._signature() != s2
    # per-module eps reaches the fold
    bn.eps = 0.5
    st = m._engine_state()
    w_raw = m.state_dict()["temp1.inc.convblock.0.weight"]
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + 0.5)
    assert torch.allclose(st["temp1.inc.convblock.0.weight"], w_raw * scale[:, None, None, None], rtol=1e-6, atol=1e-7)
    t = bsvd_amd.TSN(net2d_opt=dict(chns=[16, 32, 64], mid_ch=16, interm_ch=16, norm="bn", act="relu")).eval()
    names = dict(t._bsvd_modules())
    assert "temp1.inc.convblock.1" in names and "temp2.upc1.memconv.b2" in names
    assert set(t._engine_state()) == set(st)
