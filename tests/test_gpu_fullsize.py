"""Size-independent properties of the HIP path at BASELINE.json's full geometries, where the CPU oracle would take
minutes: the domain's own invariants stand in for it (the small/medium cases are pinned to the oracle and the reference
goldens in test_gpu_parity.py / test_gpu_f16x3.py).

* clip schedule == stream schedule, bit for bit (whole-clip formulation == the reference's 16-step-latency pipeline,
  bsvd_arch.py:485-552);
* spatial locality: a crop aligned to the two 2x scales gives, beyond the receptive field, exactly the pixels of the
  full frame (exercises tile edges / ragged tiles at positions the small cases never reach);
* temporal locality: 16 temporal-fusion convs => frame t sees frames t-16..t+16 only (bsvd_arch.py:554-560 shift_num);
* frame-window sharding: 8 shards with per-layer halos == the unsharded 80-frame clip, bit for bit.
"""
import threading

import pytest
import torch

from helpers import bsvd_keys
from seeded import seeded_state

pytestmark = pytest.mark.gpu

RF = 96          # > spatial receptive-field radius of the two DenBlocks (2 x 37 px), multiple of 4
PRECISIONS = ["f16x3", "fp32"]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


def _model(precision, mode="clip"):
    import bsvd_amd
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 11)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=64,
                      pretrain_ckpt=None, engine_mode=mode, precision=precision)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
    return m.to(_dev())


def _sigma30_clip(T, H, W, seed):
    """[T,4,H,W] on the device: uniform 'clean' frames + AWGN sigma=30/255 + constant noise map."""
    g = torch.Generator(device=_dev()).manual_seed(seed)
    x = torch.rand((T, 3, H, W), generator=g, device=_dev())
    x += torch.randn((T, 3, H, W), generator=g, device=_dev()) * (30.0 / 255.0)
    return torch.cat([x, torch.full((T, 1, H, W), 30.0 / 255.0, device=_dev())], dim=1).contiguous()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_1080p_stream_equals_clip_and_crop_is_local(precision):
    """BASELINE config 5 (1920x1080 streaming)."""
    m = _model(precision)
    x = _sigma30_clip(3, 1080, 1920, 5)
    y = m.clip_forward(x)
    assert tuple(y.shape) == (3, 3, 1080, 1920) and bool(torch.isfinite(y).all())
    m.engine_mode = "stream"
    assert torch.equal(m(x[None])[0], y)
    y0, y1, x0, x1 = 256, 256 + 512, 1104, 1920          # crop touching the right image edge; x origin a multiple of 8: the Winograd form of the
                                                         # wide layers (F(2,3) along x, default) groups output pixels in pairs from the image origin at
                                                         # every scale, so a crop is bit-identical when its origin keeps that alignment down to 1/4 scale
    yc = m.clip_forward(x[:, :, y0:y1, x0:x1].contiguous())
    inner = yc[:, :, RF:-RF, RF:]                        # right edge is the image edge in both -> no margin there
    assert torch.equal(inner, y[:, :, y0 + RF:y1 - RF, x0 + RF:x1])
    assert not torch.equal(yc[:, :, :4], y[:, :, y0:y0 + 4, x0:x1])   # inside the receptive field the zero pad shows


@pytest.mark.parametrize("precision", PRECISIONS)
def test_davis_sized_clip_temporal_locality(precision):
    """BASELINE config 2 geometry (480x856 is not a multiple of the 16-px tile; 85 frames)."""
    T, H, W = 85, 480, 856
    m = _model(precision)
    x = _sigma30_clip(T, H, W, 6)
    y = m.clip_forward(x)
    m.engine_mode = "stream"
    assert m.stream_overlap                                            # temp1(t+1) || temp2(t) on two HIP streams (default)
    assert torch.equal(m(x[None])[0], y)
    m.stream_overlap = False                                           # the plain one-stream loop over feedin_one_element
    assert torch.equal(m(x[None])[0], y)
    m.stream_overlap = True
    m.engine_mode = "clip"
    t0 = 60
    x2 = x.clone()
    x2[t0, :3] += 0.25
    y2 = m.clip_forward(x2)
    s = m.shift_num
    assert s == 16
    assert torch.equal(y2[:t0 - s], y[:t0 - s])                       # frames that cannot see t0 are untouched ...
    assert torch.equal(y2[t0 + s + 1:], y[t0 + s + 1:])               # ... in both temporal directions
    for t in (t0 - s // 2, t0, t0 + s // 2):
        assert not torch.equal(y2[t], y[t])                           # while the frames around it do change


class _ThreadHalo:
    """`world` lock-stepped frame-window 'ranks' on one GPU; same start()/finish() protocol as dist.HaloExchanger."""

    def __init__(self, ex, r, world, boxes, barrier):
        self.ex, self.r, self.world, self.boxes, self.barrier = ex, r, world, boxes, barrier

    def start(self, sp, v):
        from bsvd_amd.schedule import Halo
        fold, r = sp.fold, self.r
        self.boxes[(r, sp.key)] = (self.ex.halo_pack(v[0], 0, fold), self.ex.halo_pack(v[-1], fold, fold))
        outer = self

        class Pending:
            def finish(self):
                torch.cuda.synchronize()
                outer.barrier.wait()
                left = outer.boxes.get((r - 1, sp.key)) if r > 0 else None
                right = outer.boxes.get((r + 1, sp.key)) if r + 1 < outer.world else None
                outer.barrier.wait()
                return (None if left is None else Halo(left[1], fold, 0),
                        None if right is None else Halo(right[0], fold, 0))
        return Pending()

    def __call__(self, sp, v):
        return self.start(sp, v).finish()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_80_frames_in_8_shards_equal_unsharded(precision):
    """BASELINE config 4 (80-frame 540x960 clip over 8 frame windows), the 8 ranks emulated as threads on one GPU."""
    from bsvd_amd.dist import shard_range
    world, T = 8, 80
    x = _sigma30_clip(T, 540, 960, 7)
    whole = _model(precision).clip_forward(x)
    boxes, results, errors = {}, [None] * world, []
    barrier = threading.Barrier(world)

    def rank(r):
        try:
            torch.cuda.set_device(0)
            m = _model(precision)
            a, b = shard_range(T, world, r)
            halo = _ThreadHalo(m._executor(_dev()), r, world, boxes, barrier)
            results[r] = m.clip_forward(x[a:b], halo)
            torch.cuda.synchronize()
        except Exception as e:          # noqa: BLE001 - surface worker failures instead of dead-locking the barrier
            errors.append(e)
            barrier.abort()

    th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert torch.equal(torch.cat(results), whole)


def test_auto_mode_falls_back_to_stream_when_the_clip_does_not_fit(monkeypatch):
    """engine_mode='auto' (default): clip schedule while its activations fit the free HBM, else the O(1)-memory stream
    schedule -- same bits either way; the bound that drives the choice covers the measured allocator peak."""
    from bsvd_amd.netspec import clip_peak_bytes
    m = _model("f16x3", mode="auto")
    x = _sigma30_clip(12, 480, 856, 9)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    y = m(x[None])[0]
    torch.cuda.synchronize()
    assert m.last_mode == "clip"
    used = torch.cuda.max_memory_allocated() - base
    bound = clip_peak_bytes(m.net, 12, 480, 856)
    print("clip peak %.2f GB, bound %.2f GB" % (used / 1e9, bound / 1e9))
    assert used <= bound
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (1 << 30, 288 << 30))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a, **k: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda *a, **k: 0)
    # the decision is cached per clip geometry (no memory query per call) ...
    assert m(x[None])[0].shape == y.shape and m.last_mode == "clip"
    # ... a cached 'clip' that meets a fuller device falls back on the allocator's OutOfMemoryError ...
    real_clip, calls = m.clip_forward, []

    def oom_once(frames, halo_fn=None):
        calls.append(1)
        raise torch.cuda.OutOfMemoryError("simulated")

    m.clip_forward = oom_once
    y3 = m(x[None])[0]
    m.clip_forward = real_clip
    assert calls and m.last_mode == "stream" and torch.equal(y3, y)
    # ... and the next decision is taken afresh.  The rings the stream schedule left behind count as reusable: with "1 GiB
    # free" + 9 GB of own rings the clip fits again once they are handed back
    assert m._stream_engs
    y4 = m(x[None])[0]
    assert m.last_mode == "clip" and not m._stream_engs and torch.equal(y4, y) and len(calls) == 1
    # nothing to hand back and 1 GiB free: the stream schedule from the start
    m.__dict__.pop("_mode_cache", None)
    y2 = m(x[None])[0]
    assert m.last_mode == "stream" and torch.equal(y2, y) and len(calls) == 1


@pytest.mark.parametrize("precision", PRECISIONS)
def test_blind_c64_at_set8_geometry_vs_oracle_and_schedules(precision):
    """BASELINE config 3 at its real geometry (Set8: 540x960, blind bsvd_c64 = 3-channel input, interm_ch 30 riding on two
    zero padding channels, unbounded ReLU): 2 frames against the torch CPU oracle (WNet semantics), then 21 frames through
    the clip schedule, the per-frame stream API and the chunked streaming_forward, bit for bit."""
    import bsvd_amd
    from oracle import bsvd_oracle as O
    from seeded import seeded_clip
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 30, blind=True), 21)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu", interm_ch=30, blind=True,
                      pretrain_ckpt=None, precision=precision)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
    m = m.to(_dev())
    x = torch.from_numpy(seeded_clip((1, 2, 3, 540, 960), 22, kind="sigma30"))
    want = O.bsvd_clip(x, O.to_torch_state(st), O.default_cfg(act="relu", interm_ch=30, blind=True))
    got = m(x.to(_dev())).cpu()
    err = float((got - want).abs().max())
    print("blind c64 540x960 %s: max-abs vs CPU oracle %.2e (output magnitude %.1f)" % (precision, err, float(want.abs().max())))
    assert err < 1e-3                                     # north_star budget; measured 3e-5 (fp32) / 1e-4 (f16x3)
    xs = _sigma30_clip(21, 540, 960, 23)[:, :3].contiguous()
    y = m.clip_forward(xs)
    outs = [m.feedin_one_element(xs[i:i + 1]) for i in range(21)] + [m.feedin_one_element(None) for _ in range(16)]
    m.feedin_one_element(None)
    m.reset()
    assert torch.equal(torch.cat(outs[16:]), y)
    for chunk in (1, 4):
        m.stream_chunk = chunk
        assert torch.equal(m.streaming_forward(xs), y)
    m.release_stream_buffers()


def test_headline_clip_c1_whole_clip_vs_cpu_oracle_both_precisions(capsys):
    """BASELINE config C1 -- the clip bench.py's headline number is quoted on, [1,10,4,540,960] sigma=30 -- WHOLE, against the
    CPU oracle (the reference's algorithm on torch conv2d fp32, pinned to the reference goldens), in both arithmetic modes.
    north_star budget: 1e-3 max-abs.  About 15 s of host time on the box's 16 usable cores; the number is printed so that it
    shows in the driver's GPU-test log."""
    import os
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import bsvd_oracle as O
    torch.set_num_threads(bench.usable_cores())
    g = torch.Generator().manual_seed(20260928)
    clean = torch.nn.functional.avg_pool2d(torch.rand((10, 3, 540, 960), generator=g), 5, 1, 2)[None]
    lq = clean + torch.randn(clean.shape, generator=g) * (30.0 / 255.0)
    nm = torch.full((1, 10, 1, 540, 960), 30.0 / 255.0)
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 11)
    t0 = time.perf_counter()
    with torch.no_grad():
        want = O.bsvd_clip(lq, O.to_torch_state(st), noise_map=nm)[0]
    t_cpu = time.perf_counter() - t0
    x = torch.cat([lq, nm], dim=2).to(_dev())
    report = {}
    for precision in PRECISIONS:
        m = _model(precision, mode="auto")
        y = m(x)[0]
        torch.cuda.synchronize()
        assert m.last_mode == "clip"
        err = float((y.cpu() - want).abs().max())
        report[precision] = err
        assert tuple(y.shape) == (10, 3, 540, 960)
        assert err < 1e-3, (precision, err)
        assert err < (1.7e-4 if precision == "f16x3" else 1.6e-4), (precision, err)    # regression guard at 2x the measured 8.5e-5 / 7.9e-5
        del m, y
    with capsys.disabled():
        print("\n[C1 whole-clip parity] [1,10,4,540,960] vs CPU oracle (%.1f s on %d threads, |out|max %.2f): "
              "max-abs f16x3 %.3e, exact fp32 %.3e (budget 1e-3)"
              % (t_cpu, torch.get_num_threads(), float(want.abs().max()), report["f16x3"], report["fp32"]))


@pytest.mark.parametrize("config", ["c2", "c3"])
def test_c2_c3_ten_frame_whole_clips_vs_cpu_oracle_both_precisions(config, capsys):
    """BASELINE configs C2 (DAVIS-2017 480p geometry, 480x856, sigma 30) and C3 (Set8 geometry, 540x960, blind bsvd_c64: 3-channel
    input, interm_ch 30, unbounded ReLU -- WNet semantics) as WHOLE 10-frame clips against the CPU oracle in both arithmetic modes,
    printed into the driver's GPU-test log like the C1 test above (the 85-frame clips of the real datasets are
    covered by the size-independent properties: temporal locality, stream == clip).  ~15 s of host time each."""
    import os
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import bsvd_amd
    from oracle import bsvd_oracle as O
    torch.set_num_threads(bench.usable_cores())
    blind = config == "c3"
    h, w = (480, 856) if config == "c2" else (540, 960)
    g = torch.Generator().manual_seed(20260929 + blind)
    clean = torch.nn.functional.avg_pool2d(torch.rand((10, 3, h, w), generator=g), 5, 1, 2)[None]
    lq = clean + torch.randn(clean.shape, generator=g) * (30.0 / 255.0)
    nm = torch.full((1, 10, 1, h, w), 30.0 / 255.0)
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 30 if blind else 64, blind=blind), 12 + blind)
    cfg = O.default_cfg(act="relu", interm_ch=30, blind=True) if blind else O.default_cfg()
    t0 = time.perf_counter()
    with torch.no_grad():
        want = (O.bsvd_clip(lq, O.to_torch_state(st), cfg) if blind else O.bsvd_clip(lq, O.to_torch_state(st), cfg, noise_map=nm))[0]
    t_cpu = time.perf_counter() - t0
    x = (lq if blind else torch.cat([lq, nm], dim=2)).to(_dev())
    report = {}
    for precision in PRECISIONS:
        m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu" if blind else "relu6",
                          interm_ch=30 if blind else 64, blind=blind, pretrain_ckpt=None, precision=precision)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        y = m.to(_dev())(x)[0]
        torch.cuda.synchronize()
        err = float((y.cpu() - want).abs().max())
        report[precision] = err
        assert tuple(y.shape) == (10, 3, h, w) and err < 1e-3, (config, precision, err)
        # regression guard at 2x the measured worst (r04: c2 7.3e-5 / 5.7e-5, c3 blind 9.3e-5 / 8.6e-5 at |out| 20)
        assert err < ((1.9e-4 if precision == "f16x3" else 1.8e-4) if blind else (1.5e-4 if precision == "f16x3" else 1.2e-4)), (config, precision, err)
        del m, y
    with capsys.disabled():
        print("\n[%s whole-clip parity] [1,10,%d,%d,%d] vs CPU oracle (%.1f s on %d threads, |out|max %.2f): max-abs f16x3 %.3e, exact fp32 %.3e (budget 1e-3)"
              % (config.upper(), 3 if blind else 4, h, w, t_cpu, torch.get_num_threads(), float(want.abs().max()), report["f16x3"], report["fp32"]))
