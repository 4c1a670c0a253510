"""Ring lifetimes and step plans of bsvd_amd.stream_plan.StreamEngine, driven on CPU through the oracle-backed executor:
the ring/plan engine must reproduce the allocating StreamPipeline (and the reference goldens) exactly, for the
per-frame protocol (feed) and for the lagged two-branch schedule of streaming_forward (feed_lagged)."""
import numpy as np
import pytest
import torch

from helpers import load_golden, bsvd_keys, state_for, maxabs
from oracle_exec import OracleExecutor
from seeded import seeded_state, seeded_clip
from bsvd_amd.netspec import make_netspec
from bsvd_amd.schedule import StreamPipeline
from bsvd_amd.stream_plan import StreamEngine, ring_depth, RING_PERIOD


def _engine(net, st, H, W, C, chunk=1):
    ex = OracleExecutor(st)
    # NaN-poisoned rings: a slot read before it was written, or a padded channel never written, shows up as NaN
    return StreamEngine(net, ex, H, W, C, chunk=chunk, alloc=lambda shape: torch.zeros(shape, dtype=torch.float32),
                        poison=True), ex


def _cp(y):
    return None if y is None else y.clone()       # feed() returns a view of the exit ring: copy before the next step


def _chunks(eng, x):
    n = eng.chunk
    return [x[i:i + n] for i in range(0, x.shape[0], n)]


def _run_feed(eng, x, shift):
    ch = _chunks(eng, x)
    outs = [_cp(eng.feed(c, (eng.net.out_ch, None))) for c in ch]
    while len(outs) < len(ch) + shift:
        outs.append(_cp(eng.feed(None, (eng.net.out_ch, None))))
    assert eng.feed(None, (eng.net.out_ch, None)) is None
    assert all(o is None for o in outs[:shift])
    eng.clear()
    return torch.cat(outs[shift:])


def _run_lagged(eng, x, shift):
    ch = _chunks(eng, x)
    outs = []
    for k in range(len(ch) + shift + 1):
        y = _cp(eng.feed_lagged(list(ch[k].split(1)) if k < len(ch) else None, (eng.net.out_ch, None)))
        if k >= 1:
            outs.append(y)
    eng.feed_lagged(None, (eng.net.out_ch, None), last=True)
    eng.clear()
    return torch.cat([o for o in outs[shift:] if o is not None])


def _run_pipeline(net, st, x):
    ex = OracleExecutor(st)
    pipe = StreamPipeline(net)
    outs = []
    for t in range(x.shape[0]):
        outs.append(pipe.feed(ex, x[t:t + 1], x_planar=True, y_planar=(net.out_ch, None)))
    while len(outs) < x.shape[0] + pipe.shift_num:
        outs.append(pipe.feed(ex, None, x_planar=True, y_planar=(net.out_ch, None)))
    return torch.cat(outs[pipe.shift_num:])


def test_ring_depths_divide_the_period():
    for name in ("inc0", "inc3", "down0", "d0c1", "d0c2", "down1", "d1c1", "d1c2", "u2c1", "u2c2", "up2", "u1c1", "u1c2",
                 "up1", "out0", "out3"):
        for hand in (False, True):
            assert RING_PERIOD % ring_depth(name, hand) == 0
            assert RING_PERIOD % ring_depth(name, False, hand) == 0
    assert ring_depth("down0", False) >= 3 and ring_depth("inc3", False) >= 9 and ring_depth("d0c2", False) >= 5
    assert ring_depth("out3", True) == 10


@pytest.mark.parametrize("T", [1, 2, 3, 7])
def test_engine_reproduces_reference_goldens(T):
    g = load_golden("g4_bsvd_small_T%d" % T)
    st = state_for(g, bsvd_keys([32, 64, 128], 32, 4, 3, 32))
    net = make_netspec([32, 64, 128], 32, 4, 3, "relu6", 32)
    x = torch.from_numpy(g["x"]).reshape(-1, *g["x"].shape[2:])
    eng, _ = _engine(net, st, x.shape[-2], x.shape[-1], x.shape[1])
    want = g["out"].reshape(-1, *g["out"].shape[2:])
    assert maxabs(_run_feed(eng, x, net.shift_num).numpy(), want) < 1e-4
    assert maxabs(_run_lagged(eng, x, net.shift_num).numpy(), want) < 1e-4
    assert maxabs(_run_feed(eng, x, net.shift_num).numpy(), want) < 1e-4       # engine is reusable after clear()


def test_long_stream_on_rings_equals_allocating_pipeline_bitwise():
    """37 frames: the steady state runs through more than two ring periods; two clips back to back reuse the plans."""
    net = make_netspec([16, 32, 64], 16, 4, 3, "relu6", 16)
    st = seeded_state(bsvd_keys([16, 32, 64], 16, 4, 3, 16), 5)
    x = torch.from_numpy(seeded_clip((1, 37, 4, 8, 12), 6, kind="sigma30"))[0]
    want = _run_pipeline(net, st, x)
    eng, _ = _engine(net, st, 8, 12, 4)
    got = _run_feed(eng, x, net.shift_num)
    assert torch.equal(got, want)
    n_plans = len(eng.plans)
    # fill (16) + steady (10 per DenBlock ring period) + flush patterns; far fewer than the number of steps x 2
    assert n_plans < 2 * (37 + 17)
    got = _run_lagged(eng, x, net.shift_num)
    assert torch.equal(got, want)
    got = _run_feed(eng, x, net.shift_num)
    assert torch.equal(got, want) and len(eng.plans) == n_plans                # second clip: every plan was seen before
    # a shorter clip right after (different flush phase) is still exact
    assert torch.equal(_run_feed(eng, x[:23], net.shift_num), _run_pipeline(net, st, x[:23]))


def test_steady_state_signature_has_the_ring_period():
    net = make_netspec([16, 32, 64], 16, 4, 3, "relu6", 16)
    st = seeded_state(bsvd_keys([16, 32, 64], 16, 4, 3, 16), 5)
    eng, _ = _engine(net, st, 8, 12, 4)
    x = torch.from_numpy(seeded_clip((1, 60, 4, 8, 12), 7, kind="sigma30"))[0]
    for t in range(60):
        eng.feed(x[t:t + 1], (3, None))
    # 16 fill steps with distinct patterns, then 10 steady-state plans per DenBlock (temp1 is steady from step 8 on)
    assert len(eng.plans) <= 16 + 16 + 2 * RING_PERIOD
    before = len(eng.plans)
    x2 = torch.from_numpy(seeded_clip((1, 20, 4, 8, 12), 8, kind="sigma30"))[0]
    for t in range(20):
        eng.feed(x2[t:t + 1], (3, None))
    assert len(eng.plans) == before


@pytest.mark.parametrize("chunk", [2, 3, 8])
def test_chunked_steps_equal_the_frame_by_frame_pipeline_bitwise(chunk):
    """n frames per pipeline step (the last chunk shorter): same 16-step pipeline, same results, for both entry points."""
    torch.manual_seed(0)
    net = make_netspec([16, 32, 64], 16, 4, 3, "relu6", 16)
    st = seeded_state(bsvd_keys([16, 32, 64], 16, 4, 3, 16), 5)
    x = torch.from_numpy(seeded_clip((1, 37, 4, 8, 12), 6, kind="sigma30"))[0]
    want = _run_pipeline(net, st, x)
    eng, _ = _engine(net, st, 8, 12, 4, chunk=chunk)
    # (oneDNN picks batch-size dependent blockings: last-bit differences between a 1-frame and an n-frame conv2d call on
    #  the CPU comparator; the HIP kernel is batch invariant and the GPU twin of this test is bitwise)
    assert maxabs(_run_feed(eng, x, net.shift_num), want) < 2e-5
    assert maxabs(_run_lagged(eng, x, net.shift_num), want) < 2e-5
    assert maxabs(_run_lagged(eng, x[:5], net.shift_num), _run_pipeline(net, st, x[:5])) < 2e-5
    from bsvd_amd.stream_plan import ring_bytes_estimate
    assert ring_bytes_estimate(net, 8, 12, chunk) == eng.ring_bytes


@pytest.mark.parametrize("seed", range(6))
def test_random_lengths_chunks_and_lag_on_cpu(seed):
    """Seeded sweep of (clip length, chunk, entry point) over the ring engine with NaN-poisoned rings: the host logic alone
    (no GPU) must reproduce the allocating pipeline for every combination, also for clips shorter than the pipeline depth."""
    rs = np.random.RandomState(700 + seed)
    net = make_netspec([16, 32, 64], 16, 4, 3, "relu6", 16)
    st = seeded_state(bsvd_keys([16, 32, 64], 16, 4, 3, 16), 5)
    chunk = int(rs.randint(1, 7))
    eng, _ = _engine(net, st, 8, 8, 4, chunk=chunk)
    for clip in range(3):
        T = int(rs.randint(1, 30))
        x = torch.from_numpy(seeded_clip((1, T, 4, 8, 8), 900 + 10 * seed + clip, kind="sigma30"))[0]
        want = _run_pipeline(net, st, x)
        run = _run_lagged if rs.randint(0, 2) else _run_feed
        got = run(eng, x, net.shift_num)
        assert got.shape == want.shape and maxabs(got, want) < 2e-5, (seed, clip, T, chunk, run.__name__)


def test_mixing_the_two_step_protocols_inside_one_stream_is_refused():
    """feed (DenBlock 2 in step) and feed_lagged (one step behind) on the same engine: only after clear()"""
    net = make_netspec([16, 32, 64], 16, 4, 3, "relu6", 16)
    st = seeded_state(bsvd_keys([16, 32, 64], 16, 4, 3, 16), 5)
    ex = OracleExecutor(st)
    eng = StreamEngine(net, ex, 8, 8, 4, chunk=1, alloc=lambda shape: torch.zeros(shape))
    x = torch.zeros((1, 4, 8, 8))
    eng.feed(x, (3, None))
    with pytest.raises(RuntimeError, match="reset"):
        eng.feed_lagged(x, (3, None))
    eng.clear()
    eng.feed_lagged(x, (3, None))
    with pytest.raises(RuntimeError):
        eng.feed(x, (3, None))
