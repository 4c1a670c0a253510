"""Stream schedule on preallocated rings + HIP graphs (bsvd_amd/stream_plan.py) on the MI355X: bit-identical to the clip
schedule and to the allocating stream path, through feedin_one_element and streaming_forward, incl. graph replays."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(precision, **kw):
    import bsvd_amd
    torch.manual_seed(7)
    return bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None,
                         precision=precision, **kw).to("cuda:0").eval()


def _per_frame(m, x):
    outs = [m.feedin_one_element(x[i:i + 1]) for i in range(x.shape[0])]
    assert all(o is None for o in outs[:m.shift_num])
    while len(outs) < x.shape[0] + m.shift_num:
        outs.append(m.feedin_one_element(None))
    assert m.feedin_one_element(None) is None
    m.reset()
    return torch.cat(outs[m.shift_num:])


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_ring_graph_stream_equals_clip_bitwise(precision):
    m = _model(precision)
    for F, H, W in ((1, 32, 48), (2, 32, 48), (3, 32, 48), (7, 32, 48), (37, 64, 96), (23, 64, 96)):
        x = torch.rand(F, 4, H, W, device="cuda:0")
        want = m.clip_forward(x)
        for rep in range(3):             # rep 0: batched launches, rep 1: graph capture, rep 2: graph replay
            assert torch.equal(_per_frame(m, x), want), (F, rep)
            for chunk in (1, "auto", 3):
                m.stream_chunk = chunk
                assert torch.equal(m.streaming_forward(x), want), (F, rep, chunk)
    st = m._stream_eng.stats
    assert st["graph_captures"] > 0 and st["graph_replays"] > st["graph_captures"]


@pytest.mark.parametrize("wide_conv", ["direct", "wino2", "wino6", "wino26"])      # (= engine.WIDE_CONV: the product forms)
def test_every_form_of_the_wide_layers_streams_bit_identically_and_stays_in_the_error_class(wide_conv):
    """BSVD(wide_conv=...): the arithmetic form of the wide layers is a property of the model, so the reference's two call protocols
    (bsvd_arch.py:485-552 clip loop, :555-569 per-frame feed) agree bit for bit in every form -- also where a launch picks another tile
    of the same form (single frames: the 8-row tile; the lagged two-branch step: the full tile) -- and every form is as close to the
    exact-fp32 mode as the direct split kernel (same weights, same clip)."""
    m = _model("f16x3", wide_conv=wide_conv)
    ref = _model("fp32")
    ref.load_state_dict(m.state_dict())
    assert m._executor(torch.device("cuda", 0)).packed.wide_conv == wide_conv
    for F, H, W in ((5, 48, 80), (19, 144, 256)):          # the second: 256-channel single-frame launches on more than one tile row
        x = torch.rand(F, 4, H, W, device="cuda:0")
        want = m.clip_forward(x)
        err = float((want - ref.clip_forward(x)).abs().max())
        print("%s %dx%dx%d: max-abs vs exact fp32 %.2e" % (wide_conv, F, H, W, err))
        assert err < 3e-4
        for rep in range(2):
            assert torch.equal(_per_frame(m, x), want), (F, rep)
            for chunk in (1, "auto"):
                m.stream_chunk = chunk
                assert torch.equal(m.streaming_forward(x), want), (F, rep, chunk)


def test_graph_path_equals_allocating_stream_path_and_no_graph_path():
    x = torch.rand(29, 4, 48, 64, device="cuda:0")
    a = _model("f16x3", stream_rings=False, stream_overlap=False)
    want = _per_frame(a, x)
    assert a._stream_eng is None
    for kw in (dict(), dict(stream_graphs=False), dict(stream_overlap=False), dict(stream_chunk=1), dict(stream_chunk=4, stream_overlap=False)):
        m = _model("f16x3", **kw)
        for _ in range(3):
            assert torch.equal(_per_frame(m, x), want)
            assert torch.equal(m.streaming_forward(x), want)
        if not kw.get("stream_graphs", True):
            assert m._stream_eng.stats["graph_captures"] == 0


def test_frame_size_change_rebuilds_the_rings_and_half_input():
    m = _model("f16x3")
    for H, W in ((32, 48), (48, 32), (32, 48)):
        x = torch.rand(5, 4, H, W, device="cuda:0")
        assert torch.equal(m.streaming_forward(x), m.clip_forward(x))
    xh = torch.rand(5, 4, 32, 48, device="cuda:0").half()
    y = m.streaming_forward(xh)
    assert y.dtype == torch.float16 and torch.equal(y, m.clip_forward(xh))
    m.release_stream_buffers()
    assert m._stream_eng is None and not m._stream_engs


def test_unfinished_stream_then_reset_then_clean_clip():
    m = _model("f16x3")
    x = torch.rand(20, 4, 32, 48, device="cuda:0")
    want = m.clip_forward(x)
    for i in range(11):
        m.feedin_one_element(x[i:i + 1])
    m.reset()
    assert torch.equal(_per_frame(m, x), want)


def test_1080p_stream_reaches_steady_state_and_equals_clip():
    """BASELINE config 5 geometry, F > 16 so that every layer of both DenBlocks is active in the same step."""
    m = _model("f16x3")
    x = torch.rand(20, 4, 1080, 1920, device="cuda:0")
    want = m.clip_forward(x)
    got = _per_frame(m, x)
    assert torch.equal(got, want)
    del got
    for chunk in (1, 2):
        m.stream_chunk = chunk
        assert torch.equal(m.streaming_forward(x), want)
    m.release_stream_buffers()


def test_weight_update_rebuilds_rings_and_graphs():
    """The step plans bake packed-weight addresses into HIP graphs: an in-place parameter update must drop them together
    with the old pack (never replay a graph over freed weights)."""
    m = _model("f16x3")
    x = torch.rand(19, 4, 32, 48, device="cuda:0")
    for _ in range(3):
        y0 = _per_frame(m, x)
    assert torch.equal(y0, m.clip_forward(x)) and m._stream_eng.stats["graph_replays"] > 0
    old = m._stream_eng
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)
    want = m.clip_forward(x)                       # re-packs; the old engines are released with the old pack
    assert m._stream_eng is None and not old.rings
    assert not torch.equal(want, y0)
    for _ in range(3):
        assert torch.equal(_per_frame(m, x), want)
        assert torch.equal(m.streaming_forward(x), want)


def test_prepare_stream_leaves_nothing_to_capture_for_the_live_feed():
    m = _model("f16x3")
    st = m.prepare_stream(32, 48, all_flush_phases=True)
    assert st["graphs"] > 16 + 10 and st["graph_captures"] == st["graphs"]
    before = dict(m._stream_eng.stats)
    x = torch.rand(47, 4, 32, 48, device="cuda:0")       # any length: fill, steady state and flush are all graphs already
    want = m.clip_forward(x)
    assert torch.equal(_per_frame(m, x), want)
    after = m._stream_eng.stats
    assert after["graph_captures"] == before["graph_captures"] and after["batch_launches"] == before["batch_launches"]
    assert after["graph_replays"] - before["graph_replays"] == 47 + 16      # the 17th flush feed finds the pipeline empty: no launch
