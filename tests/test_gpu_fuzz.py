"""Seeded random layer configurations through the C ABI vs the double-accumulating oracle executor: shapes, channel
counts, strides, folds, epilogues and halo forms the hand-picked cases of test_gpu_parity.py / test_gpu_f16x3.py do not
enumerate (ragged tiles in both directions, 1-pixel images, folds that force the generic gather, masked output columns)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _draw(rs, split):
    epi = int(rs.choice([0, 0, 0, 1, 2]))
    stride = 1 if epi else int(rs.choice([1, 1, 2]))
    tsm = bool(epi == 0 and stride == 1 and rs.rand() < 0.4)
    act = str(rs.choice(["relu6", "relu", "none"])) if epi == 0 else "none"
    T = int(rs.randint(1, 4))
    H, W = int(rs.randint(1, 41)), int(rs.randint(1, 41))
    if split:
        cin = int(rs.choice([128, 256])) if tsm else int(rs.choice([16, 32, 64, 128, 256]))
        cout = int(rs.choice([64, 128, 256])) if epi == 1 else int(rs.choice([16, 32, 64, 128, 256]))
        if tsm:
            cout = cin
    else:
        cin = int(rs.choice([3, 8, 24, 30, 32, 40, 64, 100, 128]))
        cout = int(rs.choice([64, 128, 192])) if epi == 1 else int(rs.choice([3, 5, 16, 30, 48, 64, 100, 128]))
        if tsm and cin < 8:
            cin = 24
    if max(cin, cout) >= 128:            # keep the CPU oracle quick
        H, W = min(H, 20), min(W, 24)
    return cin, cout, stride, tsm, act, epi, T, H, W


@pytest.mark.parametrize("seed", range(50))
def test_random_layer_exact_fp32(seed):
    import test_gpu_parity as P
    args = _draw(np.random.RandomState(1000 + seed), split=False)
    print("case", args)
    P.test_layer_vs_oracle(*args)


@pytest.mark.parametrize("seed", range(50))
def test_random_layer_split_fp16(seed):
    import test_gpu_f16x3 as S
    args = _draw(np.random.RandomState(2000 + seed), split=True)
    print("case", args)
    S.test_layer_split_vs_oracle(*args)


@pytest.mark.parametrize("seed", range(36))
def test_random_wide_layer_winograd_forms(seed):
    """The Winograd kernels on random eligible layers (stride 1, 128 / 256 input channels; bsvd_arch.py:21-50, :257-267): ragged sizes in
    both directions, 1 to 3 frames, every halo form of the temporal gather, PixelShuffle + skip, the three activations; F(2,3) and F(6,3)
    in turn (the measurement variants draw from the same generator in tests/measure_driver.py)."""
    import test_gpu_wino as WN
    rs = np.random.RandomState(7000 + seed)
    form = WN.PRODUCT_FORMS[seed % 2]
    epi = int(rs.choice([0, 0, 1]))
    tsm = bool(epi == 0 and rs.rand() < 0.6)
    cin = int(rs.choice([128, 256]))
    cout = cin if tsm else int(rs.choice([128, 256, 512] if epi == 1 else [64, 128, 256]))
    act = str(rs.choice(["relu6", "relu", "none"])) if epi == 0 else "none"
    T, H, W = int(rs.randint(1, 4)), int(rs.randint(1, 21)), int(rs.randint(1, 49))
    print("case", form, cin, cout, tsm, act, epi, T, H, W)
    WN.test_wino_layer_vs_oracle(form, cin, cout, tsm, act, epi, T, H, W)


@pytest.mark.parametrize("seed", range(15))
def test_random_clip_whole_network(seed):
    """bsvd_c64 on random small clips (any T >= 1, H and W multiples of 4 from 4 to 48): clip schedule vs the CPU oracle
    in both arithmetic modes, stream schedule bit-identical, blind variant on odd seeds."""
    import torch
    import bsvd_amd
    from helpers import bsvd_keys, maxabs
    from seeded import seeded_state
    from oracle import bsvd_oracle as O
    rs = np.random.RandomState(3000 + seed)
    T, H, W = int(rs.randint(1, 6)), 4 * int(rs.randint(1, 13)), 4 * int(rs.randint(1, 13))
    blind = bool(seed & 1)
    chns, mid = ([64, 128, 256], 64) if seed % 3 else ([32, 64, 128], 32)       # every third case: the c32-sized network
    interm, act, cin = (30, "relu", 3) if blind else (chns[0], "relu6", 4)
    st = seeded_state(bsvd_keys(chns, mid, 4, 3, interm, blind=blind), 40 + seed)
    x = torch.from_numpy(rs.standard_normal((1, T, cin, H, W)).astype(np.float32))
    cfg = O.default_cfg(chns=chns, mid_ch=mid, act=act, interm_ch=interm, blind=blind)
    want = O.bsvd_clip(x, O.to_torch_state(st), cfg)
    dev = torch.device("cuda", 0)
    for precision in ("fp32", "f16x3"):
        m = bsvd_amd.BSVD(chns=chns, mid_ch=mid, in_ch=4, out_ch=3, norm="none", act=act, interm_ch=interm,
                          blind=blind, pretrain_ckpt=None, precision=precision)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        m = m.to(dev)
        y = m(x.to(dev))
        err = maxabs(y.cpu().numpy(), want.numpy())
        print("T=%d %dx%d chns=%s blind=%s %s max-abs %.2e" % (T, H, W, chns, blind, precision, err))
        assert err < 1e-3
        m.engine_mode = "stream"
        assert torch.equal(m(x.to(dev)), y)


@pytest.mark.parametrize("seed", range(8))
def test_random_streams_on_rings_and_graphs(seed):
    """Random clip lengths (1..45 frames, i.e. from 'never reaches the steady state' to several ring periods), frame sizes,
    chunk sizes and lag on/off through the ring/graph engine, three clips back to back on one model (plans of one clip are
    graphs for the next): always bit-identical to the clip schedule; per-frame API and the c32-sized network included."""
    import torch
    import bsvd_amd
    rs = np.random.RandomState(5000 + seed)
    chns, mid = ([64, 128, 256], 64) if seed % 3 else ([32, 64, 128], 32)
    torch.manual_seed(seed)
    m = bsvd_amd.BSVD(chns=chns, mid_ch=mid, norm="none", act="relu6", interm_ch=chns[0], pretrain_ckpt=None,
                      precision="f16x3" if seed % 2 else "fp32").to("cuda:0").eval()
    H, W = 4 * int(rs.randint(2, 10)), 4 * int(rs.randint(2, 10))
    for clip in range(3):
        T = int(rs.randint(1, 46))
        x = torch.rand(T, 4, H, W, device="cuda:0")
        want = m.clip_forward(x)
        m.stream_chunk = [1, 2, 3, 5, 8, "auto"][int(rs.randint(0, 6))]
        m.stream_overlap = bool(rs.randint(0, 2))
        for rep in range(2):
            assert torch.equal(m.streaming_forward(x), want), (seed, clip, T, H, W, m.stream_chunk, m.stream_overlap)
        outs = [m.feedin_one_element(x[i:i + 1]) for i in range(T)] + [m.feedin_one_element(None) for _ in range(m.shift_num)]
        assert m.feedin_one_element(None) is None
        m.reset()
        assert torch.equal(torch.cat([o for o in outs if o is not None]), want)
    m.release_stream_buffers()


def test_random_layers_forced_onto_the_128_accumulator_tile():
    """tests/fat_tile_fuzz_driver.py in a fresh interpreter with BSVD_FAT_MIN_WGS=1: the fat tile's prefetching K loop, its staging-only
    loop for waves below the image and its zero-chunk skip on 24 small random geometries vs the double-accumulating oracle."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BSVD_FAT_MIN_WGS="1", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(here, "fat_tile_fuzz_driver.py"), "24"], env=env, capture_output=True, text=True,
                       timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "FUZZ OK 24" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_random_networks_with_the_wide_layers_forced_onto_the_128_accumulator_tile():
    """... and six whole networks (clip vs oracle in both precisions, stream == clip bit for bit) under the same override."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, BSVD_FAT_MIN_WGS="1", PYTHONDONTWRITEBYTECODE="1")      # ("nets": the driver builds its models with wide_conv='direct')
    r = subprocess.run([sys.executable, os.path.join(here, "fat_tile_fuzz_driver.py"), "6", "nets"], env=env, capture_output=True,
                       text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "FUZZ NETS OK 6" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("seed", range(16))
def test_random_fused_pairs_equal_their_two_launches(seed):
    """The fused conv pair (BsvdConvArgs.pre_w_packed, bsvd_arch.py:194-226, 287-306) on random channel counts (16 .. 64 in, 32 / 64 mid, 16 .. 64 out),
    ragged sizes, 1 to 3 frames, the three activations, PLAIN / RESID epilogues: bit-equal to the two launches."""
    import torch
    import test_gpu_pair as P
    from test_gpu_f16x3 import to_split
    rs = np.random.RandomState(9000 + seed)
    ca, cm, cb = int(rs.choice([16, 32, 48, 64])), int(rs.choice([32, 64])), int(rs.choice([16, 32, 48, 64]))
    act_a = str(rs.choice(["relu6", "relu", "none"]))
    epi = int(rs.choice([0, 0, 2]))
    act_b = "none" if epi == 2 else str(rs.choice(["relu6", "relu", "none"]))
    T, H, W = int(rs.randint(1, 4)), int(rs.randint(1, 40)), int(rs.randint(1, 50))
    a, b, st, fused, plain = P._setup(ca, cm, cb, act_a, act_b, epi, seed=seed)
    xs = to_split(torch.from_numpy((rs.rand(T, H, W, ca) * 4 - 1).astype(np.float32))).cuda()
    kw = {}
    if epi == 2:
        kw = dict(extra=torch.from_numpy(rs.rand(T, 4, H, W).astype(np.float32)).cuda(), extra_pstride=1, extra_cstride=H * W)
    print("case", ca, cm, cb, act_a, act_b, epi, T, H, W)
    assert torch.equal(fused.conv_pair_fused(a, b, xs, **kw), plain.conv(b, plain.conv(a, xs), **kw))


@pytest.mark.parametrize("seed", range(16))
def test_random_wide_layers_with_fp32_handover(seed):
    """The Winograd kernels reading / writing plain fp32 (BsvdConvArgs.x_f32 / y_f32) on the random eligible layers of
    test_random_wide_layer_winograd_forms' generator."""
    import test_gpu_f32_handover as F
    import test_gpu_wino as WN
    rs = np.random.RandomState(7000 + seed)
    form = WN.PRODUCT_FORMS[seed % 2]
    epi = int(rs.choice([0, 0, 1]))
    tsm = bool(epi == 0 and rs.rand() < 0.6)
    cin = int(rs.choice([128, 256]))
    cout = cin if tsm else int(rs.choice([128, 256, 512] if epi == 1 else [64, 128, 256]))
    act = str(rs.choice(["relu6", "relu", "none"])) if epi == 0 else "none"
    T, H, W = int(rs.randint(1, 4)), int(rs.randint(1, 21)), int(rs.randint(1, 49))
    print("case", form, cin, cout, tsm, act, epi, T, H, W)
    F.test_wino_layer_with_fp32_input_and_output_vs_oracle(form, cin, cout, tsm, act, epi, T, H, W)
