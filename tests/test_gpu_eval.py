"""End-to-end evaluation harness on the GPU: tools/run_test.py on a synthetic image-folder dataset + a TSN-schema
checkpoint file, checked against the CPU oracle pipeline fed with the same seeded noise."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import yaml

from helpers import bsvd_keys, maxabs
from seeded import seeded_state

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_test_counterpart_end_to_end(tmp_path):
    from PIL import Image
    from oracle import bsvd_oracle as O
    from bsvd_amd import evaluation as E
    rs = np.random.RandomState(5)
    clips = {}
    for clip, n in (("clipA", 6), ("clipB", 4)):
        os.makedirs(tmp_path / "data" / clip)
        frames = rs.randint(0, 256, (n, 30, 50, 3)).astype(np.uint8)
        clips[clip] = frames
        for i in range(n):
            Image.fromarray(frames[i]).save(tmp_path / "data" / clip / ("%05d.png" % i))
    st = seeded_state(bsvd_keys([32, 64, 128], 32, 4, 3, 32), 31)
    # the authors' checkpoints are TSN-schema: write one and load it through BSVD(pretrain_ckpt=...)
    inv = {}
    for k, v in st.items():
        stage = "0" if k.startswith("temp1.") else "1"
        t = k.split(".", 1)[1]
        t = t.replace("downc0.memconv.", "downc0.convblock.3.").replace("downc1.memconv.", "downc1.convblock.3.")
        if t.startswith("upc") and ".convblock.0." in t:
            t = t.replace(".convblock.0.", ".convblock.1.")
        t = t.replace("upc2.memconv.", "upc2.convblock.0.").replace("upc1.memconv.", "upc1.convblock.0.")
        t = t.replace(".op.conv.", ".net.")
        inv["module.base_model.nets_list.%s.%s" % (stage, t)] = torch.from_numpy(v)
    ckpt = tmp_path / "tsn.pth"
    torch.save({"params": inv}, ckpt)
    opt = {"name": "t", "model_type": "DenoisingModel", "num_gpu": 1, "manual_seed": 10,
           "datasets": {"val_1": {"name": "syn30", "type": "ValFolderDataset", "valsetdir": str(tmp_path / "data"),
                                  "num_validation_frames": 5, "valnoisestd": 30}},
           "network_g": {"type": "BSVD", "chns": [32, 64, 128], "mid_ch": 32, "shift_input": False, "norm": "none",
                         "interm_ch": 32, "act": "relu6", "pretrain_ckpt": str(ckpt)},
           "path": {"pretrain_network_g": None, "strict_load_g": True},
           "val": {"temp_psz": -1, "future_buffer_len": 0,
                   "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 2, "test_y_channel": False},
                               "psnr_float": {"type": "calculate_psnr_float", "crop_border": 2, "test_y_channel": False},
                               "ssim": {"type": "calculate_ssim", "crop_border": 2, "test_y_channel": False}}}}
    yml = tmp_path / "opt.yml"
    yml.write_text(yaml.safe_dump(opt))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_test.py"), "-opt", str(yml),
                          "--csv_dir", str(tmp_path / "csv"), "--save_img", str(tmp_path / "vis")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout[out.stdout.index("{\n"):])["syn30"]
    # the reference's per-dataset log line, per-folder per-frame CSV (denoising_model.py:335-359) and save_img naming (:299)
    assert "Validation syn30\n\t # psnr: %.4f" % res["mean"]["psnr"] in out.stdout
    rows = (tmp_path / "csv" / "syn30_clipA.csv").read_text().strip().splitlines()
    assert rows[0] == ",clipA_0,clipA_1,clipA_2" and len(rows) == 1 + 5
    assert abs(np.mean([float(r.split(",")[1]) for r in rows[1:]]) - res["folders"]["clipA"]["psnr"]) < 1e-4
    shots = sorted(os.listdir(tmp_path / "vis" / "syn30" / "clipB"))
    assert shots == ["%08d_t.png" % i for i in range(4)]
    assert np.asarray(Image.open(tmp_path / "vis" / "syn30" / "clipB" / shots[0])).shape == (30, 50, 3)
    # oracle pipeline with the same RNG stream: seed -> model construction (consumes RNG) -> per-clip noise in sorted order
    import bsvd_amd
    torch.manual_seed(10)
    bsvd_amd.BSVD(precision="fp32", chns=[32, 64, 128], mid_ch=32, norm="none", interm_ch=32, act="relu6", pretrain_ckpt=None)
    P = O.to_torch_state(st)
    cfg = O.default_cfg(chns=[32, 64, 128], mid_ch=32, interm_ch=32)
    for clip in ("clipA", "clipB"):
        gt = torch.from_numpy(np.float32(clips[clip][:5].transpose(0, 3, 1, 2) / 255.))[None]
        noise = torch.FloatTensor(gt.size()).normal_(mean=0, std=30 / 255.0)
        lq = F.pad((gt + noise)[0], (0, 2, 0, 2), mode="reflect")[None]
        nm = torch.full((1, lq.shape[1], 1, 32, 52), 30 / 255.0)
        den = O.bsvd_clip(lq, P, cfg, noise_map=nm).clamp(0, 1)[0, :, :, :30, :50]
        want = np.mean([E.calculate_psnr(E.tensor2img(den[f]), E.tensor2img(gt[0, f]), 2) for f in range(den.shape[0])])
        assert abs(res["folders"][clip]["psnr"] - want) < 2e-3, (clip, res["folders"][clip], want)
    assert set(res["mean"]) == {"psnr", "psnr_float", "ssim"}


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_psnr_parity_proxy_within_0p02_db(precision):
    """north_star: "PSNR within 0.02 dB of the reference".  DAVIS/Set8 frames and the trained checkpoint are not
    available offline, so the proxy of SURVEY 8c is pinned instead: on a seeded sigma=30 clip,
    PSNR(engine output, clean) - PSNR(CPU-forward output, clean) for both of the reference's PSNR metrics
    (uint8-domain calculate_psnr with crop_border 2, and calculate_psnr_float) stays far inside 0.02 dB."""
    import bsvd_amd
    from oracle import bsvd_oracle as O
    from bsvd_amd import evaluation as E
    rs = np.random.RandomState(8)
    T, H, W = 4, 96, 128
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    gt = np.stack([np.stack([0.5 + 0.4 * np.sin(0.07 * (xx + 3 * t) + c) * np.cos(0.05 * yy + 0.3 * c) for c in range(3)])
                   for t in range(T)]).astype(np.float32)                        # smooth moving pattern in [0.1, 0.9]
    lq = gt + rs.standard_normal(gt.shape).astype(np.float32) * np.float32(30 / 255.0)
    x = torch.from_numpy(np.concatenate([lq, np.full((T, 1, H, W), 30 / 255.0, np.float32)], axis=1))[None]
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 11)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=64,
                      pretrain_ckpt=None, precision=precision)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
    dev = torch.device("cuda", 0)
    got = torch.clamp(m.to(dev)(x.to(dev)).cpu()[0], 0, 1)
    want = torch.clamp(O.bsvd_clip(x, O.to_torch_state(st))[0], 0, 1)
    clean = torch.from_numpy(gt)
    for f in range(T):
        d_u8 = (E.calculate_psnr(E.tensor2img(got[f]), E.tensor2img(clean[f]), 2)
                - E.calculate_psnr(E.tensor2img(want[f]), E.tensor2img(clean[f]), 2))
        d_fl = (E.calculate_psnr_float(got[f], clean[f], 2)
                - E.calculate_psnr_float(want[f], clean[f], 2))
        assert abs(d_u8) < 0.02 and abs(d_fl) < 0.02, (precision, f, d_u8, d_fl)
        assert abs(d_fl) < 1e-3          # in fact three orders of magnitude inside the budget


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_clip_pipeline_overlapped_transfers_match_the_direct_path(depth):
    """bsvd_amd.pipeline.ClipPipeline (uint8 upload / forward / uint8 download on three streams, pinned ring of
    `depth` slots) returns, in order, exactly the bytes of the synchronous path."""
    import bsvd_amd
    from bsvd_amd.frame_io import frames_to_input, output_to_frames
    from bsvd_amd.pipeline import ClipPipeline
    rs = np.random.RandomState(21)
    dev = torch.device("cuda", 0)
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 11)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=64,
                      pretrain_ckpt=None, precision="f16x3")
    m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
    m = m.to(dev)
    clips = [rs.randint(0, 256, (3 + (i % 2), 64, 96, 3)).astype(np.uint8) for i in range(5)]   # shapes alternate
    sigma = 30 / 255.0
    want = [output_to_frames(m.clip_forward(frames_to_input(torch.from_numpy(c).to(dev), sigma))).cpu().numpy()
            for c in clips]
    pipe = ClipPipeline(m, sigma=sigma, depth=depth)
    got = list(pipe.run(iter(clips)))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.dtype == np.uint8 and g.shape == w.shape and np.array_equal(g, w)
    tickets = [pipe.submit(c) for c in clips[:4]]          # more submissions than slots: older results are retained
    again = list(pipe.results())
    assert all(np.array_equal(a, w) for a, w in zip(again, want[:4])) and len(again) == 4
    assert all(np.array_equal(t.finish(), w) for t, w in zip(tickets, want[:4]))
    with pytest.raises(ValueError):
        pipe.submit(np.zeros((2, 30, 50, 3), np.uint8))     # not a multiple of 4


@pytest.mark.parametrize("depth,overlap", [(1, None), (2, None), (3, None), (2, False), (1, True)])
def test_live_stream_uint8_frames_match_the_clip_path(depth, overlap):
    """bsvd_amd.pipeline.LiveStream: uint8 frames in, one graph-replayed pipeline step per feed, uint8 frames out
    shift_num + depth - 1 feeds later, byte-identical to the clip schedule on the same frames; reusable after flush()."""
    import bsvd_amd
    from bsvd_amd.frame_io import frames_to_input, output_to_frames
    from bsvd_amd.pipeline import LiveStream
    rs = np.random.RandomState(31)
    dev = torch.device("cuda", 0)
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 11)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=64,
                      pretrain_ckpt=None, precision="f16x3")
    m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
    m = m.to(dev)
    sigma = 30 / 255.0
    live = LiveStream(m, sigma=sigma, depth=depth, overlap_blocks=overlap)
    assert live.overlap == ((depth >= 2) if overlap is None else overlap)
    assert live.latency == m.shift_num + depth - 1 + (1 if live.overlap else 0)
    for T in (23, 5):
        frames = rs.randint(0, 256, (T, 64, 96, 3)).astype(np.uint8)
        want = output_to_frames(m.clip_forward(frames_to_input(torch.from_numpy(frames).to(dev), sigma))).cpu().numpy()
        got, first = [], None
        for k in range(T):
            r = live.feed(frames[k])
            if r is not None:
                first = k if first is None else first
                got.append(r)
        if T > live.latency:
            assert first == live.latency                            # network latency + host pipelining (+ 1 with overlapped blocks)
        got += live.flush()
        assert len(got) == T and all(g.dtype == np.uint8 for g in got)
        assert np.array_equal(np.stack(got), want)
    with pytest.raises(ValueError):
        live.feed(np.zeros((30, 50, 3), np.uint8))


def test_validation_through_the_test_pipeline_calls(tmp_path):
    """What basicsr.test_pipeline does with a model (BasicSR/basicsr/test.py:26-41): build_dataset, a batch-1 DataLoader,
    build_model, model.validation(loader, current_iter=name, tb_logger=None, save_img=...) -- on the HIP engine, with the
    per-folder CSVs next to the log file and the PNG dump of val.save_img (denoising_model.py:192-367)."""
    import logging
    from PIL import Image
    import bsvd_amd
    from bsvd_amd import evaluation as E
    rs = np.random.RandomState(9)
    for clip, n in (("a", 3), ("b", 2)):
        os.makedirs(tmp_path / "data" / clip)
        for i in range(n):
            Image.fromarray(rs.randint(0, 256, (30, 50, 3)).astype(np.uint8)).save(tmp_path / "data" / clip / ("%03d.png" % i))
    opt = {"name": "run3", "model_type": "DenoisingModel", "num_gpu": 1, "dist": False, "rank": 0,
           "network_g": {"type": "BSVD", "chns": [32, 64, 128], "mid_ch": 32, "shift_input": False, "norm": "none",
                         "interm_ch": 32, "act": "relu6", "pretrain_ckpt": None},
           "path": {"visualization": str(tmp_path / "vis")},
           "val": {"temp_psz": -1, "save_img": True,
                   "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 2},
                               "psnr_float": {"type": "calculate_psnr_float", "crop_border": 2}}}}
    dopt = {"name": "syn", "type": "ValFolderDataset", "valsetdir": str(tmp_path / "data"), "num_validation_frames": 85,
            "valnoisestd": 30, "phase": "val"}
    bsvd_amd.install(replace=True)
    log = logging.getLogger("basicsr")
    old = list(log.handlers)
    fh = logging.FileHandler(tmp_path / "test_run3.log")
    log.addHandler(fh)
    log.setLevel(logging.INFO)
    try:
        torch.manual_seed(10)
        ds = bsvd_amd.build_dataset(dopt)
        loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
        model = bsvd_amd.build_model(opt)
        assert isinstance(model.net_g, bsvd_amd.BSVD)
        total = model.validation(loader, current_iter=opt["name"], tb_logger=None, save_img=opt["val"]["save_img"])
    finally:
        log.removeHandler(fh)
        fh.close()
        assert log.handlers == old
    assert set(total) == {"psnr", "psnr_float"} and all(np.isfinite(v) for v in total.values())
    text = (tmp_path / "test_run3.log").read_text()
    assert "Validation syn\n\t # psnr: %.4f" % total["psnr"] in text
    rows = (tmp_path / "test_run3a.csv").read_text().strip().splitlines()
    assert rows[0] == ",a_0,a_1" and len(rows) == 4
    assert sorted(os.listdir(tmp_path / "vis" / "syn" / "b")) == ["%08d_run3.png" % i for i in range(2)]
    # the same numbers through the stand-alone evaluate() with the same noise realisation
    torch.manual_seed(10)
    model2 = bsvd_amd.build_model(opt)
    model2.net_g.load_state_dict(model.net_g.state_dict())
    torch.manual_seed(10)
    bsvd_amd.BSVD(**{k: v for k, v in opt["network_g"].items() if k != "type"})      # same RNG consumption as build_model
    per_folder, tot2 = E.evaluate(model2, bsvd_amd.build_dataset(dopt), opt["val"]["metrics"])
    assert abs(tot2["psnr"] - total["psnr"]) < 1e-4 and set(per_folder) == {"a", "b"}
    img = np.asarray(Image.open(tmp_path / "vis" / "syn" / "a" / "00000000_run3.png"))
    assert img.shape == (30, 50, 3)


def test_live_stream_falls_back_when_the_ring_engine_is_unavailable():
    """LiveStream's default (overlap_blocks=None -> on for depth >= 2) must not break a model whose ring engine cannot
    be built (stream_rings=False here; a ring allocation that hits OOM takes the same path): it runs the plain per-frame feed with
    one feed less latency and the same bytes; an EXPLICIT overlap_blocks=True raises before the stream is touched."""
    import bsvd_amd
    from bsvd_amd.frame_io import frames_to_input, output_to_frames
    from bsvd_amd.pipeline import LiveStream
    rs = np.random.RandomState(33)
    dev = torch.device("cuda", 0)
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 11)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu6", interm_ch=64,
                      pretrain_ckpt=None, precision="f16x3", stream_rings=False)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
    m = m.to(dev)
    sigma = 30 / 255.0
    frames = rs.randint(0, 256, (19, 32, 48, 3)).astype(np.uint8)
    want = output_to_frames(m.clip_forward(frames_to_input(torch.from_numpy(frames).to(dev), sigma))).cpu().numpy()
    live = LiveStream(m, sigma=sigma, depth=2)
    assert live.overlap and live.latency == m.shift_num + 2 and not live.latency_final
    got = [r for r in (live.feed(f) for f in frames) if r is not None]
    assert not live.overlap and live.latency == m.shift_num + 1 and live.latency_final     # decided at the first frame
    got += live.flush()
    assert len(got) == 19 and np.array_equal(np.stack(got), want)
    # flush() re-opens the decision for the next stream (the free HBM or the frame size may have changed) ...
    assert live.overlap and not live.latency_final
    got2 = [r for r in (live.feed(f) for f in frames) if r is not None] + live.flush()
    assert np.array_equal(np.stack(got2), want)
    # ... a frame shape at construction makes the latency final before the first feed ...
    early = LiveStream(m, sigma=sigma, depth=2, frame_shape=frames.shape[1:3])
    assert early.latency_final and not early.overlap and early.latency == m.shift_num + 1
    # ... and an explicit request raises with the stream untouched, AGAIN on a retried feed (the decision stays open)
    strict = LiveStream(m, sigma=sigma, depth=2, overlap_blocks=True)
    for _ in range(2):
        with pytest.raises(RuntimeError, match="ring engine"):
            strict.feed(frames[0])
        assert strict.count == 0 and not strict.inflight and not strict.latency_final      # nothing half-advanced
