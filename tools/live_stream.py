#!/usr/bin/env python3
"""Host-to-host rate and latency of one live 1080p feed (BASELINE config 5 with the PCIe legs included; never bench.py's
`value`): uint8 frames from pinned host memory through bsvd_amd.pipeline.LiveStream (upload, u8->planar, one graph-replayed
pipeline step, planar->u8, download on three HIP streams).   python tools/live_stream.py [--size 1080x1920] [--frames 96]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bsvd_amd
from bsvd_amd.pipeline import LiveStream

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1080x1920")
ap.add_argument("--frames", type=int, default=96)
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--json", default=None)
a = ap.parse_args()
H, W = map(int, a.size.split("x"))
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None,
                  precision=a.precision).to(dev).eval()
frames = np.random.RandomState(0).randint(0, 256, (a.frames, H, W, 3)).astype(np.uint8)
res = {"size": a.size, "frames": a.frames, "precision": a.precision, "rows": []}
for depth, overlap in ((1, False), (2, False), (2, True), (3, True)):
    live = LiveStream(m, sigma=30 / 255.0, depth=depth, overlap_blocks=overlap)
    for rep in range(3):                                  # rep 0/1: plans -> graphs; rep 2 is timed
        lat = []
        t0 = time.perf_counter()
        n_out = 0
        for k in range(a.frames):
            t1 = time.perf_counter()
            r = live.feed(frames[k])
            lat.append(time.perf_counter() - t1)
            n_out += r is not None
        t_feed = time.perf_counter() - t0
        n_out += len(live.flush())
        total = time.perf_counter() - t0
    assert n_out == a.frames
    steady = np.array(lat[live.latency + 1:]) * 1e3
    row = {"depth": depth, "overlap_blocks": overlap, "host_to_host_fps_steady": 1e3 / float(steady.mean()), "whole_feed_fps": a.frames / total,
           "feed_call_ms": {"p50": float(np.percentile(steady, 50)), "p99": float(np.percentile(steady, 99)), "max": float(steady.max())},
           "frame_latency_feeds": live.latency}
    res["rows"].append(row)
    print(json.dumps(row), flush=True)
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)
