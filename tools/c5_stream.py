#!/usr/bin/env python3
"""BASELINE config 5: bsvd_c64 1080p (1920x1080) streaming through ``feedin_one_element`` on one MI355X.

Protocol (mirrors /root/reference/profile.py:70-83 and Experimental_root/scripts/profiler.py:32-67: device-resident input,
warm-up, synchronised timing), applied to the per-frame streaming API the config names:
  * steady-state throughput: F >= 64 frames fed back to back, one synchronise at the end, counted over the steps in which
    every layer of both DenBlocks is active (steps 16 .. F-1);
  * per-frame latency: the same stream with a device synchronise after every feed (time from handing over frame k to having
    frame k-16 denoised), p50 / p90 / p99 / max over the steady-state steps;
  * the kernel instantiation ("LDS tile configuration") every layer dispatches to at this geometry;
  * bitwise equality of the streamed output with the clip schedule on the first frames.
    python tools/c5_stream.py [--size 1080x1920] [--frames 96] [--precision f16x3] [--json profiles/r02_c5_stream.json]"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bsvd_amd
from bsvd_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1080x1920")
ap.add_argument("--frames", type=int, default=96)
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--json", default=None)
a = ap.parse_args()
H, W = map(int, a.size.split("x"))
F = a.frames
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None,
                  precision=a.precision).to(dev).eval()
g = torch.Generator().manual_seed(0)
gt = torch.rand((F, 3, H, W), generator=g)
x = torch.cat([gt + torch.randn(gt.shape, generator=g) * (30 / 255.0), torch.full((F, 1, H, W), 30 / 255.0)], dim=1).to(dev)
del gt
S = m.shift_num


def stream(sync_each):
    times, outs = [], []
    t_all = time.perf_counter()
    for k in range(F + S):
        t0 = time.perf_counter()
        y = m.feedin_one_element(x[k:k + 1] if k < F else None)
        if sync_each:
            torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        if y is not None and len(outs) < 4:
            outs.append(y)
        if k == S - 1:
            torch.cuda.synchronize(); t_steady0 = time.perf_counter()
        if k == F - 1:
            torch.cuda.synchronize(); t_steady1 = time.perf_counter()
    m.feedin_one_element(None)
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    m.reset()
    return times, outs, (t_steady1 - t_steady0) / (F - S), total


with torch.no_grad():
    stream(False); stream(False)                       # warm-up: plans captured, graphs instantiated
    _, outs, steady_s, total_s = stream(False)
    lat, _, _, _ = stream(True)
    want = m.clip_forward(x[:24])[:4]                  # 16 temporal-fusion layers: frame t depends on frames t-16 .. t+16
lat_steady = np.array(lat[S:F]) * 1e3
eng = m._stream_eng
ex = m._executor(dev)
ex.record_variants = True
variants = {}
h, w = H, W
for blk in (m.net.temp1, m.net.temp2):
    h, w = H, W
    for name, sp in blk.items():
        aargs = _lib.BsvdConvArgs()
        aargs.x = aargs.y = aargs.w_packed = 256
        aargs.extra = 256
        aargs.frames, aargs.H, aargs.W, aargs.Cin, aargs.Cout, aargs.stride = 1, h, w, sp.cin_pad, sp.cout_pad, sp.stride
        aargs.fold, aargs.act, aargs.epilogue, aargs.dtype = sp.fold, _lib.ACT[sp.act], sp.epilogue, ex.dtype
        aargs.resid_ch = 3 if sp.epilogue == 2 else 0
        aargs.w_wino_packed = 256
        aargs.wino_m = ex.packed.wino_layer_abi.get(sp.key, 0)            # the form the pack chose for this layer (0: direct tile)
        aargs.x_f32 = 1 if sp.key in getattr(ex.packed, "f32_in", ()) else 0
        aargs.y_f32 = 1 if sp.key in getattr(ex.packed, "f32_out", ()) else 0
        if sp.key == "temp1.inc.convblock.0":
            aargs.x_planar_ch = sp.cin
        if sp.key == "temp2.outc.convblock.3":
            aargs.y_planar_ch = sp.cout
        buf = ctypes.create_string_buffer(96)
        ex.lib.bsvd_conv3x3_variant(ctypes.byref(aargs), buf, 96)
        variants[sp.key] = "%s @ %dx%d" % (buf.value.decode(), h, w)
        if sp.stride == 2:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        if sp.epilogue == 1:
            h, w = 2 * h, 2 * w
flop = 2.0 * m.net.macs_per_frame(H, W)
res = {
    "config": "BASELINE C5: bsvd_c64 sigma=30, %dx%d, %d frames through feedin_one_element (per-frame API, 16-step latency), "
              "synthetic S2 input resident in HBM, random-init weights" % (W, H, F),
    "precision": a.precision,
    "steady_state_fps": 1.0 / steady_s, "steady_state_ms_per_frame": steady_s * 1e3,
    "whole_stream_fps_incl_fill_and_flush": F / total_s,
    "latency_ms_per_feed_synchronised": {"p50": float(np.percentile(lat_steady, 50)), "p90": float(np.percentile(lat_steady, 90)),
                                         "p99": float(np.percentile(lat_steady, 99)), "max": float(lat_steady.max()),
                                         "mean": float(lat_steady.mean()), "n": int(lat_steady.size)},
    "pipeline_delay_frames": S,
    "algorithmic_tflops": flop / steady_s / 1e12, "flop_per_frame": flop,
    "frac_of_peak": flop / steady_s / 1e12 / (2500.0 if a.precision == "f16x3" else 157.3),
    "first_frames_equal_clip_schedule_bitwise": bool(all(torch.equal(o[0], want[i]) for i, o in enumerate(outs))),
    "engine": dict(eng.stats, ring_GB=round(eng.ring_bytes / 1e9, 2), graphs=sum(1 for g_ in eng.graphs.values() if g_[0]),
                   plans=len(eng.plans)),
    "kernel_variants": variants,
    "device": torch.cuda.get_device_name(0),
}
print(json.dumps(res, indent=1))
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)
