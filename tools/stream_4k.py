#!/usr/bin/env python3
"""4K (2160x3840) through the per-frame streaming API on one MI355X: the stream schedule's rings hold 155 GB -- the clip schedule
of the same 36 frames would need more HBM than the chip has -- and the f16x3 kernels address a frame with 32-bit byte offsets,
which a 4K 64-channel frame (2.12 GB) just fits.  Steady-state frames/s and a bitwise check of the first frames against the
clip schedule on an interior crop (spatial locality, receptive field 96 px).   usage: python tools/stream_4k.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
m = bench.build_model(dev, "f16x3")
H, W, F = 2160, 3840, 36
x = torch.rand(F, 4, H, W, device=dev)
print("free GB before", torch.cuda.mem_get_info()[0] / 1e9)
with torch.no_grad():
    outs = []
    t0 = None
    for rep in range(2):
        outs = []
        for k in range(F + 16):
            if rep == 1 and k == 16:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            y = m.feedin_one_element(x[k:k + 1] if k < F else None)
            if rep == 1 and k == F - 1:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            if y is not None and len(outs) < 2: outs.append(y)
        m.feedin_one_element(None); m.reset()
    torch.cuda.synchronize()
    eng = m._stream_eng
    print("4K per-frame: %.1f frames/s steady (%.1f ms/frame), rings %.1f GB, finite=%s" % ((F - 16) / (t1 - t0), (t1 - t0) / (F - 16) * 1e3, eng.ring_bytes / 1e9, bool(torch.isfinite(outs[0]).all())))
    # spatial locality vs a 1080p crop through the clip schedule (receptive field 96 px): frame 0 depends on frames 0..16
    m.release_stream_buffers()
    y0, x0 = 512, 1024
    crop = x[:20, :, y0:y0 + 512, x0:x0 + 768].contiguous()
    yc = m.clip_forward(crop)[:2]
    RF = 96
    ok = all(torch.equal(yc[i][:, RF:-RF, RF:-RF], outs[i][0][:, y0 + RF:y0 + 512 - RF, x0 + RF:x0 + 768 - RF]) for i in range(2))
    print("4K stream == clip on an interior crop (bitwise):", ok)
