#!/usr/bin/env python3
"""BASELINE config 3 as the reference actually runs it: the blind c64 network through the TSN class and denoise_seq with
temp_psz = 11, future_buffer_len = 2 (segments + look-ahead + queued past slices), on a synthetic 85 x 540 x 960 clip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsvd_amd
from bsvd_amd.arch import TSN

dev = torch.device("cuda", 0)
F, H, W = (int(a) for a in (sys.argv[1:4] + ["85", "540", "960"][len(sys.argv) - 1:]))
seq = torch.rand(F, 3, H, W, device=dev)
for prec in ("f16x3", "fp32"):
    torch.manual_seed(0)
    m = TSN(num_segments=11, net2d_opt=dict(chns=[64, 128, 256], mid_ch=64, in_ch=4, out_ch=3, norm="none", act="relu",
                                           interm_ch=30, blind=True), precision=prec).to(dev).eval()
    for psz, fbl in ((11, 2), (-1, 0)):
        with torch.no_grad():
            bsvd_amd.denoise_seq(seq if psz < 0 else seq[:13], None, psz, m, future_buffer_len=fbl)      # warm-up (whole clip at once: its buffers too)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = bsvd_amd.denoise_seq(seq, None, psz, m, future_buffer_len=fbl)
            torch.cuda.synchronize()
        print("TSN blind c64 %dx%dx%d %s temp_psz=%d future_buffer_len=%d: %.1f frames/s" % (F, H, W, prec, psz, fbl, F / (time.perf_counter() - t0)))
