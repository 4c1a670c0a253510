#!/usr/bin/env python3
"""The c32-sized network (chns = [32, 64, 128], the other size the reference trains) on a [1,10,4,540,960] clip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bsvd_amd

dev = torch.device("cuda", 0)
for prec in ("f16x3", "fp32"):
    for mode in ("clip", "stream"):
        torch.manual_seed(0)
        m = bsvd_amd.BSVD(chns=[32, 64, 128], mid_ch=32, norm="none", act="relu6", interm_ch=32, pretrain_ckpt=None,
                          precision=prec, engine_mode=mode).to(dev)
        x = torch.rand(1, 10, 4, 540, 960, device=dev)
        with torch.no_grad():
            for _ in range(3):
                m(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                m(x)
            torch.cuda.synchronize()
        print("c32-sized 540x960 %s %s: %.1f frames/s (%.1f GMAC/frame)" % (prec, mode, 100 / (time.perf_counter() - t0),
                                                                            m.net.macs_per_frame(540, 960) / 1e9))
