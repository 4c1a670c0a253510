#!/usr/bin/env python3
"""BASELINE config 3 geometry: blind bsvd_c64 (3-channel input, interm_ch 30, ReLU) on a [1,10,3,540,960] clip, both modes."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bsvd_amd
dev = torch.device('cuda', 0)
for prec in ('f16x3', 'fp32'):
    torch.manual_seed(0)
    m = bsvd_amd.BSVD(chns=[64,128,256], mid_ch=64, norm='none', act='relu', interm_ch=30, blind=True, pretrain_ckpt=None, precision=prec).to(dev)
    x = torch.rand(1, 10, 3, 540, 960, device=dev)
    with torch.no_grad():
        for _ in range(3): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m(x)
        torch.cuda.synchronize()
    print('blind c64 540x960 %s: %.1f frames/s' % (prec, 100 / (time.perf_counter() - t0)))
