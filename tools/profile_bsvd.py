#!/usr/bin/env python3
"""Counterpart of the reference's profile.py (/root/reference/profile.py:55-83 + scripts/profiler.py:32-67) for the
MI355X engine: builds the bsvd_c64 network through the registry, one warm-up forward on a device-resident
randn(1,10,4,540,960) clip, then "10 loops, mean of best 1" wall time per clip and the peak device memory.
(The reference picks a GPU by parsing nvidia-smi; set HIP_VISIBLE_DEVICES instead.)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bsvd_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--size", type=int, nargs=2, default=[540, 960])
ap.add_argument("--half", action="store_true", help="net_g.half() + autocast like profile.py:79-82 (I/O dtype only)")
ap.add_argument("--mode", default="clip", choices=["clip", "stream"])
ap.add_argument("--repeat", type=int, default=10)
ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3"])
ap.add_argument("--trace", default=None, metavar="FILE.json",
                help="also record one forward with torch.profiler (the reference's 'torchprofile19' mode, scripts/profiler.py:85-101) "
                     "and export a chrome trace; the hand-written kernels show up under their own names")
args = ap.parse_args()

bsvd_amd.install(replace=True)
name = "BSVD"
net = bsvd_amd.build_network(dict(type=name, chns=[64, 128, 256], mid_ch=64, shift_input=False, in_ch=4, out_ch=3,
                                  norm="none", act="relu6", interm_ch=64, blind=False, pretrain_ckpt=None,
                                  engine_mode=args.mode, precision=args.precision)).cuda().eval()
inp = torch.randn(1, args.frames, 4, *args.size).cuda()
if args.half:
    net, inp = net.half(), inp.half()
print("size of tensor", tuple(inp.shape), "device", torch.cuda.get_device_name(0))
with torch.no_grad():
    out = net(inp)
    print("output shape is", tuple(out.shape))
    best = float("inf")
    for _ in range(args.repeat):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net(inp)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
print("%d loops, mean of best 1: %.6f sec per loop  (%.1f frames/s)" % (args.repeat, best, args.frames / best))
print("max memory required \t\t %.2fGB" % (torch.cuda.max_memory_allocated() / 1024 ** 3))
if args.trace:
    from torch.profiler import ProfilerActivity, profile
    with torch.no_grad(), profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        net(inp)
        torch.cuda.synchronize()
    prof.export_chrome_trace(args.trace)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8))
