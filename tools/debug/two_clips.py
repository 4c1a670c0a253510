"""Two independent C1 clips at once on two HIP streams (two module instances) against one clip at a time: what kernel-boundary
bubbles (drain of the last round + in-phase start of the next launch) would be worth if something else could fill them.
usage (GPU box): python tools/debug/two_clips.py [seconds=4]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda", 0)
ms = [bench.build_model(dev, precision="f16x3") for _ in range(2)]
xs = [torch.rand((1, 10, 4, 540, 960), device=dev) for _ in range(2)]
ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
for m, x in zip(ms, xs):
    for _ in range(2):
        m(x)
torch.cuda.synchronize()


def run(n_streams):
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(4):
            for i in range(n_streams):
                with torch.cuda.stream(ss[i]):
                    ms[i](xs[i])
                n += 1
        torch.cuda.synchronize()
    return n * 10 / (time.time() - t0)


for k in (1, 2, 1, 2):
    print("%d clip(s) in flight: %.1f frames/s aggregate" % (k, run(k)))
