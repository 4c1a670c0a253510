#!/usr/bin/env python3
"""One feedin_one_element call on a tiny frame (64x96): the wall time per feed there (~0.95 ms) is the GPU's serial chain of 32
dependent kernels (15-40 us each even for a single tile), not the host; the cProfile table shows the host's share (~0.1-0.2 ms
per feed: the pipeline state machine, the plan signature, one graph launch) and was what exposed the 0.4 ms parameter-tree walk
per call that _HipNet._signature now avoids by caching the tensor list.  usage: python tools/host_overhead.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
m = bench.build_model(dev, "f16x3")
x = torch.rand(1, 4, 64, 96, device=dev)
with torch.no_grad():
    for rep in range(3):
        for _ in range(80): m.feedin_one_element(x)
        for _ in range(17): m.feedin_one_element(None)
        m.reset()
    for _ in range(40): m.feedin_one_element(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): m.feedin_one_element(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host time per feed %.1f us (enqueue only), %.1f us incl. drain; %d frames/s at 64x96" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6, 2000 / (t2 - t0)))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(500): m.feedin_one_element(x)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
