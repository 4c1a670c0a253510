#!/usr/bin/env python3
"""How much does kernel-level concurrency add in stream mode?  N independent BSVD streams (one model, one HIP stream, one
Python thread each) vs one: if the aggregate rate barely moves, the frames=1 launches already fill the GPU and pipelining
temp1/temp2 of consecutive frames on two HIP streams cannot help.  usage: python tools/concurrent_streams.py [frames=20]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(n_streams, frames, precision):
    dev = torch.device("cuda", 0)
    models = [bench.build_model(dev, precision) for _ in range(n_streams)]
    lq, nm = bench.synth_clip(frames, 100, dev)
    x = torch.cat([lq, nm], dim=2)[0].contiguous()
    streams = [torch.cuda.Stream() for _ in range(n_streams)]

    def work(i, reps):
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(reps):
                models[i].streaming_forward(x)
            streams[i].synchronize()

    for reps in (1, 3):                     # warm-up, then timed
        th = [threading.Thread(target=work, args=(i, reps)) for i in range(n_streams)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return n_streams * 3 * frames / dt


if __name__ == "__main__":
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for prec in ("f16x3",):
        for n in (1, 2, 3):
            print("%s: %d concurrent stream(s): %.1f frames/s aggregate" % (prec, n, run(n, frames, prec)))
