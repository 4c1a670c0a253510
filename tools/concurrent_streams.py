#!/usr/bin/env python3
"""A camera wall on one GPU: N independent live streams (one model instance, one HIP stream, one Python thread each), every
stream fed frame by frame through ``feedin_one_element`` (one HIP-graph replay per frame).  The single-frame launches of one
stream run one in-phase round; the launches of the other streams fill its tails.  Prints the aggregate frames/s.
usage: python tools/concurrent_streams.py [frames=60] [HxW=540x960] [json out]"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(n_streams, frames, precision, H, W):
    dev = torch.device("cuda", 0)
    models = [bench.build_model(dev, precision) for _ in range(n_streams)]
    lq, nm = bench.synth_clip(frames, 100, dev, H, W)
    x = torch.cat([lq, nm], dim=2)[0].contiguous()
    streams = [torch.cuda.Stream() for _ in range(n_streams)]

    def work(i, reps):
        m = models[i]
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(reps):
                for k in range(frames):
                    m.feedin_one_element(x[k:k + 1])
                for k in range(m.shift_num + 1):
                    m.feedin_one_element(None)
                m.reset()
            streams[i].synchronize()

    dt = None
    for reps in (2, 3):                     # warm-up (plans -> graphs), then timed
        th = [threading.Thread(target=work, args=(i, reps)) for i in range(n_streams)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for m in models:
        m.release_stream_buffers()
    return n_streams * 3 * frames / dt


if __name__ == "__main__":
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    H, W = map(int, (sys.argv[2] if len(sys.argv) > 2 else "540x960").split("x"))
    rows = []
    for n in (1, 2, 3, 4):
        fps = run(n, frames, "f16x3", H, W)
        rows.append({"streams": n, "aggregate_fps": fps, "per_stream_fps": fps / n})
        print("f16x3 %dx%d: %d concurrent live stream(s): %.1f frames/s aggregate (%.1f per stream)" % (H, W, n, fps, fps / n), flush=True)
    if len(sys.argv) > 3:
        json.dump({"size": "%dx%d" % (H, W), "frames_per_stream": frames, "api": "feedin_one_element (HIP-graph replay per frame)",
                   "rows": rows}, open(sys.argv[3], "w"), indent=1)
