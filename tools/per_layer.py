#!/usr/bin/env python3
"""Per-layer time of one clip forward (HIP events around every launch): which of the 32 layers are furthest from the rate
of their kernel variant.  usage: python tools/per_layer.py [precision=f16x3] [frames=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, prec)
    lq, nm = bench.synth_clip(frames, 100, dev)
    x = torch.cat([lq, nm], dim=2)[0].contiguous()
    ex = model._executor(dev)
    with torch.no_grad():
        for _ in range(3):
            model.clip_forward(x)
        timer = bench.LaunchTimer(ex)
        reps = 5
        for _ in range(reps):
            model.clip_forward(x)
        torch.cuda.synchronize()
        timer.detach()
    per = {}
    for sp, T, Hh, Ww, e0, e1, name, _zs in timer.records:
        flop = 2.0 * sp.macs(Hh, Ww) * T
        if hasattr(sp, "sps"):          # the fused network entry: two layers in one launch, listed under the second one's key
            sp = sp.sps[-1]
        d = per.setdefault(sp.key, [0.0, flop, name, sp, Hh, Ww])
        d[0] += e0.elapsed_time(e1) / reps
    tot = sum(v[0] for v in per.values())
    print("%-34s %-44s %9s %8s %7s" % ("layer", "kernel", "ms", "TFLOP/s", "share"))
    for k, (ms, flop, name, sp, Hh, Ww) in per.items():
        print("%-34s %-44s %9.3f %8.0f %6.1f%%   %d->%d s%d %dx%d%s%s" % (k, name, ms, flop / ms / 1e9, 100 * ms / tot, sp.cin, sp.cout,
              sp.stride, Hh, Ww, " tsm" if sp.tsm else "", " ps" if sp.epilogue == 1 else (" resid" if sp.epilogue == 2 else "")))
    print("total %.2f ms" % tot)


if __name__ == "__main__":
    main()
