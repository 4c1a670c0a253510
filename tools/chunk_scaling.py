#!/usr/bin/env python3
"""Time of one split-fp16 layer launch vs its number of 16-channel chunks (Cin) at fixed output shape: the intercept of
the linear fit is the per-tile fixed cost (first patch load latency + epilogue) that the K loop cannot hide.
usage: python tools/chunk_scaling.py [Cout=64] [H=540] [W=960] [frames=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec


def main():
    cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 540
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 960
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    res = []
    for cin in (32, 64, 128, 256):
        class Net:
            pass
        pre = ConvSpec("pre", "pre", 16, cin, 1, False, "relu6", 0)       # produces a valid split16 tensor of cin channels
        sp = ConvSpec("l", "l", cin, cout, 1, False, "relu6", 0)
        post = ConvSpec("post", "post", cout, 16, 1, False, "none", 0)
        net = Net(); net.layers = [pre, sp, post]
        st = {}
        for s in net.layers:
            st[s.key + ".weight"] = torch.from_numpy((rs.standard_normal((s.cout, s.cin, 3, 3)) * (1.5 / np.sqrt(9 * s.cin))).astype(np.float32))
            st[s.key + ".bias"] = torch.zeros(s.cout)
        ex = HipExecutor(PackedNet(net, st, dev, precision="f16x3"))
        x0 = torch.rand((T, 4, H, W), device=dev)
        # first layer = head (planar in) so that the test layer sees a genuine split16 input
        pre4 = ConvSpec("pre", "pre", 4, cin, 1, False, "relu6", 0)
        net.layers[0] = pre4
        st["pre.weight"] = torch.from_numpy((rs.standard_normal((cin, 4, 3, 3)) * 0.2).astype(np.float32))
        ex = HipExecutor(PackedNet(net, st, dev, precision="f16x3"))
        a = ex.conv(pre4, x0, x_planar=True)
        ex.record_variants = True
        y = ex.conv(sp, a)
        name = ex.last_variant
        ex.record_variants = False
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ex.conv(sp, a, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        flop = 2.0 * sp.macs(H, W) * T
        res.append((cin // 16, ms))
        print("%s Cin %3d (%2d chunks) -> Cout %d: %.3f ms  %.0f TFLOP/s algorithmic" % (name, cin, cin // 16, cout, ms, flop / ms / 1e9))
    k = np.array([r[0] for r in res], float); t = np.array([r[1] for r in res], float)
    A = np.stack([k, np.ones_like(k)], 1)
    (slope, icpt), *_ = np.linalg.lstsq(A, t, rcond=None)
    print("fit: %.4f ms per chunk + %.4f ms fixed  (fixed = %.0f%% of the 4-chunk launch)" % (slope, icpt, 100 * icpt / (4 * slope + icpt)))


if __name__ == "__main__":
    main()
