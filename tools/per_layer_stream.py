#!/usr/bin/env python3
"""Per-layer cost of the stream schedule's frames=1 launches vs the clip schedule's frames=T launches (ms per frame, each
layer launched back to back on one stream): where the single-frame launches lose against the clip (tail rounds, tile choice).
usage: python tools/per_layer_stream.py [precision=f16x3] [HxW=540x960] [T=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bsvd_amd.schedule import Halo


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    H, W = map(int, (sys.argv[2] if len(sys.argv) > 2 else "540x960").split("x"))
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev, prec)
    ex = model._executor(dev)
    ex.record_variants = True
    rows = []
    tot = {1: 0.0, T: 0.0}
    for blk in (model.net.temp1, model.net.temp2):
        h, w = H, W
        for name, sp in blk.items():
            res = {}
            for frames in (1, T):
                planar = sp.key == "temp1.inc.convblock.0"
                x = torch.rand((frames, sp.cin, h, w), device=dev) if planar else torch.rand((frames, h, w, sp.cin_pad), device=dev)
                kw = {}
                if planar:
                    kw["x_planar"] = True
                ho, wo = (h - 1) // sp.stride + 1, (w - 1) // sp.stride + 1
                if sp.epilogue == 1:
                    kw["extra"] = torch.rand((frames, 2 * ho, 2 * wo, sp.cout_pad // 4), device=dev)
                    kw["extra_pstride"] = sp.cout_pad // 4
                if sp.epilogue == 2:
                    if sp.key.startswith("temp2"):
                        kw["extra"] = torch.rand((frames, h, w, 64), device=dev); kw["extra_pstride"] = 64
                        kw["y_planar"] = (3, None)
                    else:
                        kw["extra"] = torch.rand((frames, 4, h, w), device=dev); kw["extra_pstride"] = 1; kw["extra_cstride"] = h * w
                if sp.tsm and frames == 1:
                    nb = torch.rand((2, h, w, sp.cin_pad), device=dev)
                    kw["halo_prev"] = Halo(nb[0], sp.cin_pad, sp.fold); kw["halo_next"] = Halo(nb[1], sp.cin_pad, 0)
                out = ex.conv(sp, x, **kw)
                for _ in range(3):
                    ex.conv(sp, x, out=out, **kw)
                reps = 20 if frames == 1 else 5
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ex.conv(sp, x, out=out, **kw)
                e1.record(); torch.cuda.synchronize()
                res[frames] = (e0.elapsed_time(e1) / reps / frames, ex.last_variant)
                tot[frames] += res[frames][0]
            flop = 2.0 * sp.macs(h, w)
            rows.append((sp.key, res[1][0], res[T][0], res[1][1], res[T][1], flop))
            if sp.stride == 2:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            if sp.epilogue == 1:
                h, w = 2 * h, 2 * w
    print("%-34s %9s %9s %6s  %s" % ("layer", "T=1 ms/f", "T=%d ms/f" % T, "ratio", "kernel (T=1 | T=%d)" % T))
    for k, a, b, na, nb, flop in rows:
        print("%-34s %9.4f %9.4f %6.2f  %s%s   %.0f / %.0f TFLOP/s" % (k, a, b, a / b, na.replace("conv3x3_kernel", ""),
              "" if na == nb else " | " + nb.replace("conv3x3_kernel", ""), flop / a / 1e9, flop / b / 1e9))
    print("sum: T=1 %.3f ms/frame, T=%d %.3f ms/frame (ratio %.3f)" % (tot[1], T, tot[T], tot[1] / tot[T]))


if __name__ == "__main__":
    main()
