"""Cost of the overlapped sharded schedule on ONE GPU (no communication): every temporal-fusion layer runs as
interior + first + last launches instead of one.  usage: python tools/split_overhead.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bsvd_amd.schedule import Halo


class ZeroHalo:
    """start()/finish() protocol of dist.HaloExchanger with locally made zero slices (a middle rank's launch shape)."""
    def __init__(self, ex):
        self.ex, self.cache = ex, {}

    def start(self, sp, v):
        key = (sp.key, tuple(v.shape))
        if key not in self.cache:
            z = torch.zeros((v.shape[1], v.shape[2], sp.fold), dtype=v.dtype, device=v.device)
            self.cache[key] = (Halo(z, sp.fold, 0), Halo(z, sp.fold, 0))
        h = self.cache[key]

        class P:
            def finish(self):
                return h
        return P()

    def __call__(self, sp, v):
        return self.start(sp, v).finish()


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    for prec in ("f16x3", "fp32"):
        model = bench.build_model(dev, prec)
        lq, nm = bench.synth_clip(frames, 100, dev)
        x = torch.cat([lq, nm], dim=2)[0].contiguous()
        halo = ZeroHalo(model._executor(dev))
        with torch.no_grad():
            for name, fn in (("one launch per layer", None), ("interior+first+last", halo)):
                for _ in range(5):
                    model.clip_forward(x, fn)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    model.clip_forward(x, fn)
                torch.cuda.synchronize()
                print("%s %-22s %.2f ms/clip" % (prec, name, (time.perf_counter() - t0) / 20 * 1e3))


if __name__ == "__main__":
    main()
