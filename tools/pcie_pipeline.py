#!/usr/bin/env python3
"""PCIe-inclusive throughput of the hot path: uint8 clips from pinned host memory through bsvd_amd.pipeline.ClipPipeline
(upload / forward / download on three streams) vs the HBM-resident rate bench.py reports as `value`.
usage: python tools/pcie_pipeline.py [clips=12] [frames=10] [H=540] [W=960]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402
from bsvd_amd.pipeline import ClipPipeline   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 540
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 960
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    clips = [rs.randint(0, 256, (F, H, W, 3)).astype(np.uint8) for _ in range(3)]
    for prec in ("f16x3", "fp32"):
        model = bench.build_model(dev, prec)
        x = torch.rand((F, 4, H, W), device=dev)
        with torch.no_grad():
            for _ in range(3):
                model.clip_forward(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                model.clip_forward(x)
            torch.cuda.synchronize()
            resident = n * F / (time.perf_counter() - t0)
            print("%s HBM-resident fp32 clip                         %.1f frames/s" % (prec, resident))
            for depth in (1, 2, 3):
                pipe = ClipPipeline(model, sigma=30 / 255.0, depth=depth)
                for _ in pipe.run(clips[i % 3] for i in range(3)):
                    pass
                t0 = time.perf_counter()
                k = 0
                for out in pipe.run(clips[i % 3] for i in range(n)):
                    k += out.shape[0]
                dt = time.perf_counter() - t0
                print("%s uint8 host -> device -> host, depth %d            %.1f frames/s (%.0f%% of resident)"
                      % (prec, depth, k / dt, 100.0 * k / dt / resident))


if __name__ == "__main__":
    main()
