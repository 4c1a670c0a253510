"""stress: is the head kernel deterministic while other kernels run concurrently?"""
import os, sys, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import bench
dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
side_mode = sys.argv[2] if len(sys.argv) > 2 else "conv"      # conv | none | copy | otherprec | fused (the fused entry kernel under test, conv alongside)
m = bench.build_model(dev, prec)
ex = m._executor(dev)
net = m.net
m2 = bench.build_model(dev, "fp32" if prec == "f16x3" else "f16x3")
ex2 = m2._executor(dev)
sp0, sp3 = net.temp1["inc0"], net.temp1["inc3"]
torch.manual_seed(0)
for (T, H, W) in ((1, 64, 96), (3, 64, 96), (1, 540, 960)):
    x = torch.rand((T, 4, H, W), device=dev)
    fused = side_mode.startswith("fused")
    run = (lambda: ex.conv_head_fused(sp0, sp3, x)) if fused else (lambda: ex.conv(sp0, x, x_planar=True))
    ref = run().clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.rand((2, 270, 480, 128), device=dev)
    spb = net.temp1["d0c1"]
    bad = 0
    for it in range(300):
        with torch.cuda.stream(side):
            if side_mode in ("conv", "fused"):
                ex.conv(spb, big)
            elif side_mode == "copy":
                big2 = big * 1.5
            elif side_mode == "otherprec":
                ex2.conv(spb, big)
        y = run()
        if it % 3 == 0:
            torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            if bad <= 3:
                d = (y != ref)
                idx = d.nonzero()
                print("  mismatch it", it, "count", int(d.sum()), "first", idx[0].tolist(), "last", idx[-1].tolist(),
                      "rows", sorted(set(idx[:, 1].tolist()))[:12], "chs", sorted(set(idx[:, 3].tolist()))[:12])
                f_, r_, c_, k_ = idx[0].tolist()
                yi, ri = y.view(torch.int32), ref.view(torch.int32)
                print("    got %08x ref %08x | ref row+1 %08x | ref col+1 %08x | ref ch+1 %08x ch-1 %08x | got==any other ref in this pixel: %s" % (
                    yi[f_, r_, c_, k_] & 0xffffffff, ri[f_, r_, c_, k_] & 0xffffffff, ri[f_, min(r_ + 1, ref.shape[1] - 1), c_, k_] & 0xffffffff,
                    ri[f_, r_, min(c_ + 1, ref.shape[2] - 1), k_] & 0xffffffff, ri[f_, r_, c_, min(k_ + 1, 63)] & 0xffffffff, ri[f_, r_, c_, max(k_ - 1, 0)] & 0xffffffff,
                    [int(j) for j in (ri[f_, r_, c_] == yi[f_, r_, c_, k_]).nonzero().flatten().tolist()]))
    torch.cuda.synchronize()
    print(prec, (T, H, W), "mismatching runs:", bad, "digest", hashlib.sha256(ref.cpu().numpy().tobytes()).hexdigest()[:12])
