"""The wide stride-1 layers of bsvd_c64 at C1 geometry, one launch form after the other on the same realistic input (ReLU6-ranged split16
activations produced by a real layer, seeded weights): ms per launch over a sustained loop (the power cap applies) + max-abs difference of
each Winograd form to the direct kernel's output.   usage: python tools/debug/wino_layer_bench.py [seconds=2] [forms=direct,wino2,wino4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
forms = (sys.argv[2] if len(sys.argv) > 2 else "direct,wino2,wino2s,wino4,wino6").split(",")
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
LAYERS = [  # cin, cout, tsm, act, epi, H, W, T
    (128, 128, True, "relu6", 0, 270, 480, 10),
    (256, 256, True, "relu6", 0, 135, 240, 10),
    (128, 256, False, "none", 1, 270, 480, 10),
    (256, 512, False, "none", 1, 135, 240, 10),
    (128, 128, True, "relu6", 0, 270, 480, 1),
    (256, 256, True, "relu6", 0, 135, 240, 1),
    (128, 256, False, "none", 1, 270, 480, 1),
    (256, 512, False, "none", 1, 135, 240, 1),
    (128, 128, True, "relu6", 0, 540, 960, 1),      # the 1080p stream (C5): single-frame launches at twice the size
    (256, 256, True, "relu6", 0, 270, 480, 1),
    (128, 256, False, "none", 1, 540, 960, 1),
    (256, 512, False, "none", 1, 270, 480, 1),
    (64, 64, False, "relu6", 0, 540, 960, 10),      # the 64-channel layers (not Winograd-eligible by default: PackedNet(wino_min_cin=64) to try)
]
if os.environ.get("WINO_LAYERS"):
    LAYERS = [LAYERS[int(i)] for i in os.environ["WINO_LAYERS"].split(",")]


class Net:
    pass


for cin, cout, tsm, act, epi, H, W, T in LAYERS:
    pre = ConvSpec("pre", "pre", 4, cin, 1, False, "relu6", 0)
    sp = ConvSpec("l", "l", cin, cout, 1, tsm, act, epi)
    net = Net(); net.layers = [pre, sp]
    st = {}
    for s in net.layers:
        st[s.key + ".weight"] = torch.from_numpy((rs.standard_normal((s.cout, s.cin, 3, 3)) * (1.5 / np.sqrt(9 * s.cin))).astype(np.float32))
        st[s.key + ".bias"] = torch.from_numpy((rs.standard_normal(s.cout) * 0.1).astype(np.float32))
    ref = None
    x = None
    for form in forms:
        ex = HipExecutor(PackedNet(net, st, dev, "f16x3", form))
        if x is None:
            x = ex.conv(pre, torch.rand((T, 4, H, W), device=dev) * 2 - 0.5, x_planar=True)
            extra = None
            if epi == 1:
                extra = ex.conv(ConvSpec("pre", "pre", 4, cout // 4, 1, False, "relu6", 0) if cout // 4 == cin else pre,
                                torch.rand((T, 4, 2 * H, 2 * W), device=dev), x_planar=True) if cout // 4 == cin else torch.zeros((T, 2 * H, 2 * W, cout // 4), device=dev)
        kw = dict(extra=extra, extra_pstride=cout // 4) if epi == 1 else {}
        ex.record_variants = True
        y = ex.conv(sp, x, **kw)
        name = ex.last_variant
        ex.record_variants = False
        torch.cuda.synchronize()
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(20):
                ex.conv(sp, x, out=y, **kw)
            torch.cuda.synchronize()
            n += 20
        el = time.time() - t0
        yh = y.view(torch.float16).reshape(*y.shape[:-1], y.shape[-1] // 16, 32)
        yv = (yh[..., :16].float() + yh[..., 16:].float())
        if ref is None:
            ref = yv.clone()
        flop = 2.0 * cin * cout * 9 * H * W * T
        print("%-34s %d->%d epi %d %dx%d x%d: %.4f ms  %.0f TFLOP/s algorithmic   max-abs vs %s %.2e (|y| %.1f)"
              % (name, cin, cout, epi, H, W, T, el / n * 1e3, flop / (el / n) / 1e12, forms[0], float((yv - ref).abs().max()), float(ref.abs().max())), flush=True)
        del ex
