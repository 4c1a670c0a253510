"""The plain-fp32 hand-over (BsvdConvArgs.x_f32 / y_f32) layer by layer: the wide layers at C1 geometry in the four combinations of input /
output format on the same realistic values, sustained loops (ms per launch).   usage: python tools/debug/wino_f32_bench.py [seconds=2] [form=wino2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
form = sys.argv[2] if len(sys.argv) > 2 else "wino2"
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
LAYERS = [(128, 128, True, "relu6", 0, 270, 480, 10), (256, 256, True, "relu6", 0, 135, 240, 10), (128, 256, False, "none", 1, 270, 480, 10),
          (256, 512, False, "none", 1, 135, 240, 10), (256, 256, True, "relu6", 0, 135, 240, 1)]


class Net:
    pass


def decode(t):
    h = t.view(torch.float16).reshape(*t.shape[:-1], t.shape[-1] // 16, 32)
    return (h[..., :16].float() + h[..., 16:].float()).reshape(t.shape)


for cin, cout, tsm, act, epi, H, W, T in LAYERS:
    pre = ConvSpec("pre", "pre", 4, cin, 1, False, "relu6", 0)
    sp = ConvSpec("l", "l", cin, cout, 1, tsm, act, epi)
    net = Net(); net.layers = [pre, sp]
    st = {}
    for s in net.layers:
        st[s.key + ".weight"] = torch.from_numpy((rs.standard_normal((s.cout, s.cin, 3, 3)) * (1.5 / np.sqrt(9 * s.cin))).astype(np.float32))
        st[s.key + ".bias"] = torch.from_numpy((rs.standard_normal(s.cout) * 0.1).astype(np.float32))
    ex = HipExecutor(PackedNet(net, st, dev, "f16x3", form))
    xp = ex.conv(pre, torch.rand((T, 4, H, W), device=dev) * 2 - 0.5, x_planar=True)       # fp16 pairs
    xf = decode(xp).contiguous()                                                            # the same values as plain fp32
    extra = torch.zeros((T, 2 * H, 2 * W, cout // 4), device=dev) if epi == 1 else None
    kw = dict(extra=extra, extra_pstride=cout // 4) if epi == 1 else {}
    ref = None
    for x_f32, y_f32 in ((False, False), (True, False), (False, True), (True, True)):
        ex.force_x_f32, ex.force_y_f32 = x_f32, y_f32
        x = xf if x_f32 else xp
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
        yv = y if y_f32 else decode(y)
        if ref is None:
            ref = yv.clone()
        print("%-40s %d->%d epi %d %dx%d x%d  in %-5s out %-5s: %.4f ms   max-abs vs pairs/pairs %.2e" % (name, cin, cout, epi, H, W, T, "f32" if x_f32 else "pairs", "f32" if y_f32 else "pairs", el / n * 1e3, float((yv - ref).abs().max())), flush=True)
