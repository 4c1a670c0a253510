"""Shader clock and package power (rocm-smi) while ONE wide layer loops back to back: what does a kernel variant (BSVD_HIP_LIB=..., e.g. the
timing-only ablation builds) run at?   usage: python tools/debug/layer_power.py [form=wino2] [Cin=256] [H=135] [W=240] [frames=10] [seconds=6]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec

form = sys.argv[1] if len(sys.argv) > 1 else "wino2"
cin, H, W, T = ([int(v) for v in sys.argv[2:6]] + [256, 135, 240, 10][len(sys.argv[2:6]):])
secs = float(sys.argv[6]) if len(sys.argv) > 6 else 6.0
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)


class Net:
    pass


sp = ConvSpec("l", "l", cin, cin, 1, True, "relu6", 0)
net = Net(); net.layers = [sp]
st = {"l.weight": torch.from_numpy((rs.standard_normal((cin, cin, 3, 3)) * (1.5 / np.sqrt(9 * cin))).astype(np.float32)),
      "l.bias": torch.from_numpy((rs.standard_normal(cin) * 0.1).astype(np.float32))}
ex = HipExecutor(PackedNet(net, st, dev, "f16x3", form))
xf = form != "direct"
ex.force_x_f32 = ex.force_y_f32 = xf
x = torch.rand((T, H, W, cin), device=dev) * 3
if not xf:
    v = x.reshape(T, H, W, cin // 16, 16)                  # split16 container: per 16-channel chunk [hi x16 | lo x16] fp16 in the same 64 bytes
    hi = v.half()
    x = torch.cat([hi, (v - hi.float()).half()], dim=-1).contiguous().view(torch.float32).reshape(T, H, W, cin)
ex.record_variants = True
y = ex.conv(sp, x)
name = ex.last_variant
ex.record_variants = False
torch.cuda.synchronize()
rows, stop = [], False


def sample():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"GPU\[0\].*?sclk clock level:.*?\((\d+)Mhz\)", out)
        pw = re.search(r"GPU\[0\].*?Package Power \(W\):\s*([0-9.]+)", out)
        rows.append((int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))
        time.sleep(0.4)


th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        ex.conv(sp, x, out=y)
    torch.cuda.synchronize()
    n += 50
el = time.time() - t0
stop = True; th.join()
rows = rows[2:] or rows
sc = [r[0] for r in rows if r[0]]; pw = [r[1] for r in rows if r[1]]
print("%s %d->%d %dx%d x%d: %.4f ms per launch; sclk %.0f MHz (min %s max %s), package %.0f W over %d samples" %
      (name, cin, cin, H, W, T, el / n * 1e3, sum(sc) / max(1, len(sc)), min(sc) if sc else None, max(sc) if sc else None, sum(pw) / max(1, len(pw)), len(rows)))
