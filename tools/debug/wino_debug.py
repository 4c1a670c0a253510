"""structured single-layer probes of the Winograd kernel (delta weights / delta inputs) -- prints where the output differs from the direct kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec
dev = torch.device("cuda", 0)
form = sys.argv[1] if len(sys.argv) > 1 else "wino4"
H, W, T, cin, cout = 16, 16, 1, 128, 128

def to_split(x):
    *lead, C = x.shape
    v = x.reshape(*lead, C // 16, 16); hi = v.half(); lo = (v - hi.float()).half()
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).reshape(*lead, C)
def from_split(s):
    *lead, C = s.shape
    h = s.contiguous().view(torch.float16).reshape(*lead, C // 16, 32)
    return (h[..., :16].float() + h[..., 16:].float()).reshape(*lead, C)

class Net: pass
sp = ConvSpec("l", "l", cin, cout, 1, False, "none", 0)
pre = ConvSpec("pre", "pre", 4, 16, 1, False, "none", 0)
post = ConvSpec("post", "post", 16, 3, 1, False, "none", 2)
net = Net(); net.layers = [pre, sp, post]
def run(w, x, b=None):
    st = {"pre.weight": torch.zeros(16, 4, 3, 3), "pre.bias": torch.zeros(16), "post.weight": torch.zeros(3, 16, 3, 3), "post.bias": torch.zeros(3),
          "l.weight": w, "l.bias": torch.zeros(cout) if b is None else b}
    outs = []
    for f in ("direct", form):
        ex = HipExecutor(PackedNet(net, st, dev, "f16x3", f))
        outs.append(from_split(ex.conv(sp, to_split(x).to(dev)).cpu()))
    return outs
# probe 1: centre-tap identity weights (cout n <- cin n), random input: y == x
w = torch.zeros(cout, cin, 3, 3); w[torch.arange(cout), torch.arange(cin), 1, 1] = 1.0
x = torch.randn(T, H, W, cin)
d, g = run(w, x)
print("probe identity: direct err %.2e, %s err %.2e" % (float((d - x).abs().max()), form, float((g - x).abs().max())))
e = (g - x).abs()[0]
print(" err by row:", [round(float(v), 2) for v in e.amax(dim=(1, 2))])
print(" err by col:", [round(float(v), 2) for v in e.amax(dim=(0, 2))])
print(" err by ch :", [round(float(v), 2) for v in e.amax(dim=(0, 1))][:64])
# which input does output (y=5,x=6,ch=3) equal?
if float(e.max()) > 1e-3:
    tgt = g[0, 5, 6, 3]
    cand = (x[0] - tgt).abs()
    idx = torch.nonzero(cand < 1e-3)
    print(" out[5,6,3]=%.4f matches x at" % float(tgt), idx[:5].tolist(), " x[5,6,3]=%.4f" % float(x[0, 5, 6, 3]))
# probe 2: per-tap deltas with a single hot input pixel/channel
for (ky, kx) in ((1, 1), (0, 0), (2, 2), (1, 0), (1, 2), (0, 1)):
    w = torch.zeros(cout, cin, 3, 3); w[:, :, ky, kx] = torch.eye(cout)
    x = torch.zeros(T, H, W, cin); x[0, 7, 9, 5] = 1.0; x[0, 3, 2, 77] = 2.0
    d, g = run(w, x)
    nz = torch.nonzero(g[0].abs() > 1e-3)
    print("probe tap (%d,%d): direct nonzeros %s ; %s nonzeros %s vals %s" % (ky, kx, torch.nonzero(d[0].abs() > 1e-3).tolist(), form, nz[:8].tolist(), [round(float(g[0][tuple(i)]), 3) for i in nz[:8]]))
