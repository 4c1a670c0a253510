import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec
dev = torch.device("cuda", 0)
form = sys.argv[1] if len(sys.argv) > 1 else "wino4"
def to_split(x):
    *lead, C = x.shape
    v = x.reshape(*lead, C // 16, 16); hi = v.half(); lo = (v - hi.float()).half()
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).reshape(*lead, C)
def from_split(s):
    *lead, C = s.shape
    h = s.contiguous().view(torch.float16).reshape(*lead, C // 16, 32)
    return (h[..., :16].float() + h[..., 16:].float()).reshape(*lead, C)
class Net: pass
def run(cin, cout, w, x):
    sp = ConvSpec("l", "l", cin, cout, 1, False, "none", 0)
    net = Net(); net.layers = [ConvSpec("pre", "pre", 4, 16, 1, False, "none", 0), sp, ConvSpec("post", "post", 16, 3, 1, False, "none", 2)]
    st = {"pre.weight": torch.zeros(16, 4, 3, 3), "pre.bias": torch.zeros(16), "post.weight": torch.zeros(3, 16, 3, 3), "post.bias": torch.zeros(3), "l.weight": w, "l.bias": torch.zeros(cout)}
    outs = []
    for f in ("direct", form):
        ex = HipExecutor(PackedNet(net, st, dev, "f16x3", f))
        outs.append(from_split(ex.conv(sp, to_split(x).to(dev)).cpu()))
    return outs
torch.manual_seed(0)
for cin, cout, H, W, nzch in ((128, 128, 16, 16, None), (256, 128, 16, 16, None), (128, 128, 16, 16, range(0, 16)), (128, 128, 16, 16, range(0, 32)), (128, 128, 16, 16, range(16, 48)), (128,128,16,16,range(0,128,16))):
    w = torch.randn(cout, cin, 3, 3) * 0.05
    if nzch is not None:
        m = torch.zeros(cin); m[list(nzch)] = 1; w = w * m.view(1, -1, 1, 1)
    x = torch.randn(1, H, W, cin)
    d, g = run(cin, cout, w, x)
    e = (g - d).abs()[0]
    print("cin %d cout %d nz %s: max err %.3e |y| %.2f" % (cin, cout, None if nzch is None else (nzch[0], nzch[-1]), float(e.max()), float(d.abs().max())))
    if float(e.max()) > 1e-3:
        print("  by row:", [round(float(v), 2) for v in e.amax(dim=(1, 2))])
        print("  by col:", [round(float(v), 2) for v in e.amax(dim=(0, 2))])
        print("  by ch/8:", [round(float(v), 2) for v in e.amax(dim=(0, 1)).reshape(-1, 8).amax(dim=1)])
