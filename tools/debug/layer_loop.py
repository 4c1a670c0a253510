"""One conv layer, realistic operands (seeded weights, ReLU6-ranged split16 input produced by a real layer), launched back to back for a few
seconds (sustained: the power cap applies): ms per launch.  For attributing costs with the timing-only ablation builds (BSVD_HIP_LIB=...,
-DBSVD_ABL=...): the layer's INPUT stays realistic whatever the ablated kernel writes.
usage: BSVD_HIP_LIB=... python tools/debug/layer_loop.py [Cin=128] [Cout=128] [H=270] [W=480] [frames=10] [seconds=3] [stride=1] [tsm=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec

a = [int(v) for v in sys.argv[1:6]]
cin, cout, H, W, T = (a + [128, 128, 270, 480, 10][len(a):])
secs = float(sys.argv[6]) if len(sys.argv) > 6 else 3.0
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
tsm = bool(int(sys.argv[8])) if len(sys.argv) > 8 else False
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)


class Net:
    pass


pre = ConvSpec("pre", "pre", 4, cin, 1, False, "relu6", 0)
sp = ConvSpec("l", "l", cin, cout, stride, tsm, "relu6", 0)
net = Net(); net.layers = [pre, sp]
st = {}
for s in net.layers:
    st[s.key + ".weight"] = torch.from_numpy((rs.standard_normal((s.cout, s.cin, 3, 3)) * (1.5 / np.sqrt(9 * s.cin))).astype(np.float32))
    st[s.key + ".bias"] = torch.from_numpy((rs.standard_normal(s.cout) * 0.1).astype(np.float32))
ex = HipExecutor(PackedNet(net, st, dev, precision="f16x3"))
# the input of the measured layer comes from the UNABLATED library when BSVD_INPUT_LIB is given (a file written by a previous run)
inp = os.environ.get("BSVD_LAYER_INPUT")
if inp and os.path.exists(inp):
    x = torch.load(inp).to(dev)
else:
    x = ex.conv(pre, torch.rand((T, 4, H, W), device=dev) * 2 - 0.5, x_planar=True)
    if inp:
        torch.save(x.cpu(), inp)
ex.record_variants = True
y = ex.conv(sp, x)
name = ex.last_variant
ex.record_variants = False
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        ex.conv(sp, x, out=y)
    torch.cuda.synchronize()
    n += 50
el = time.time() - t0
print("%s  %d->%d stride %d %dx%d x%d: %.4f ms per launch (%d launches)" % (name, cin, cout, stride, H, W, T, el / n * 1e3, n))
