"""Per-section cycle split of the Winograd kernel's workgroups (measurement build -DBSVD_WX_TL): where a wave's time goes.
usage: BSVD_HIP_LIB=build/ab/lib_ab0.so python tools/debug/wx_timeline.py [form=wino2] [Cin=256] [Cout=256] [H=135] [W=240] [frames=10]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bsvd_amd import _lib
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec
form = sys.argv[1] if len(sys.argv) > 1 else "wino2"
a = [int(v) for v in sys.argv[2:7]]
cin, cout, H, W, T = (a + [256, 256, 135, 240, 10][len(a):])
dev = torch.device("cuda", 0); rs = np.random.RandomState(0)
class Net: pass
pre = ConvSpec("pre", "pre", 4, cin, 1, False, "relu6", 0)
sp = ConvSpec("l", "l", cin, cout, 1, True, "relu6", 0)
net = Net(); net.layers = [pre, sp]
st = {}
for s in net.layers:
    st[s.key + ".weight"] = torch.from_numpy((rs.standard_normal((s.cout, s.cin, 3, 3)) * (1.5 / np.sqrt(9 * s.cin))).astype(np.float32))
    st[s.key + ".bias"] = torch.from_numpy((rs.standard_normal(s.cout) * 0.1).astype(np.float32))
ex = HipExecutor(PackedNet(net, st, dev, "f16x3", form))
x = ex.conv(pre, torch.rand((T, 4, H, W), device=dev) * 2 - 0.5, x_planar=True)
y = ex.conv(sp, x)
for _ in range(5): ex.conv(sp, x, out=y)
torch.cuda.synchronize()
lib = _lib.load()
n = 4096
buf = np.zeros((n, 12, 8), dtype=np.uint64)
lib.bsvd_debug_wx_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.bsvd_debug_wx_timeline(buf.ctypes.data, n)
assert rc == 0, rc
live = buf[:, 0, 6] > 0
b = buf[live].astype(np.float64)
nw = int((b[0, :, 6] > 0).sum())
print("%s %d->%d %dx%d x%d: %d workgroups sampled, %d waves each, %d chunks" % (form, cin, cout, H, W, T, live.sum(), nw, int(b[0, 0, 7])))
names = ["xform slot A", "MFMA steps", "xform slot B", "chunk barrier", "prologue", "epilogue", "total"]
for w in range(nw):
    m = b[:, w, :7].mean(axis=0)
    print("  wave %2d: " % w + "  ".join("%s %7.0f" % (names[k], m[k]) for k in range(7)) + "   per chunk: A %.0f M %.0f B %.0f bar %.0f" % tuple(m[k] / b[0, 0, 7] for k in range(4)))
