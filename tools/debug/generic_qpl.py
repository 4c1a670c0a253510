"""debug: where does the generic path differ from the oracle?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from test_gpu_parity import _one_layer_net, _gpu_exec, _dev
from oracle_exec import OracleExecutor
from seeded import seeded_state
from bsvd_amd.netspec import pad16
from bsvd_amd.schedule import Halo
cin, cout, stride, tsm, act, epi, T, H, W = 24, 40, 1, True, "relu6", 0, 3, 8, 12
rs = np.random.RandomState(1)
st = seeded_state([("l.weight", (cout, cin, 3, 3)), ("l.bias", (cout,))], 7)
net, sp = _one_layer_net(cin, cout, stride, tsm, act, epi)
gex, oex = _gpu_exec(net, st), OracleExecutor(st, double=True)
x = torch.zeros((T, H, W, pad16(cin)))
x[..., :cin] = torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32))
want = oex.conv(sp, x, None, None, None, 0, 1)
got = gex.conv(sp, x.to(_dev()), None, None, None, 0, 1).cpu()
err = (got - want).abs()
print("max", float(err.max()))
bad = (err > 1e-4)
print("bad count", int(bad.sum()), "of", bad.numel())
print("bad per frame", bad.sum(dim=(1, 2, 3)).tolist())
print("bad per row", bad.sum(dim=(0, 2, 3)).tolist())
print("bad per col", bad.sum(dim=(0, 1, 3)).tolist())
print("bad per channel", bad.sum(dim=(0, 1, 2)).tolist())
