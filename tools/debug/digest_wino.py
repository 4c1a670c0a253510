import sys, os, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec
dev = torch.device("cuda", 0); rs = np.random.RandomState(0)
class Net: pass
for form in ("wino2", "wino6", "wino4"):
    for cin, cout, tsm, epi, H, W, T in ((128,128,True,0,37,50,3),(256,512,False,1,19,33,2)):
        pre = ConvSpec("pre","pre",4,cin,1,False,"relu6",0); sp = ConvSpec("l","l",cin,cout,1,tsm,"relu6" if epi==0 else "none",epi)
        net = Net(); net.layers=[pre,sp]; st={}
        for s in net.layers:
            st[s.key+".weight"]=torch.from_numpy((rs.standard_normal((s.cout,s.cin,3,3))*(1.5/np.sqrt(9*s.cin))).astype(np.float32)); st[s.key+".bias"]=torch.from_numpy((rs.standard_normal(s.cout)*0.1).astype(np.float32))
        ex = HipExecutor(PackedNet(net, st, dev, "f16x3", form))
        torch.manual_seed(1)
        x = ex.conv(pre, torch.rand((T,4,H,W),device=dev)*2-0.5, x_planar=True)
        kw = dict(extra=torch.zeros((T,2*H,2*W,cout//4),device=dev), extra_pstride=cout//4) if epi==1 else {}
        y = ex.conv(sp, x, **kw)
        print(form, cin, cout, hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16])
