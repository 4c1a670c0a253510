"""sha256 of the f16x3 / fp32 outputs of one seeded bsvd_c64 clip (compare two library builds: BSVD_HIP_LIB=... )"""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import bsvd_amd
from helpers import bsvd_keys
from seeded import seeded_state, seeded_clip
st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 3)
shape = (1, 10, 4, 540, 960) if "c1" in sys.argv[1:] else (1, 4, 4, 136, 200)      # c1: the headline geometry (fat tiles, dead waves, zero-chunk skip)
x = torch.from_numpy(seeded_clip(shape, 4, kind="sigma30")).cuda()
for prec in ("f16x3", "fp32"):
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None, precision=prec)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    y = m.cuda()(x)
    torch.cuda.synchronize()
    print(prec, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16], float(y.abs().max()))
