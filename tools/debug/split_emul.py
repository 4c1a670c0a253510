"""CPU emulation of split-precision conv schemes over the whole bsvd_c64 network (seeded weights, sigma-30 clip):
which operand precisions do the two CORRECTION passes of the split-fp16 kernel need?  main term fp16 x fp16 always;
cross terms w_hi*x_lo + w_lo*x_hi in fp16 (= the shipped f16x3) or in block-scaled fp8 / fp6 / fp4 (MX, 32-channel blocks).
Accumulation in float64, so the numbers isolate the operand rounding.  usage: python tools/debug/split_emul.py [H W frames]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from helpers import bsvd_keys
from seeded import seeded_state, seeded_clip
from oracle import bsvd_oracle as O

H = int(sys.argv[1]) if len(sys.argv) > 1 else 72
W = int(sys.argv[2]) if len(sys.argv) > 2 else 96
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.set_num_threads(16)

FMT = {"e4m3": (3, 7, 448.0, 8), "e5m2": (2, 15, 57344.0, 15), "e2m3": (3, 1, 7.5, 2), "e3m2": (2, 3, 28.0, 4), "e2m1": (1, 1, 6.0, 2)}


def minifloat(x, fmt):
    mbits, bias, maxv, _ = FMT[fmt]
    ax = x.abs().clamp(max=maxv)
    e = torch.floor(torch.log2(ax.clamp(min=1e-300))).clamp(min=1 - bias)
    step = torch.pow(2.0, e - mbits)
    return torch.sign(x) * (torch.round(ax / step) * step).clamp(max=maxv)


def mx(x, fmt, dim=1, block=32):
    """MX block quantisation along `dim`: shared power-of-two scale per `block` elements, element format `fmt`."""
    emax = FMT[fmt][3]
    x = x.movedim(dim, -1)
    n = x.shape[-1]
    pad = (-n) % block
    xp = F.pad(x, (0, pad)).reshape(*x.shape[:-1], -1, block)
    amax = xp.abs().amax(dim=-1, keepdim=True)
    sc = torch.pow(2.0, torch.floor(torch.log2(amax.clamp(min=1e-300))) - emax)
    q = minifloat(xp / sc, fmt) * sc
    q = torch.where(amax > 0, q, torch.zeros_like(q))
    return q.reshape(*x.shape[:-1], -1)[..., :n].movedim(-1, dim)


def f16(x):
    return x.to(torch.float16).to(torch.float64)


def split(x):
    h = f16(x)
    return h, f16(x - h)


SCHEME = "exact3"


def conv_emul(x, w, b, stride):
    x = x.double(); w = w.double()
    xh, xl = split(x)
    wh, wl = split(w)
    c = lambda a, ww: F.conv2d(a, ww, None, stride=stride, padding=1)
    y = c(xh, wh)
    s = SCHEME
    if s == "fp16":
        pass
    elif s == "exact3" or x.shape[1] < 32:
        y = y + c(xl, wh) + c(xh, wl)
    elif s.startswith("mx:"):
        fmt = s[3:]
        y = y + c(mx(xl, fmt), mx(wh, fmt)) + c(mx(xh, fmt), mx(wl, fmt))
    elif s.startswith("mxlo:"):          # only the lo operands low precision is not an MFMA form; kept as the error floor of `fmt` lo parts
        fmt = s[5:]
        y = y + c(mx(xl, fmt), wh) + c(xh, mx(wl, fmt))
    elif s.startswith("mx1:"):           # ONE correction MFMA along a doubled K: [w_hi | w_lo] . [x_lo | x_hi] -- same arithmetic as mx:
        fmt = s[4:]
        y = y + c(mx(xl, fmt), mx(wh, fmt)) + c(mx(xh, fmt), mx(wl, fmt))
    else:
        raise ValueError(s)
    return (y + b.double().view(1, -1, 1, 1)).float()


def patched_conv(x, P, key, stride=1):
    assert O._norm_key(key) is None or O._norm_key(key) + ".running_mean" not in P
    return conv_emul(x, P[key + ".weight"], P[key + ".bias"], stride)


st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 3)
P = {k: torch.from_numpy(v) for k, v in st.items()}
x = torch.from_numpy(seeded_clip((1, T, 4, H, W), 4, kind="sigma30"))
ref = O.bsvd_clip(x, P)
orig = O._conv
O._conv = patched_conv
print("clip %dx%dx%d, |out|max %.3f" % (T, H, W, float(ref.abs().max())))
for s in sys.argv[4:] or ["fp16", "exact3", "mx:e4m3", "mx:e5m2", "mx:e3m2", "mx:e2m3", "mx:e2m1", "mxlo:e2m1"]:
    SCHEME = s
    y = O.bsvd_clip(x, P)
    d = (y - ref).abs()
    print("%-12s max-abs %.3e   mean-abs %.3e" % (s, float(d.max()), float(d.mean())))
O._conv = orig
