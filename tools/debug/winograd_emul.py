"""CPU emulation of 1-D Winograd forms of the split-fp16 3x3 conv over the whole bsvd_c64 network (VERDICT r03 #1a).

The wide stride-1 layers (Cin >= 128: 16 temporal-fusion convs + 4 up-convs of a frame) are computed as
    out[y, m q + j] = sum_xi AT[j, xi] * sum_{ky, c} U[xi][co, c, ky] * V[xi][c, y + ky, q],      V = BT d along x,  U = G g along kx
with the kernel's arithmetic: activations arrive as fp16 pairs (hi + lo), the input transform runs in fp32 and RE-SPLITS every
transformed value into an fp16 pair, the weights are transformed in float64 at pack time and split, the three MFMA passes
(hi*hi + hi*lo + lo*hi) accumulate in fp32 (emulated by an fp32 conv) or float64 (isolates the operand rounding), the output
transform runs in fp32.  Everything else (64-channel layers, stride 2) stays the direct 3-pass split conv.

usage: python tools/debug/winograd_emul.py [H W frames] [scheme ...]
schemes: direct | f23 | f43 (points 0,+-1,+-2,inf) | f43h (points 0,+-1,+-1/2,inf) | f33 (0,+-1,2,inf) ; suffix ':acc32' = fp32 accumulate
"""
import os, sys
from fractions import Fraction
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from helpers import bsvd_keys
from seeded import seeded_state, seeded_clip
from oracle import bsvd_oracle as O

args = [a for a in sys.argv[1:]]
nums = [a for a in args if a.isdigit()]
schemes = [a for a in args if not a.isdigit()]
H = int(nums[0]) if len(nums) > 0 else 72
W = int(nums[1]) if len(nums) > 1 else 96
T = int(nums[2]) if len(nums) > 2 else 4
torch.set_num_threads(16)


def cook_toom(points, m, r=3):
    """AT (m x a), G (a x r), BT (a x a) of F(m, r) on the finite `points` + infinity, a = m + r - 1, exact rationals."""
    a = m + r - 1
    pts = [Fraction(p) for p in points]
    assert len(pts) == a - 1
    AT = [[(pts[i] ** j if i < a - 1 else (1 if j == m - 1 else 0)) for i in range(a)] for j in range(m)]
    G = []
    for i in range(a - 1):
        n = Fraction(1)
        for k in range(a - 1):
            if k != i:
                n *= pts[i] - pts[k]
        G.append([pts[i] ** k / n for k in range(r)])
    G.append([Fraction(0)] * (r - 1) + [Fraction(1)])
    # BT row i (finite point): coefficients of  prod_{k != i} (x - p_k)  ... solved instead from the defining identity
    # sum_i AT[j][i] G[i][a_] BT[i][b] = [b == j + a_]   (linear in BT), exact Gaussian elimination per column b
    BT = [[Fraction(0)] * a for _ in range(a)]
    for b in range(a):
        rows, rhs = [], []
        for j in range(m):
            for a_ in range(r):
                rows.append([AT[j][i] * G[i][a_] for i in range(a)])
                rhs.append(Fraction(1 if b == j + a_ else 0))
        # least squares is not needed: the system is consistent with a unique solution (rank a)
        Mx = [row[:] + [v] for row, v in zip(rows, rhs)]
        piv = []
        rr = 0
        for c in range(a):
            p = next((k for k in range(rr, len(Mx)) if Mx[k][c] != 0), None)
            if p is None:
                continue
            Mx[rr], Mx[p] = Mx[p], Mx[rr]
            pv = Mx[rr][c]
            Mx[rr] = [v / pv for v in Mx[rr]]
            for k in range(len(Mx)):
                if k != rr and Mx[k][c] != 0:
                    f = Mx[k][c]
                    Mx[k] = [u - f * v for u, v in zip(Mx[k], Mx[rr])]
            piv.append(c)
            rr += 1
        assert len(piv) == a and all(all(v == 0 for v in row) for row in Mx[rr:]), "inconsistent"
        for k, c in enumerate(piv):
            BT[c][b] = Mx[k][a]
    return AT, G, BT


FORMS = {"f23": ([0, 1, -1], 2), "f43": ([0, 1, -1, 2, -2], 4), "f43h": ([0, 1, -1, Fraction(1, 2), Fraction(-1, 2)], 4),
         "f33": ([0, 1, -1, 2], 3), "f33h": ([0, 1, -1, Fraction(1, 2)], 3),
         "f63": ([0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2)], 6),
         "f63q": ([0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(1, 4), Fraction(-1, 4)], 6),
         "f63t": ([0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 2), Fraction(-3, 2)], 6),
         "f63u": ([0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 4), Fraction(-3, 4)], 6)}


def balance(AT, G, BT):
    """scale row i of G by a power of two (exact in fp16 pairs) and BT's row i inversely so that max|BT row| is in [1, 2)"""
    a = len(G)
    for i in range(a):
        mx = max(abs(v) for v in BT[i])
        s = Fraction(1)
        while mx * s >= 2: s /= 2
        while mx * s < 1: s *= 2
        BT[i] = [v * s for v in BT[i]]
        G[i] = [v / s for v in G[i]]
    return AT, G, BT


def f16(x):
    return x.to(torch.float16).to(x.dtype)


def split(x):
    h = f16(x)
    return h, f16(x - h)


SCHEME = "direct"
ACC32 = False
MATS = {}


def mats(name):
    if name not in MATS:
        pts, m = FORMS[name]
        AT, G, BT = cook_toom(pts, m)
        MATS[name] = (m, np.array(AT, dtype=np.float64), np.array([[float(v) for v in r] for r in G]), np.array([[float(v) for v in r] for r in BT]))
        if "-v" in sys.argv:
            print(name, "BT=\n", MATS[name][3], "\nG=\n", MATS[name][2], "\nAT=\n", MATS[name][1])
    return MATS[name]


def conv3(xh, xl, wh, wl, stride, pad):
    """three split passes; fp32 or float64 accumulate"""
    dt = torch.float32 if ACC32 else torch.float64
    c = lambda a, ww: F.conv2d(a.to(dt), ww.to(dt), None, stride=stride, padding=pad)
    if ACC32:      # one fp32 accumulation chain over the three passes, like the MFMA accumulator: concatenate along K
        return F.conv2d(torch.cat([xh, xl, xh], 1).to(dt), torch.cat([wh, wh, wl], 1).to(dt), None, stride=stride, padding=pad).double()
    return c(xh, wh) + c(xl, wh) + c(xh, wl)


def conv_emul(x, w, b, stride):
    x = x.double(); w = w.double()
    xh, xl = split(x)
    name = SCHEME
    if name == "direct" or stride != 1 or x.shape[1] < 128:
        wh, wl = split(w)
        y = conv3(xh, xl, wh, wl, stride, 1)
        return (y + b.double().view(1, -1, 1, 1)).float()
    m, AT, G, BT = mats(name)
    a = m + 2
    N, C, Hh, Ww = x.shape
    Q = (Ww + m - 1) // m
    d = (xh + xl).float()                                  # what the staging decodes: exact in fp32
    d = F.pad(d, (1, 1 + Q * m - Ww, 1, 1))                # [N, C, H+2, Q*m + 2]
    # tiles along x: d_t[..., q, i] = d[..., m q + i], i < a
    idx = (torch.arange(Q)[:, None] * m + torch.arange(a)[None, :]).reshape(-1)
    dt_ = d[..., idx].reshape(N, C, Hh + 2, Q, a)
    BTt = torch.from_numpy(BT).float()
    V = torch.einsum("xi,nchqi->xnchq", BTt, dt_)          # fp32 transform (einsum order != the kernel's FMA chain: same error class)
    U = torch.einsum("xk,ocyk->xocy", torch.from_numpy(G), w)      # float64 at pack time  [a, Co, C, 3(ky)]
    Ms = []
    for xi in range(a):
        vh, vl = split(V[xi].double())
        uh, ul = split(U[xi])
        Ms.append(conv3(vh, vl, uh[..., None], ul[..., None], 1, 0).float())     # [N, Co, H, Q]   kernel (3,1) over rows
    Mst = torch.stack(Ms, 0)
    out = torch.einsum("jx,xnohq->nohqj", torch.from_numpy(AT).float(), Mst).reshape(N, -1, Hh, Q * m)[..., :Ww]
    return (out.double() + b.double().view(1, -1, 1, 1)).float()


def patched_conv(x, P, key, stride=1):
    return conv_emul(x, P[key + ".weight"], P[key + ".bias"], stride)


if __name__ == "__main__":
    st = seeded_state(bsvd_keys([64, 128, 256], 64, 4, 3, 64), 3)
    P = {k: torch.from_numpy(v) for k, v in st.items()}
    x = torch.from_numpy(seeded_clip((1, T, 4, H, W), 4, kind="sigma30"))
    ref = O.bsvd_clip(x, P)
    orig = O._conv
    # float64 reference of the same network (what both the oracle's fp32 and the emulations deviate from)
    O._conv = lambda xx, PP, key, stride=1: F.conv2d(xx.double(), PP[key + ".weight"].double(), PP[key + ".bias"].double(), stride=stride, padding=1)
    ref64 = O.bsvd_clip(x.double(), P)
    print("clip %dx%dx%d, |out|max %.3f; fp32 oracle vs float64: max-abs %.3e" % (T, H, W, float(ref.abs().max()), float((ref.double() - ref64).abs().max())))
    O._conv = patched_conv
    for s in schemes or ["direct", "direct:acc32", "f23", "f23:acc32", "f43", "f43:acc32", "f43h", "f43h:acc32", "f33:acc32"]:
        SCHEME, _, acc = s.partition(":")
        ACC32 = acc == "acc32"
        y = O.bsvd_clip(x, P)
        d = (y - ref).abs()
        d64 = (y.double() - ref64).abs()
        print("%-14s vs fp32 oracle: max-abs %.3e mean-abs %.3e   vs float64: max-abs %.3e mean-abs %.3e" % (s, float(d.max()), float(d.mean()), float(d64.max()), float(d64.mean())))
    O._conv = orig
