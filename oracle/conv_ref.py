"""ctypes binding + clip-level composition of the plain-C oracle (oracle/conv_ref.c).

TEST INFRASTRUCTURE ONLY.  ``bsvd_clip_c`` evaluates the whole network with the double-accumulating
C conv, frame by frame, following SURVEY.md Appendix C (same function as
/root/reference/Experimental_root/archs/bsvd_arch.py:490-552).  It is slow (tiny shapes only) and
exists to arbitrate between oneDNN's fp32 conv and the HIP kernels.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_FP = ctypes.POINTER(ctypes.c_float)
ACT = {"none": 0, "relu": 1, "relu6": 2}


def build(force=False):
    so = os.path.join(_HERE, "libconv_ref.so")
    src = os.path.join(_HERE, "conv_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", src, "-o", so])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_conv3x3.restype = ctypes.c_int
        _LIB.oracle_conv3x3.argtypes = [_FP, _FP, _FP, ctypes.c_int, _FP, _FP] + [ctypes.c_int] * 7 + [_FP, _FP]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(_FP)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def conv3x3(cur, w, bias, prev_sl=None, next_sl=None, fold=0, stride=1, act="none", epilogue=0, extra=None):
    """One frame, NCHW.  cur [Cin,H,W] -> [Cout,Ho,Wo] (epilogue 0/2) or [Cout/4,2Ho,2Wo] (epilogue 1)."""
    cur, w, bias, prev_sl, next_sl, extra = map(_c, (cur, w, bias, prev_sl, next_sl, extra))
    cin, h, wd = cur.shape
    cout = w.shape[0]
    ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
    out = np.empty((cout // 4, 2 * ho, 2 * wo) if epilogue == 1 else (cout, ho, wo), dtype=np.float32)
    rc = lib().oracle_conv3x3(_p(cur), _p(prev_sl), _p(next_sl), fold, _p(w), _p(bias), cin, cout, h, wd,
                              stride, ACT[act], epilogue, _p(extra), _p(out))
    if rc != 0:
        raise ValueError("oracle_conv3x3 rejected its arguments")
    return out


def _denblock_clip_c(x, P, pre, act):
    T = x.shape[0]

    def plain(v, key, stride=1, a=act, epilogue=0, extra=None):
        return np.stack([conv3x3(v[t], P[pre + key + ".weight"], P[pre + key + ".bias"], stride=stride, act=a,
                                 epilogue=epilogue, extra=None if extra is None else extra[t])
                         for t in range(T)])

    def tsm(v, key):
        fold = v.shape[1] // 8
        outs = []
        for t in range(T):
            prev_sl = v[t - 1, fold:2 * fold] if t > 0 else None
            next_sl = v[t + 1, :fold] if t + 1 < T else None
            outs.append(conv3x3(v[t], P[pre + key + ".weight"], P[pre + key + ".bias"], prev_sl, next_sl, fold,
                                act=act))
        return np.stack(outs)

    x0 = plain(plain(x, "inc.convblock.0"), "inc.convblock.3")
    x1 = tsm(tsm(plain(x0, "downc0.convblock.0", 2), "downc0.memconv.c1.op.conv"), "downc0.memconv.c2.op.conv")
    x2 = tsm(tsm(plain(x1, "downc1.convblock.0", 2), "downc1.memconv.c1.op.conv"), "downc1.memconv.c2.op.conv")
    u = tsm(tsm(x2, "upc2.memconv.c1.op.conv"), "upc2.memconv.c2.op.conv")
    v = plain(u, "upc2.convblock.0", a="none", epilogue=1, extra=x1)
    v = tsm(tsm(v, "upc1.memconv.c1.op.conv"), "upc1.memconv.c2.op.conv")
    w = plain(v, "upc1.convblock.0", a="none", epilogue=1, extra=x0)
    o = plain(w, "outc.convblock.0")
    return plain(o, "outc.convblock.3", a="none", epilogue=2, extra=x)


def bsvd_clip_c(x, P, act="relu6"):
    """x [N,F,C,H,W] numpy, P {key: numpy}; returns [N,F,out_ch,H,W]."""
    n, f, c, h, w = x.shape
    v = np.ascontiguousarray(x.reshape(n * f, c, h, w), dtype=np.float32)
    y = _denblock_clip_c(_denblock_clip_c(v, P, "temp1.", act), P, "temp2.", act)
    return y.reshape(n, f, y.shape[1], h, w)
