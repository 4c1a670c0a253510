"""Consumer ceiling of the transformed-domain hand-over (BsvdConvArgs.x_v): the wide layers at C1 geometry, every
Winograd form with (a) its in-kernel input transform on plain-fp32 input (what ships) and (b) a transformed-domain input prepared by
bsvd_to_v -- the K loop of (b) is a 16-byte copy.  Sustained loops (ms per launch), output of (b) against (a) (same values, same weights:
the two differ only in where BT ran), and the stand-alone transform's own time for scale.
usage: python tools/debug/v_layer_bench.py [seconds=1.5] [forms=wino2,wino6]      (wino4: BSVD_HIP_LIB=build/measure/libbsvd_hip.so)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bsvd_amd import _lib
from bsvd_amd.engine import HipExecutor, PackedNet, _stream_ptr
from bsvd_amd.netspec import ConvSpec

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
forms = (sys.argv[2] if len(sys.argv) > 2 else "wino2,wino6").split(",")
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
LAYERS = [(128, 128, True, "relu6", 0, 270, 480, 10), (256, 256, True, "relu6", 0, 135, 240, 10), (128, 256, False, "none", 1, 270, 480, 10),
          (256, 512, False, "none", 1, 135, 240, 10), (256, 256, True, "relu6", 0, 135, 240, 1), (128, 128, True, "relu6", 0, 270, 480, 1),
          (128, 128, True, "relu6", 0, 240, 428, 10), (256, 256, True, "relu6", 0, 120, 214, 10)]
if os.environ.get("V_LAYERS"):
    LAYERS = [LAYERS[int(i)] for i in os.environ["V_LAYERS"].split(",")]
lib = _lib.load()


class Net:
    pass


def decode(t):
    h = t.view(torch.float16).reshape(*t.shape[:-1], t.shape[-1] // 16, 32)
    return (h[..., :16].float() + h[..., 16:].float()).reshape(t.shape)


def loop(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    return (time.time() - t0) / n * 1e3


for cin, cout, tsm, act, epi, H, W, T in LAYERS:
    pre = ConvSpec("pre", "pre", 4, cin, 1, False, "relu6", 0)
    sp = ConvSpec("l", "l", cin, cout, 1, tsm, act, epi)
    net = Net(); net.layers = [pre, sp]
    st = {}
    for s in net.layers:
        st[s.key + ".weight"] = torch.from_numpy((rs.standard_normal((s.cout, s.cin, 3, 3)) * (1.5 / np.sqrt(9 * s.cin))).astype(np.float32))
        st[s.key + ".bias"] = torch.from_numpy((rs.standard_normal(s.cout) * 0.1).astype(np.float32))
    xf = None
    for form in forms:
        ex = HipExecutor(PackedNet(net, st, dev, "f16x3", form))
        m = ex.packed.wino_m
        if xf is None:
            xf = decode(ex.conv(pre, torch.rand((T, 4, H, W), device=dev) * 2 - 0.5, x_planar=True)).contiguous()     # realistic values as plain fp32
        extra = torch.zeros((T, 2 * H, 2 * W, cout // 4), device=dev) if epi == 1 else None
        kw = dict(extra=extra, extra_pstride=cout // 4) if epi == 1 else {}
        ex.force_x_f32, ex.force_y_f32 = True, False
        a, y0 = ex.build_args(sp, xf, **kw)
        buf = ctypes.create_string_buffer(96)
        _lib.check(lib.bsvd_conv3x3_variant(ctypes.byref(a), buf, 96), "variant")
        name0 = buf.value.decode()
        ms0 = loop(lambda: _lib.check(lib.bsvd_conv3x3(ctypes.byref(a), _stream_ptr()), "conv f32 in"))
        # the same layer on a transformed-domain input
        vfe = lib.bsvd_v_frame_elems(H, W, cin, m)
        v = torch.empty((T, vfe), dtype=torch.float32, device=dev)
        tv = lambda: _lib.check(lib.bsvd_to_v(xf.data_ptr(), H * W * cin, 1, v.data_ptr(), vfe, T, H, W, cin, m, _stream_ptr()), "bsvd_to_v")
        ms_tv = loop(tv)
        b, y1 = ex.build_args(sp, xf, **kw)
        b.x, b.x_frame_stride, b.x_f32, b.x_v = v.data_ptr(), vfe, 0, m
        _lib.check(lib.bsvd_conv3x3_variant(ctypes.byref(b), buf, 96), "variant")
        name1 = buf.value.decode()
        ms1 = loop(lambda: _lib.check(lib.bsvd_conv3x3(ctypes.byref(b), _stream_ptr()), "conv V in"))
        ms2 = None
        if epi == 0 and m == 6:          # ... and with the epilogue writing the transformed domain (producer side of the hand-over, patch pass included)
            from bsvd_amd.engine import VT
            yv = VT.empty(T, H, W, cout, m, dev)
            c, _ = ex.build_args(sp, xf, **kw)
            c.x, c.x_frame_stride, c.x_f32, c.x_v = v.data_ptr(), vfe, 0, m
            c.y, c.y_frame_stride, c.y_f32, c.y_v = yv.data_ptr(), yv.frame_stride, 0, m
            ms2 = loop(lambda: _lib.check(lib.bsvd_conv3x3(ctypes.byref(c), _stream_ptr()), "conv V in V out"))
        d0, d1 = decode(y0), decode(y1)
        flop = 2.0 * cin * cout * 9 * H * W * T
        print("%d->%d epi %d %dx%d x%d  %-44s %.4f ms (%4.0f TF) | %-40s %.4f ms (%4.0f TF, %+.1f %%) | V in + V out %s ms | to_v %.3f ms | max-abs V-in vs f32-in %.2e, equal %s (|y| %.1f)"
              % (cin, cout, epi, H, W, T, name0, ms0, flop / ms0 / 1e9, name1, ms1, flop / ms1 / 1e9, (ms1 / ms0 - 1) * 100, ("%.4f" % ms2) if ms2 else "-", ms_tv,
                 float((d0 - d1).abs().max()), bool(torch.equal(y0, y1)), float(d0.abs().max())), flush=True)
        del ex, v
