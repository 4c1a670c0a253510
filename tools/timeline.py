#!/usr/bin/env python3
"""Per-workgroup timeline of one conv launch from a -DBSVD_TIMELINE build of the library (BSVD_HIP_LIB=...):
where a tile's time goes (prologue / K loop / epilogue) and how long a CU slot stays empty between two workgroups.
usage: BSVD_HIP_LIB=build/ab/lib_tl.so python tools/timeline.py [Cin=128] [Cout=128] [H=270] [W=480] [frames=10] [stride=1]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bsvd_amd import _lib
from bsvd_amd.engine import HipExecutor, PackedNet
from bsvd_amd.netspec import ConvSpec


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 270
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 480
    T = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)

    class Net:
        pass
    pre = ConvSpec("pre", "pre", 4, cin, 1, False, "relu6", 0)
    sp = ConvSpec("l", "l", cin, cout, stride, False, "relu6", 0)
    post = ConvSpec("post", "post", cout, 16, 1, False, "none", 0)
    net = Net(); net.layers = [pre, sp, post]
    st = {}
    for s in net.layers:
        st[s.key + ".weight"] = torch.from_numpy((rs.standard_normal((s.cout, s.cin, 3, 3)) * (1.5 / np.sqrt(9 * s.cin))).astype(np.float32))
        st[s.key + ".bias"] = torch.zeros(s.cout)
    ex = HipExecutor(PackedNet(net, st, dev, precision="f16x3"))
    a = ex.conv(pre, torch.rand((T, 4, H, W), device=dev), x_planar=True)
    ex.record_variants = True
    y = ex.conv(sp, a)
    print("kernel:", ex.last_variant)
    ex.record_variants = False
    for _ in range(3):
        ex.conv(sp, a, out=y)
    torch.cuda.synchronize()
    ex.conv(sp, a, out=y)              # the launch whose stamps we read
    torch.cuda.synchronize()
    lib = _lib.load()
    n = 1 << 16
    buf = np.zeros((n, 8), dtype=np.uint64)
    lib.bsvd_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = lib.bsvd_debug_timeline(buf.ctypes.data, n)
    assert rc == 0, rc
    live = buf[:, 0] > 0
    t = buf[live].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t[:, :4] - t0) / 100.0                       # s_memrealtime ticks at 100 MHz
    pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    print("%d workgroups, launch span %.1f us" % (len(us), us[:, 3].max()))
    if os.environ.get("TL_PHASES"):                    # -DBSVD_TIMELINE=2 build: slots 5..7 = shader cycles per epilogue phase
        ph = t[:, 5:8].astype(np.float64)
        tot = ph.sum(axis=1)
        print("  epilogue phases of wave 0 (shader cycles, mean): staging writes+wait %.0f, scratch reads+wait %.0f, convert+store issue %.0f"
              "  (= %.2f / %.2f / %.2f of their sum)" % (ph[:, 0].mean(), ph[:, 1].mean(), ph[:, 2].mean(),
                                                         *(ph / tot[:, None]).mean(axis=0)))
    else:
        e1 = (t[:, 5] - t0) / 100.0 - us[:, 2]; e2 = (t[:, 6] - t0) / 100.0 - us[:, 2]
        print("  inside the epilogue: first item done after %.2f us, half of the items after %.2f us, all after %.2f us (means)"
              % (e1.mean(), e2.mean(), (us[:, 3] - us[:, 2]).mean()))
    for name, v in (("prologue", pro), ("K loop", loop), ("epilogue", epi), ("whole tile", us[:, 3] - us[:, 0])):
        print("  %-10s mean %7.2f us   p10 %7.2f   p50 %7.2f   p90 %7.2f" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
    hw = t[:, 4]
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("  distinct CUs seen: %d" % len(np.unique(cukey)))
    # per CU: busy intervals -> idle gaps between the end of one workgroup and the start of the next one on that CU
    gaps, conc = [], []
    for k in np.unique(cukey):
        m = cukey == k
        s_, e_ = us[m, 0], us[m, 3]
        order = np.argsort(s_)
        s_, e_ = s_[order], e_[order]
        conc.append((e_ - s_).sum() / (e_.max() - s_.min()))
        # gap a new workgroup waits after some workgroup on this CU ended (nearest earlier end)
        for i in range(len(s_)):
            prev_end = e_[e_ <= s_[i] + 1e-9]
            if len(prev_end):
                gaps.append(s_[i] - prev_end.max())
    gaps = np.array(gaps)
    print("  mean resident workgroups per CU %.2f" % np.mean(conc))
    print("  slot refill gap (end of a workgroup -> start of the next on that CU): mean %.2f us  p50 %.2f  p90 %.2f" % (gaps.mean(), np.percentile(gaps, 50), np.percentile(gaps, 90)))


if __name__ == "__main__":
    main()
