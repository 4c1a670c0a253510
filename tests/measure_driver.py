"""Runs in a fresh interpreter with BSVD_HIP_LIB = a MEASUREMENT build of the library (tools/build_measure.sh; -DBSVD_MEASURE): the kernel
variants DESIGN.md 4.1d records as slower -- F(4,3), the forced 8-row tiles, the 4-wave workgroup, one-tile-per-workgroup and persistent
F(2,3), the all-positions-per-wave kernel of conv3x3_wino.hip -- against the CPU oracle on the product suite's layer cases and a few random
ones, and the bit-identity claims between variants of one form.  Test infrastructure; the product never loads this library.

    python tests/measure_driver.py forms            every measurement form vs the oracle (+ bit-identical variants of F(2,3) / F(6,3))
    python tests/measure_driver.py digest <out>     output digests of the product forms on two fixed layers (run under two builds and
                                                    compare: is a compile-time knob such as BSVD_WX_MIXASM bit-identical?)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def forms():
    from bsvd_amd import _lib, engine
    import test_gpu_wino as WN
    lib = _lib.load()
    assert lib.bsvd_build_info() & _lib.BUILD_MEASURE, "measure_driver needs BSVD_HIP_LIB = a -DBSVD_MEASURE build"
    n = 0
    for form in sorted(engine.MEASURE_WIDE_CONV):
        for case in WN.CASES:
            WN.test_wino_layer_vs_oracle(form, *case)
            n += 1
    # random eligible layers, the generator of test_gpu_fuzz.test_random_wide_layer_winograd_forms
    names = sorted(engine.MEASURE_WIDE_CONV)
    for seed in range(24):
        rs = np.random.RandomState(7000 + seed)
        form = names[seed % len(names)]
        epi = int(rs.choice([0, 0, 1]))
        tsm = bool(epi == 0 and rs.rand() < 0.6)
        cin = int(rs.choice([128, 256]))
        cout = cin if tsm else int(rs.choice([128, 256, 512] if epi == 1 else [64, 128, 256]))
        act = str(rs.choice(["relu6", "relu", "none"])) if epi == 0 else "none"
        T, H, W = int(rs.randint(1, 4)), int(rs.randint(1, 21)), int(rs.randint(1, 49))
        WN.test_wino_layer_vs_oracle(form, cin, cout, tsm, act, epi, T, H, W)
        n += 1
    # variants of ONE form compute the same bits: tile height, one tile per workgroup / persistent walk
    from bsvd_amd.netspec import ConvSpec
    from seeded import seeded_state
    from test_gpu_f16x3 import _Net, to_split
    rs = np.random.RandomState(11)
    for cin, cout, epi, T, H, W in ((128, 128, 0, 2, 37, 50), (256, 256, 0, 3, 20, 33), (128, 256, 1, 2, 18, 40)):
        sp = ConvSpec("l", "l", cin, cout, 1, epi == 0, "relu6" if epi == 0 else "none", epi)
        st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                           ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
        x = to_split(torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32))).cuda()
        extra, eps = None, 0
        if epi == 1:
            extra, eps = to_split(torch.from_numpy(rs.standard_normal((T, 2 * H, 2 * W, cout // 4)).astype(np.float32))).cuda(), cout // 4
        outs = {}
        for form in ("wino2", "wino2h", "wino2n", "wino2p", "wino6", "wino6h"):
            outs[form] = WN._exec(_Net(sp), st, form).conv(sp, x, None, None, extra, eps, 1).clone()
        for a, b in (("wino2", "wino2h"), ("wino2", "wino2n"), ("wino2", "wino2p"), ("wino6", "wino6h")):
            assert torch.equal(outs[a], outs[b]), (a, b, cin, cout, epi)
        n += 1
    print("MEASURE FORMS OK", n)


def digest(out):
    import test_gpu_wino as WN
    from bsvd_amd.netspec import ConvSpec
    from seeded import seeded_state
    from test_gpu_f16x3 import _Net, to_split
    rs = np.random.RandomState(3)
    res = {}
    for cin, cout, epi, T, H, W in ((128, 128, 0, 2, 40, 64), (256, 512, 1, 1, 19, 31)):
        sp = ConvSpec("l", "l", cin, cout, 1, epi == 0, "relu6" if epi == 0 else "none", epi)
        st = seeded_state([("e0.weight", (16, 4, 3, 3)), ("e0.bias", (16,)), ("l.weight", (cout, cin, 3, 3)),
                           ("l.bias", (cout,)), ("e1.weight", (3, 16, 3, 3)), ("e1.bias", (3,))], 7)
        x = to_split(torch.from_numpy(rs.standard_normal((T, H, W, cin)).astype(np.float32))).cuda()
        extra, eps = None, 0
        if epi == 1:
            extra, eps = to_split(torch.from_numpy(rs.standard_normal((T, 2 * H, 2 * W, cout // 4)).astype(np.float32))).cuda(), cout // 4
        for form in ("wino2", "wino6"):
            y = WN._exec(_Net(sp), st, form).conv(sp, x, None, None, extra, eps, 1)
            res["%s %d->%d epi%d" % (form, cin, cout, epi)] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()
    json.dump(res, open(out, "w"), indent=1)
    print("DIGEST OK", out)


if __name__ == "__main__":
    if sys.argv[1] == "forms":
        forms()
    elif sys.argv[1] == "digest":
        digest(sys.argv[2])
    else:
        raise SystemExit("usage: measure_driver.py forms | digest <out.json>")
