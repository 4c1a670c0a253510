#!/usr/bin/env python3
"""Container-only: times the REAL reference forward (imported read-only through tests/golden/make_golden.py's shim) beside
the oracle's restatement (oracle.bsvd_oracle.stream_forward -- what bench.py's cpu_baseline times on the GPU box) on the
same clip, weights and thread count, and checks that both produce the same tensor.  SURVEY 8d: this legitimises the
restatement as the CPU stand-in where /root/reference does not exist.
usage: python tools/cpu_ref_vs_port.py [frames=4] [H=540] [W=960] [out.json]   (writes profiles/cpu_ref_vs_port.json by default)"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np          # noqa: E402
import torch                # noqa: E402
import make_golden as MG    # noqa: E402
from seeded import seeded_clip   # noqa: E402
from oracle import bsvd_oracle as O   # noqa: E402


RUNS = {}


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 540
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 960
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    ref = MG.import_reference()
    net = ref.BSVD(chns=[64, 128, 256], mid_ch=64, shift_input=False, in_ch=4, out_ch=3, norm="none", act="relu6",
                   interm_ch=64, blind=False, pretrain_ckpt=None)
    st = MG.load_seeded(net, 501)
    x = torch.from_numpy(seeded_clip((1, F, 4, H, W), 77, kind="sigma30"))
    P = O.to_torch_state(st)
    legs = (("reference BSVD.forward (bsvd_arch.py:490-552)", lambda: net(x)[0]),
            ("oracle.stream_forward (restatement)", lambda: O.stream_forward(x, P)[0]))
    res = {name: [1e30, None] for name, _ in legs}
    for name, _ in legs:
        RUNS[name] = []
    with torch.no_grad():
        for it in range(3):                       # interleaved, best of 3 (the container's cores are shared)
            for name, fn in legs:
                t0 = time.perf_counter()
                out = fn()
                dt = time.perf_counter() - t0
                res[name] = [min(res[name][0], dt), out]
                RUNS[name].append(dt)
                print("  run %d %-46s %.2f s" % (it, name, dt), flush=True)
    for name, (best, _) in res.items():
        print("%-48s %.2f s for %d frames = %.3f frames/s (%d threads, %dx%d)" % (name, best, F, F / best, threads, H, W))
    (ta, a), (tb, b) = res.values()
    diff = float((a - b).abs().max())
    print("max-abs difference between the two outputs: %.3e   time ratio port/reference: %.3f" % (diff, tb / ta))
    cpu = "?"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "cpu_ref_vs_port.json")
    json.dump({"what": "the REAL reference BSVD.forward (/root/reference, imported read-only) timed beside the oracle restatement "
                       "that bench.py's cpu_baseline uses where the reference cannot travel; same clip, weights, threads; "
                       "interleaved, best of 3",
               "clip": [1, F, 4, H, W], "threads": threads, "cpu": cpu, "torch": torch.__version__,
               "reference_s": ta, "port_s": tb, "reference_fps": F / ta, "port_fps": F / tb, "port_over_reference_time": tb / ta,
               "max_abs_difference": diff, "bit_identical": diff == 0.0,
               "all_runs_s": {k: v for k, v in RUNS.items()}}, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
