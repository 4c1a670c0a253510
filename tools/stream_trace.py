#!/usr/bin/env python3
"""Workload + analysis for a rocprofv3 kernel trace of the per-frame stream schedule.
  run    : python tools/stream_trace.py run [HxW] [frames]        (under rocprofv3 --kernel-trace --output-format csv)
  analyse: python tools/stream_trace.py analyse <kernel_trace.csv>   -> busy time vs span of the steady-state steps"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch, bsvd_amd
    H, W = map(int, (sys.argv[2] if len(sys.argv) > 2 else "540x960").split("x"))
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    m = bsvd_amd.BSVD(chns=[64, 128, 256], mid_ch=64, norm="none", act="relu6", interm_ch=64, pretrain_ckpt=None,
                      precision="f16x3", stream_overlap=os.environ.get("LAG", "0") == "1").to(dev).eval()
    x = torch.rand(F, 4, H, W, device=dev)
    with torch.no_grad():
        for rep in range(3):
            if os.environ.get("LAG", "0") == "1":
                m.streaming_forward(x)
            else:
                for i in range(F):
                    m.feedin_one_element(x[i:i + 1])
                for i in range(17):
                    m.feedin_one_element(None)
                m.reset()
            torch.cuda.synchronize()


def analyse(path):
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
    # last third of the trace = third repetition; take the kernels of its middle 50 % (steady state)
    n = len(ks)
    seg = ks[int(n * 0.75):int(n * 0.92)]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    conv = [k for k in seg if "conv3x3" in k[2] or "head_kernel" in k[2] or "tail_kernel" in k[2]]
    print("kernels %d, span %.3f ms, sum of durations %.3f ms (%.1f %%), mean gap %.2f us, median gap %.2f us, max gap %.1f us"
          % (len(seg), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3, max(gaps) / 1e3))
    per = {}
    for s, e, name in seg:
        short = name.split("(")[0][-70:]
        d = per.setdefault(short, [0, 0])
        d[0] += e - s; d[1] += 1
    for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print("  %8.3f ms %5d x %7.1f us  %s" % (t / 1e6, c, t / c / 1e3, k))
    frames = sum(1 for k in seg if "head_kernel" in k[2])
    print("frames in window: %d -> %.3f ms/frame span, %.3f ms/frame busy" % (frames, span / 1e6 / max(frames, 1), busy / 1e6 / max(frames, 1)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        analyse(sys.argv[2])
