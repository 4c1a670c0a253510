#!/usr/bin/env python3
"""Parity at BASELINE's full clip sizes, recorded (the test-suite pins small/medium cases to the oracle and checks the full
geometries through size-independent properties; this tool spends the minutes the CPU oracle needs on the whole clips):
max-abs of the HIP path (both arithmetic modes, clip schedule and chunked stream) against the CPU oracle (the reference's
algorithm on torch conv2d fp32, pinned to the reference goldens) and the PSNR-parity proxy: PSNR(out, clean) of
the engine minus PSNR(out, clean) of the oracle on the same noisy clip.
usage: python tools/full_clip_parity.py [--json out.json] [--workloads c1,c2,c3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import bench
from oracle import bsvd_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
ap.add_argument("--workloads", default="c1,c2,c3")
ap.add_argument("--wide-conv", default="auto", help="arithmetic form of the wide split-fp16 layers (bsvd_amd.engine.WIDE_CONV)")
ap.add_argument("--v-handover", default="auto", choices=["auto", "on", "off"])
ap.add_argument("--frames", default=None, help="frames per workload instead of 10 / 85 / 20, e.g. 10,10,10")
a = ap.parse_args()
FRAMES = {"c1": 10, "c2": 85, "c3": 20}
if a.frames:
    FRAMES = dict(zip(a.workloads.split(","), (int(v) for v in a.frames.split(","))))
dev = torch.device("cuda", 0)
torch.set_num_threads(bench.usable_cores())


def psnr(x, ref):
    return float(10.0 * torch.log10(1.0 / ((x.clamp(0, 1).double() - ref.double()) ** 2).mean()))


res = {"cpu_threads": torch.get_num_threads(), "rows": []}
for wname in a.workloads.split(","):
    wl = bench.WORKLOADS[wname]
    F, h, w = FRAMES[wname], wl["h"], wl["w"]
    g = torch.Generator().manual_seed(4242)
    clean = torch.rand((1, F, 3, h, w), generator=g)
    # smooth the clean clip a little so that PSNR is meaningful
    clean = torch.nn.functional.avg_pool2d(clean[0], 5, 1, 2)[None]
    lq = clean + torch.randn(clean.shape, generator=g) * bench.SIGMA
    nm = None if wl["blind"] else torch.full((1, F, 1, h, w), bench.SIGMA)
    models = {p: bench.build_model(dev, p, wl["blind"], a.wide_conv, v_handover={"auto": "auto", "on": True, "off": False}[a.v_handover]) for p in ("fp32", "f16x3")}
    P = {k: v.detach().float().cpu() for k, v in models["fp32"].state_dict().items()}
    cfg = O.default_cfg(act="relu", interm_ch=30, blind=True) if wl["blind"] else O.default_cfg()
    t0 = time.perf_counter()
    with torch.no_grad():
        want = O.stream_forward(lq, P, cfg, noise_map=nm)[0]
    t_cpu = time.perf_counter() - t0
    x = (lq if nm is None else torch.cat([lq, nm], dim=2))[0].to(dev)
    row = {"workload": wname, "wide_conv": models["f16x3"].wide_conv, "clip": [F, x.shape[1], h, w], "oracle_s": t_cpu, "oracle_fps": F / t_cpu,
           "oracle_psnr_db": psnr(want, clean[0]), "output_max": float(want.abs().max())}
    for p, m in models.items():
        with torch.no_grad():
            y = m.clip_forward(x)
            ys = m.streaming_forward(x)
        torch.cuda.synchronize()
        yc = y.cpu()
        row[p] = {"max_abs_vs_oracle": float((yc - want).abs().max()), "psnr_db": psnr(yc, clean[0]),
                  "psnr_minus_oracle_db": psnr(yc, clean[0]) - row["oracle_psnr_db"],
                  "stream_equals_clip_bitwise": bool(torch.equal(ys, y))}
        m.release_stream_buffers()
    res["rows"].append(row)
    print(json.dumps(row), flush=True)
    del models
    torch.cuda.empty_cache()
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)
