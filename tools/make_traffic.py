#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (tools/pmc_run.sh): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
reads, hence the factor 2 (/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  WRITE_SIZE calibrates 1:1 on this
workload: the conv kernels' measured 597 MB/launch equals their algorithmic output bytes exactly.
usage: make_traffic.py <pmc dir> <out json>"""
import collections, csv, glob, json, os, re, sys


def kernel_key(name):
    """rocprof kernel name -> the variant names bench.py uses (LaunchTimer.variant)."""
    m = re.search(r"conv3x3_kernel<bsvd::ConvCfg<(\d+), (\d+), (\d+), (\d+), (\d+), \d+(?:, (?:true|false))?>, (true|false), (\d)>", name)
    if m:
        mt, nt, wm, wn, st, fast, prec = m.groups()
        return "conv3x3_kernel<%s,%s,%s,%s,%s>%s%s" % (mt, nt, wm, wn, st, "[f16x3]" if prec == "1" else "[f32]",
                                                      "" if fast == "true" else "[generic]")
    m = re.search(r"(head|tail)_kernel<(\d)>", name)
    if m:
        return m.group(0)          # bench.py appends the mode tag; match on the prefix
    for k in ("pack_weights_split_kernel", "pack_weights_kernel", "nchw_to_nhwc_kernel",
              "nhwc_to_nchw_kernel", "halo_pack_kernel"):
        if k in name:
            return k
    return name.split("(")[0]

d, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE") or "bsvd::" not in r["Kernel_Name"]:
            continue
        k = kernel_key(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
res = {}
for k, v in agg.items():
    nf, nw = len(disp[k]["FETCH_SIZE"]), len(disp[k]["WRITE_SIZE"])
    fetch = v["FETCH_SIZE"] / max(nf, 1) * 1024.0
    write = v["WRITE_SIZE"] / max(nw, 1) * 1024.0
    res[k] = {"launches_sampled": nf, "fetch_bytes_raw": fetch, "write_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write}
json.dump({"source": os.path.basename(d.rstrip("/")), "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch",
           "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
