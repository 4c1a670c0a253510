#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (tools/pmc_run.sh): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
FETCH_SIZE/WRITE_SIZE are reported in KiB.  The factor 2 on the read side is CALIBRATED on this kernel's own access pattern
(tools/traffic_cal/, profiles/traffic_calibration.json): on gfx950 every memory-side read request is one 128-byte L2 line
fill (TCC_EA0_RDREQ == TCC_MISS, TCC_EA0_RDREQ_32B == 0, TCC_BUBBLE == 0 for the patch-staging pattern and for a plain
stream alike) and FETCH_SIZE tallies each at 64 B, so bytes = 2 x FETCH_SIZE for every pattern -- a plain stream reads back
exactly 2.000.  What differs between patterns is how many LINES are fetched: the 16x16-pixel patch walk over 16-channel
chunks fetches 1.08x the tensor (a 128-B line holds two chunks and is sometimes evicted between their passes), 1.11x with
the 1-pixel halo (L2 absorbs most of the 26 % halo re-reads); a temporal-fusion layer additionally fetches the line with
chunks 0/1 from BOTH neighbour frames for half a line of use each (+25 % of the input).  WRITE_SIZE calibrates 1:1 on this
workload: the conv kernels' measured write bytes equal their algorithmic output bytes.
usage: make_traffic.py <pmc dir> <out json>"""
import collections, csv, glob, json, os, re, sys


def kernel_key(name):
    """rocprof kernel name -> the variant names bench.py uses (LaunchTimer.variant)."""
    m = re.search(r"conv3x3_kernel<bsvd::ConvCfg<(\d+), (\d+), (\d+), (\d+), (\d+), \d+(?:, (?:true|false))?>, (true|false), (\d)(?:, (true|false))?(?:, (true|false))?(?:, (true|false))?>", name)
    if m:
        mt, nt, wm, wn, st, fast, prec, mix, headf, pref = m.groups()
        planar = "[planar out]" if (mt, nt, wm, wn) == ("2", "1", "4", "1") and prec == "1" else ""     # the exit tile's only use
        return "conv3x3_kernel<%s,%s,%s,%s,%s>%s%s%s%s" % (mt, nt, wm, wn, st, "[f16x3]" if prec == "1" else "[f32]",
                                                          ("[fold8]" if mix == "true" else "") if fast == "true" else "[generic]",
                                                          planar, "[fused entry]" if headf == "true" else "") + ("[fused pair]" if pref == "true" else "")
    # winox_kernel<M, NH, NTW, MT, PERSIST, XF, YV> (XF: 0 pairs, 1 plain fp32, 2 transformed-domain input; rounds 4-5: a bool); the product's 16-row launches
    # are winox_kernel_tail<M, NH, NTW, XF, YV> (an 8-row body or folded tiles for a short last band)
    xin = {"0": "", "false": "", "1": "[f32 in]", "true": "[f32 in]", "2": "[V in]", None: ""}
    m = re.search(r"winox_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)(?:, (\d|true|false))?(?:, (true|false))?>", name)
    if m:
        return "winox_kernel<F(%s,3),%sx%s>[f16x3]" % m.groups()[:3] + ("[8 rows]" if m.group(4) == "2" else "") + ("[persistent]" if m.group(5) == "true" else "") + \
               xin[m.group(6)] + ("[V out]" if m.group(7) == "true" else "")
    m = re.search(r"winox_kernel_tail<(\d+), (\d+), (\d+), (\d|true|false)(?:, (true|false))?>", name)
    if m:
        return "winox_kernel<F(%s,3),%sx%s>[f16x3]" % m.groups()[:3] + xin[m.group(4)] + ("[V out]" if m.group(5) == "true" else "")
    m = re.search(r"wino_kernel<(\d+)>", name)
    if m:
        return "wino_kernel<F(%s,3)>[f16x3]" % m.group(1)
    m = re.search(r"(head|tail)_kernel<(\d)>", name)
    if m:
        return m.group(0)          # bench.py appends the mode tag; match on the prefix
    for k in ("pack_weights_wino_kernel", "pack_weights_split_kernel", "pack_weights_kernel", "nchw_to_nhwc_kernel",
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
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench                      # one definition of the hash: bench.kernel_sources_sha16
json.dump({"source": os.path.basename(d.rstrip("/")), "sources_sha16": _bench.kernel_sources_sha16(),
           "sources_sha16_note": "sha256 over bsvd_amd/csrc/*.hip, *.h and include/bsvd_hip.h (comments stripped) of the tree the PMC passes ran on (bench.py compares it with the tree it runs from: roofline.traffic_sources_match)",
           "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch",
           "calibration": "profiles/traffic_calibration.json: every memory-side read request is a 128-B line fill tallied at 64 B "
                          "(stream 2.000; conv patch pattern: same factor, 1.08-1.11x the tensor in lines)",
           "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
