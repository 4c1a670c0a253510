#!/usr/bin/env python3
"""FETCH_SIZE (KiB, rocprofv3) per launch of each calibration kernel vs the bytes it is known to request -> correction factors."""
import collections, csv, glob, json, os, re, sys
d, out = sys.argv[1], sys.argv[2]
known = {}
for line in open(os.path.join(d, "cal.log")):
    m = re.match(r"CAL (\S+) tensor_bytes (\d+) requested_bytes (\d+) launches (\d+)", line)
    if m:
        known[m.group(1)] = dict(tensor_bytes=float(m.group(2)), requested_bytes=float(m.group(3)), launches=int(m.group(4)))
order = list(known)
vals = collections.OrderedDict()
rows = []
for f in sorted(glob.glob(os.path.join(d, "pmc", "**", "*counter_collection.csv"), recursive=True)):
    rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "cal_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
res = {}
pos = 0
for tag in order:          # dispatches arrive in program order: `launches` consecutive dispatches per case
    n = int(known[tag]["launches"])
    grp = rows[pos:pos + n]
    pos += n
    if len(grp) < n:
        break
    v = [float(r["Counter_Value"]) * 1024.0 for r in grp]
    raw = sum(v[1:]) / max(len(v) - 1, 1)          # skip the first launch
    k = known[tag]
    res[tag] = dict(kernel=grp[0]["Kernel_Name"].split("(")[0], fetch_size_bytes_raw=raw, per_launch=v, **k,
                    factor_vs_tensor=k["tensor_bytes"] / raw, factor_vs_requested=k["requested_bytes"] / raw)
# second pass (optional): raw request counters -> bytes per memory-side read request
req = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "pmc2", "**", "*counter_collection.csv"), recursive=True)):
    rr = [r for r in csv.DictReader(open(f)) if "cal_" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rr})
    pos = 0
    for tag in order:
        n = int(known[tag]["launches"])
        mine = set(ids[pos + 1:pos + n])           # skip the first launch of each case
        pos += n
        for r in rr:
            if int(r["Dispatch_Id"]) in mine:
                req[tag][r["Counter_Name"]] = req[tag].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / max(len(mine), 1)
for tag, c in req.items():
    if tag in res and c.get("TCC_EA0_RDREQ_sum"):
        res[tag]["read_requests"] = c
        res[tag]["requested_bytes_per_read_request"] = res[tag]["requested_bytes"] / c["TCC_EA0_RDREQ_sum"]
        res[tag]["tensor_bytes_per_read_request"] = res[tag]["tensor_bytes"] / c["TCC_EA0_RDREQ_sum"]
json.dump({"what": "rocprofv3 FETCH_SIZE calibration on bsvd::conv3x3_kernel's patch-staging access pattern (tools/traffic_cal/cal.hip)",
           "cases": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
