#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSV passes per kernel.  usage: pmc_summary.py <dir with pass*/pmc_counter_collection.csv>"""
import collections, csv, glob, os, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bsvd::" not in k:
            continue
        k = k.split("(")[0].replace("void ", "").replace("bsvd::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, f)].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = max(len(s) for (kk, _), s in disp.items() if kk == k)
    print("%s   dispatches/pass=%d" % (k, n))
    for c, x in sorted(v.items()):
        print("   %-36s total %.4g   per-dispatch %.4g" % (c, x, x / n))
