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
    # derived (MI355X: 8 XCDs, 256 CUs x 4 SIMDs): GRBM_GUI_ACTIVE is summed over the XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts
    # SIMD cycles (32 per v_mfma_f32_32x32x16_f16, 64 per v_mfma_f32_32x32x2_f32); SQ_* wave counters count quad-cycles
    if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        simd_cycles = v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        print("   %-36s %.3f of the SIMD-cycles the kernel was resident" % ("=> MFMA pipe busy", v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles))
    if v.get("SQ_WAVE_CYCLES"):
        wc = v["SQ_WAVE_CYCLES"]
        print("   %-36s waiting (s_waitcnt/barrier) %.2f, issue-stalled %.2f, issuing %.2f" % (
            "=> wave time", v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print("   %-36s %.2f of the LDS-active cycles" % ("=> LDS bank conflicts", v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]))
