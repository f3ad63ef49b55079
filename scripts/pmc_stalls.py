"""Condense rocprofv3 --pmc counter CSVs (one directory per counter group) into per-kernel averages for the MFMA kernels.
usage (on the GPU box): python scripts/pmc_stalls.py <dir with */b_counter_collection.csv> > summary.txt"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(sys.argv[1] + "/*/b_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith("k_conv_zr<2") or k.startswith("k_conv_wgrad_zs") or k.startswith("k_conv_wgrad_tr"):
            agg[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])   # sum over XCCs / instances of a dispatch
names = sorted({c for v in agg.values() for c in v})
for k, v in sorted(agg.items()):
    m = {c: sum(d.values()) / len(d) for c, d in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    print(f"\n{k}   ({len(next(iter(v.values())))} launches)")
    for c in names:
        if c in m:
            print(f"   {c:24s} {m[c]:16.0f}" + (f"   / SQ_WAVE_CYCLES {m[c] / wc:7.3f}" if wc else ""))
