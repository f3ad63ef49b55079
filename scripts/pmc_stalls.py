"""Condense rocprofv3 --pmc counter CSVs (one directory per counter group) into per-kernel averages for the MFMA kernels.
usage (on the GPU box): python scripts/pmc_stalls.py <dir with */b_counter_collection.csv> > summary.txt"""
import collections
import csv
import glob
import subprocess
import sys



def demangle(m):
    """_Z<len><name>I<args>E...: Li<n>E int, Lb<0|1>E bool, DF16_ _Float16, DF16b __bf16, f float"""
    import re
    mm = re.match(r"_Z(\d+)", m)
    if not mm:
        return m
    n = int(mm.group(1))
    name, rest = m[mm.end():mm.end() + n], m[mm.end() + n:]
    if not rest.startswith("I"):
        return name
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        t = re.match(r"Li(\d+)E|Lb([01])E|(DF16_)|(DF16b)|(f)", rest[i:])
        if not t:
            break
        args.append(t.group(1) if t.group(1) is not None else ("true" if t.group(2) == "1" else "false") if t.group(2) is not None
                    else "f16" if t.group(3) else "bf16" if t.group(4) else "float")
        i += t.end()
    return f"{name}<{', '.join(args)}>"

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(sys.argv[1] + "/*/b_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith("_Z"):   # rocprofv3 (and c++filt) leave kernels with _Float16 / __bf16 template arguments mangled
            k = demangle(k)
        if k.startswith("k_conv_zr<2") or k.startswith("k_conv_wgrad_zs") or k.startswith("k_conv_wgrad_tr") or k.startswith("k_conv_fwd_mfma"):
            agg[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])   # sum over XCCs / instances of a dispatch
names = sorted({c for v in agg.values() for c in v})
for k, v in sorted(agg.items()):
    m = {c: sum(d.values()) / len(d) for c, d in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    print(f"\n{k}   ({len(next(iter(v.values())))} launches)")
    for c in names:
        if c in m:
            print(f"   {c:24s} {m[c]:16.0f}" + (f"   / SQ_WAVE_CYCLES {m[c] / wc:7.3f}" if wc else ""))
