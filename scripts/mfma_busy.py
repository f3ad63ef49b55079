"""Matrix-pipe utilisation and clock per kernel from ONE rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_BUSY_CYCLES pass (the columns `mfma_busy` / `GHz` of profiles/r0N_pmc_summary.txt, for any bench arithmetic):
    python scripts/mfma_busy.py <dir with *_counter_collection.csv + *_kernel_trace.csv> "<the command that was profiled>"
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); the clock = GRBM_GUI_ACTIVE / 8 / duration over the
launches of >= 200 us (shorter launches count the dispatch ramp)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
cc = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))[0]
kt = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[0]
dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    agg[r["Kernel_Name"].replace("void ", "").split("(")[0][:74]][r["Counter_Name"]].append((float(r["Counter_Value"]), r["Dispatch_Id"]))
print(f"# {sys.argv[2] if len(sys.argv) > 2 else ''}")
print(f"{'kernel':74s} {'n':>5s} {'avg_us':>8s} {'total_ms':>9s} {'mfma_busy':>9s} {'GHz(>=200us)':>12s}")
rows = []
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", [])
    if not gui:
        continue
    cyc = sum(v for v, _ in gui) / 8
    busy = sum(v for v, _ in c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / 1024 / cyc if cyc else 0.0
    t = [dur.get(i, 0.0) for _, i in gui]
    long_ = [(v, dur[i]) for v, i in gui if dur.get(i, 0) >= 200e3]
    ghz = (sum(v for v, _ in long_) / 8) / sum(x for _, x in long_) if long_ else float("nan")
    rows.append((sum(t), k, len(gui), sum(t) / len(t) / 1e3, busy, ghz))
for tot, k, n, avg, busy, ghz in sorted(rows, reverse=True)[:24]:
    print(f"{k:74s} {n:5d} {avg:8.1f} {tot / 1e6:9.3f} {busy:9.3f} {ghz:12.2f}")
