"""Condense the rocprofv3 outputs of one round (gpurun_out/<tag>/...) into the small text/JSON
summaries committed under profiles/.
usage: python scripts/summarize_profiles.py gpurun_out/r1 profiles/r01 STEPS_TOTAL"""
import collections
import csv
import json
import os
import sys

src, dst, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
os.makedirs(os.path.dirname(dst), exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:78]


# ---- kernel stats (rocprofv3 --kernel-trace --stats) ----
rows = list(csv.DictReader(open(os.path.join(src, "rocprof", "bench_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst + "_kernel_stats.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras  ({steps} steps incl. warm-up)\n")
    f.write(f"# GPU busy {tot / 1e6 / steps:.2f} ms/step\n")
    f.write(f"{'kernel':80s} {'calls':>6s} {'avg_us':>10s} {'ms/step':>9s} {'%':>6s}\n")
    for r in rows[:45]:
        f.write(f"{short(r['Name']):80s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} "
                f"{float(r['TotalDurationNs']) / 1e6 / steps:9.3f} {float(r['Percentage']):6.2f}\n")

# ---- PMC: HBM traffic per launch (separate passes), MFMA busy ----
def pmc(dirname):
    rows = list(csv.DictReader(open(os.path.join(src, dirname, "bench_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def durations(dirname):
    rows = list(csv.DictReader(open(os.path.join(src, dirname, "bench_kernel_trace.csv"))))
    d = collections.defaultdict(list)
    for r in rows:
        d[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d


fetch, write, sq = pmc("pmc_FETCH_SIZE"), pmc("pmc_WRITE_SIZE"), pmc("pmc_sq")
dur = durations("pmc_sq")
# per-launch pairing of counters and durations of the SQ pass (same run: Dispatch_Id is the key)
sq_ids = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(os.path.join(src, "pmc_sq", "bench_counter_collection.csv"))):
    sq_ids[short(r["Kernel_Name"])][r["Counter_Name"]].append(r.get("Dispatch_Id"))
dur_by_id = {r.get("Dispatch_Id"): float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
             for r in csv.DictReader(open(os.path.join(src, "pmc_sq", "bench_kernel_trace.csv")))}
traffic = {}
with open(dst + "_pmc_summary.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <counters> (one pass per TCC counter group, as MI355X_MICROARCH.md prescribes)\n")
    f.write("# HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts 64 B per 128-B request for\n")
    f.write("# 16-B/lane coalesced reads (x2 correction); WRITE_SIZE matched the known output bytes exactly (512 MiB for a\n")
    f.write("# 2x128^3x32 fp32 output).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8).\n")
    f.write(f"{'kernel':80s} {'launches':>8s} {'avg_us':>9s} {'fetchx2_MB':>11s} {'write_MB':>9s} {'GB/s':>8s} {'mfma_busy':>9s} {'GHz':>5s}\n")
    names = sorted(sq, key=lambda k: -sum(dur.get(k, [0])))
    for k in names[:30]:
        fe = sum(fetch[k]["FETCH_SIZE"]) / max(len(fetch[k]["FETCH_SIZE"]), 1) if k in fetch else 0.0
        wr = sum(write[k]["WRITE_SIZE"]) / max(len(write[k]["WRITE_SIZE"]), 1) if k in write else 0.0
        d = sum(dur[k]) / len(dur[k])
        m = {c: sum(v) / len(v) for c, v in sq[k].items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc if cyc else 0
        byts = (2 * fe + wr) * 1024
        traffic[k] = byts
        # GRBM_GUI_ACTIVE also counts the dispatch ramp around a launch: the derived clock is meaningless (> 2.4 GHz) for
        # launches shorter than ~40 us, so it is only printed for longer ones
        # ... so: the clock of a kernel = GRBM_GUI_ACTIVE / 8 / duration over its launches of >= 200 us only, paired by
        # Dispatch_Id (shorter launches still showed > 2.4 GHz; r4: a family whose AVERAGE is below 200 us --
        # k_conv_wgrad_tr, 197 us -- got no cell although its big layers run 400 us)
        long_c = [(c, dur_by_id[i]) for c, i in zip(sq[k].get("GRBM_GUI_ACTIVE", []), sq_ids[k].get("GRBM_GUI_ACTIVE", []))
                  if dur_by_id.get(i, 0) >= 200e3]
        g = (sum(c for c, _ in long_c) / 8) / sum(t for _, t in long_c) if long_c else 0.0
        ghz = f"{g:5.2f}" if (long_c and g <= 2.45) else "    -"
        f.write(f"{k:80s} {len(dur[k]):8d} {d / 1e3:9.1f} {2 * fe / 1024:11.1f} {wr / 1024:9.1f} "
                f"{byts / d:8.0f} {busy:9.3f} {ghz}\n")
json.dump(traffic, open(dst + "_traffic_bytes_per_launch.json", "w"), indent=1)
print(open(dst + "_kernel_stats.txt").read()[:3000])
print(open(dst + "_pmc_summary.txt").read()[:4500])
