"""Per-kernel register / spill / scratch summary of a .hip file (no GPU needed).
usage: python scripts/kres.py file.hip [extra hipcc flags]"""
import re
import subprocess
import sys

f, extra = sys.argv[1], sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f, "-o", "/tmp/kres.o",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
    elif "error" in line:
        print(line)
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(.*", "", n)[:80]
    g = lambda k: r.get(k, "?")
    print(f"{n:80s} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>3} spillV {g('VGPRs Spill'):>4} spillS {g('SGPRs Spill'):>3} "
          f"scratch {g('ScratchSize'):>4} occ {g('Occupancy')}")
