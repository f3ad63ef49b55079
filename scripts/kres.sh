#!/bin/bash
# usage: scripts/kres.sh file.hip [extra hipcc flags]   -- per-kernel register / spill / LDS summary (no GPU needed)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur={"name":m.group(1)};rows.append(cur);continue
    m=re.search(r"remark: [^:]+:\d+:\d+:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)",l)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"\(.*","",n)[:90]
    print(f"{n:90s} vgpr {r.get(\"VGPRs\",\"?\"):>4} agpr {r.get(\"AGPRs\",\"?\"):>3} spillV {r.get(\"VGPRs Spill\",\"?\"):>4} spillS {r.get(\"SGPRs Spill\",\"?\"):>3} scratch {r.get(\"ScratchSize\",\"?\"):>4} occ {r.get(\"Occupancy\",\"?\")}")
'
