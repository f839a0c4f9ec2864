#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS / occupancy per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line) or re.search(r"Name: (\S+) \[", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"\s(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
print("%-60s %5s %5s %7s %5s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for k, v in rows.items():
    print("%-60s %5s %5s %7s %5s %6s" % (k[:60], v.get("VGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize"), v.get("Occupancy"), v.get("LDS")))
