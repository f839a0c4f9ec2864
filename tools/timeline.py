#!/usr/bin/env python3
"""Print the tail of a rocprofv3 --kernel-trace csv as a timeline: timeline.py <dir> [n rows]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "nuslam" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void nuslam::", "")[:28]
    print("%-28s q%-3s start %8.1f us  end %8.1f us  (%.1f)" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                              (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
