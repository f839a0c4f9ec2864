#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch of the sweep kernel.
usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json> [kernel substring, e.g. "k_update2<double"] [last N launches]
       [state_len L] [bytes per element w] [filters B] [ticks per launch]
The record carries nuslam_build_info() of the library in the tree it is run from and the full kernel name rocprofv3
reported, so bench.py quotes it only beside the code it was measured on."""
import csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd"))
fetch_dir, write_dir, out_path = sys.argv[1:4]
kernel = sys.argv[4] if len(sys.argv) > 4 else "k_update2<double"
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 320
L = int(sys.argv[6]) if len(sys.argv) > 6 else 2003
w = int(sys.argv[7]) if len(sys.argv) > 7 else 8
B = int(sys.argv[8]) if len(sys.argv) > 8 else 1
ticks = int(sys.argv[9]) if len(sys.argv) > 9 else 1        # ticks inside one launch (k_run_fused): per-tick figures are added
out = {}
kname = None
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    kname = rows[-1]["Kernel_Name"]
    vals = [float(r["Counter_Value"]) for r in rows][-launches:]
    out[name] = {"dispatches": len(vals), "median_KiB": statistics.median(vals), "min_KiB": min(vals), "max_KiB": max(vals)}
f, wr = out["FETCH_SIZE"]["median_KiB"] * 1024, out["WRITE_SIZE"]["median_KiB"] * 1024
out["per_launch_bytes"] = {
    "fetch_raw": f, "fetch_corrected_x2": 2 * f, "write": wr, "hbm_traffic": 2 * f + wr,
    "min_bytes": 2 * L * L * w * B,
    "note": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section) "
            "-> doubled; WRITE_SIZE is exact for 16-byte stores.  Separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with "
            "--kernel-trace only.  min_bytes = 2*L^2*w*B: P read once and written once per launch."}
if ticks > 1:
    out["per_tick_bytes"] = {"ticks_per_launch": ticks, "hbm_traffic": (2 * f + wr) / ticks, "fetch_corrected_x2": 2 * f / ticks, "write": wr / ticks,
                             "note": "one launch runs `ticks` ticks: P is read once when it begins and written once when it ends (min_bytes), "
                                     "in between a tick moves the rows / columns of the next index set, the edge tiles and the strips"}
out["kernel_name"] = kname
try:
    import nuslam_hip
    out["build_info"] = nuslam_hip.build_info()
except Exception as e:
    out["build_info"] = "unknown (%s)" % e
out["command"] = "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py ..."
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out["per_launch_bytes"]), out["kernel_name"], out["build_info"])
