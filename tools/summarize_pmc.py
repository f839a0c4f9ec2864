#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per k_update launch.
usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, statistics, sys
fetch_dir, write_dir, out_path = sys.argv[1:4]
kernel = sys.argv[4] if len(sys.argv) > 4 else "k_update"
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 320
out = {}
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if (kernel + "<") in r["Kernel_Name"] and r["Counter_Name"] == name]
    vals = [float(r["Counter_Value"]) for r in rows][-launches:]      # the timed + event passes
    out[name] = {"dispatches": len(vals), "median_KiB": statistics.median(vals), "min_KiB": min(vals), "max_KiB": max(vals)}
f, w = out["FETCH_SIZE"]["median_KiB"] * 1024, out["WRITE_SIZE"]["median_KiB"] * 1024
out["per_launch_bytes"] = {
    "fetch_raw": f, "fetch_corrected_x2": 2 * f, "write": w, "hbm_traffic": 2 * f + w,
    "algorithmic": 2 * 2003 * 2003 * 8 * (2 if kernel == "k_update2" else 1),
    "algorithmic_note": "2*L^2*w per correction x corrections per launch (k_update2: 2); the pair kernel really reads and "
                        "writes P once per launch (64.19 MB + strips), which is what the counters show",
    "note": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section) "
            "-> doubled; WRITE_SIZE is exact for 16-byte stores.  Separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with "
            "--kernel-trace only.  N = 1000, fp64; k_update2 applies two corrections per launch (algorithmic figure: 2 x 64.19 MB)."}
out["kernel"] = kernel
out["command"] = "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0"
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out["per_launch_bytes"]))
