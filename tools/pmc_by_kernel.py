#!/usr/bin/env python3
"""Median of every PMC counter per kernel from a rocprofv3 --pmc --output-format csv run: pmc_by_kernel.py <dir> [name substring ...]"""
import csv, glob, statistics, sys
d = sys.argv[1]
keys = sys.argv[2:] or ["k_tick", "k_update", "k_predict"]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    kn = r["Kernel_Name"]
    if not any(k in kn for k in keys):
        continue
    short = kn.split("(")[0].replace("void nuslam::", "")
    acc.setdefault(short, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, "dispatches", len(next(iter(c.values()))))
    for name, vals in c.items():
        print("   %-28s median %.4g" % (name, statistics.median(vals)))
